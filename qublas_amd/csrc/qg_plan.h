// qg_plan.h — host-side analysis of a qgemul_desc: validation, exactness class (SURVEY.md
// §8-a13), required integer width, kernel choice and the pre-resolved per-node tables the
// device kernels read.  Pure host code: compiled into libqugemm.so, usable without a GPU.
#pragma once
#include <stdint.h>

#include "../../include/qgemul.h"
#include "qg_ops.h"

// everything a kernel needs about the arithmetic, laid out for device reads (plan-owned buffer)
struct QTreeTable {
    int32_t is_complex, cmul, n_levels, parts;
    QNode mul[8];                         // product sub-ops in qgemul.h slot order
    QNode level_add[2][QG_MAX_LEVELS];    // pair add of level l (inputs have equal formats)
    QStep level_cvt[2][QG_MAX_LEVELS];    // store into the level buffer (identity for real GEMMs)
    QStep leftover[2][QG_MAX_LEVELS];     // odd-leftover copy into level l's buffer
    QStep c_cvt[2];                       // root -> C, valid when the tree has all n_levels levels
};

// epilogue of the linear class: exact dot product at frac (Fa+Fb) -> C
struct QLinearEpilogue {
    QStep to_c[2];
    // complex linear class: with the raw dot products P1 = sum a*c, P2 = sum b*d, P3 = sum a*d, P4 = sum b*c
    // (x = a+bi from A, y = c+di from B),  re = (P1 << sh[0]) - (P2 << sh[1]),  im = (P3 << sh[2]) + (P4 << sh[3]),
    // each then rounded/overflowed ONCE into C's part by to_c[part]
    int32_t sh[4];
};

struct QAnalysis {
    int status;              // QG_OK / QG_EINVAL / QG_EUNSUPPORTED
    int cls;                 // QG_CLASS_*
    int max_bits;            // widest signed intermediate (bits incl. sign)
    int linear_ok;           // every conversion on the path is provably the identity
    int dot_bits;            // class L: bits of the exact K-term dot product
    int max_bits_np;         // widest intermediate not counting unrounded products
    int tree_fast_ok;        // the 32-bit tree kernel applies
    int split_s;             // > 0: product evaluated split at its rounding shift
    int mul24_ok;            // multiplies fit v_mul_i32_i24
    int cplx_fast_ok;        // the 32-bit complex tree kernel applies
    int fast_mode;           // 0 runtime modes; 1 one format everywhere, TCPL + SAT::ZERO; 2 TCPL + SAT::TCPL
    char reason[96];
    QTreeTable tree;
    QLinearEpilogue lin;
};

void qg_analyze(const qgemul_desc* d, QAnalysis* out);

// host-layout element geometry (int32/int64 per part; complex = struct {real; imag;})
struct QHostElem {
    int size, off[2], sb[2];
};
QHostElem qg_host_elem(const qfmt f[2], int is_complex);

// int8 limbs needed to hold every raw value of format f as balanced base-256 digits
int qg_limbs_for(qfmt f);
