// qg_plan.h — host-side analysis of a qgemul_desc: validation, exactness class (SURVEY.md
// §8-a13), required integer width, kernel choice and the pre-resolved per-node tables the
// device kernels read.  Pure host code: compiled into libqugemm.so, usable without a GPU.
#pragma once
#include <stdint.h>

#include "../../include/qgemul.h"
#include "qg_ops.h"

// One fixed-mode step of the complex kernel (qg_tree_cplx.hip, QAnalysis::cplx_fixed_ok == 2) in the form the kernel
// executes without a branch, 8 dwords = one scalar load:
//   add / sub slot:  v = (x * ka + t) +/- y * kb;   v >>= d;   v = clamp(v, lo, hi)       (ka, kb = 2^alignment shift)
//   multiply slot:   v = (x * ka) * y + t;          v >>= d;   v = clamp(v, lo, hi)       (ka = 2^(left shift of an exact product))
//   tree node:       v = x + v + t;   v <<= ls;     v >>= d;   v = clamp(v, lo, hi)       (ls: a level type with MORE fraction bits)
// t = the rounding mode's addend (TRN::TCPL 0, RND::POS_INF 2^(d-1), RND::NEG_INF 2^(d-1) - 1); an identity step is d = 0,
// t = 0 and the full int32 range.  cplx_fixed_ok == 3 / >= 8: `skip` also packs the step's rounding / overflow kind
// (qg_fix.h: fx_finish_packed / fx_finish_feat; the latter's rounding factor is `ls` of a slot, `ka` of a node).
// Records of the REAL kernel (product, nodes) have no factors.  fast_mode 3 (every step clamps): as above.  fast_mode 4 (a step
// tests the range or wraps): the value is kept biased by -lo of its format, and the fields mean: t = the node's one constant
// (rounding addend, change of bias), kb = overflow kind (0 clamp, 1 SAT::ZERO, 2 WRP::TCPL, 4 none), hi = span = hi - lo,
// lo = the bias B = -lo (the biased zero); fmul[0].ka = the root's bias (qg_plan.cpp).  fast_mode 5 (a format too wide for that):
// unbiased values, lo / hi = the bounds, kb = 0 clamp, 1 SAT::ZERO, 2 / 3 WRP::TCPL signed / unsigned.
struct QFix {
    int32_t ka, kb, t, d, lo, hi, skip, ls;
};
// parameters of the LEFT-JUSTIFIED forms (QTreeTable::lj, lj16)
struct QJustify {
    int32_t s, t[4], e[6];
    int32_t g[4];   // complex: product i has g[i] FEWER fraction bits than the common format (same integer bits): its mask clears s + g[i] bits
    int32_t pad_;
};

// everything a kernel needs about the arithmetic, laid out for device reads (plan-owned buffer)
struct QTreeTable {
    int32_t is_complex, cmul, n_levels, parts;
    // the 32-bit / 64-bit register-counter kernels walk at least 5 levels (32 leaves): a shorter tree is continued with identity
    // levels (x + 0 in x's own format), its operands zero-padded to 32 leaves; n_levels_k = max(n_levels, 5) is what those
    // kernels and the packed geometry use (the general kernel keeps n_levels)
    int32_t n_levels_k, pad_[3];
    QNode mul[8];                         // product sub-ops in qgemul.h slot order
    QNode level_add[2][QG_MAX_LEVELS];    // pair add of level l (inputs have equal formats)
    QStep level_cvt[2][QG_MAX_LEVELS];    // store into the level buffer (identity for real GEMMs)
    QStep leftover[2][QG_MAX_LEVELS];     // odd-leftover copy into level l's buffer
    QStep c_cvt[2];                       // root -> C, valid when the tree has all n_levels levels
    // complex fixed-mode kernel, register-lean form (valid when QAnalysis::cplx_fixed_ok == 2)
    QFix fmul[8];
    QFix fadd[2][QG_MAX_LEVELS];
    QFix fcvt[2][QG_MAX_LEVELS];
    // ... its "one clamp for the whole loop" form (cplx_fixed_ok == 4): the common range, then per product (TF: A, B, C; Basic: ac,
    // bd, ad, bc) the rounding addend and the right shift, then the factors (powers of two) the operand planes are staged with
    // (TF: (a+b), (c+d), (b-a); Basic: a, b, c, d) — which carry the products' exact left shifts, the shifts growing to match
    struct { int32_t lo, hi, t[4], d[4], k[4], pad_[2]; } uni;
    // LEFT-JUSTIFIED forms (qg_fix.h: one signed SAT::TCPL format held as x * 2^s; real fast_mode 6, complex cplx_fixed_ok 5): the
    // shift s, per product the rounding addend scaled to the justified product, and the left shifts the operand planes are staged
    // with (real: A, B; TF: (a+b), b, (b-a), c, (c+d), d; Basic: a, b, c, d), which justify the products
    QJustify lj;
    // ... and in packed 16-bit halves (x * 2^s, s = 16 - width; real fast_mode 7, complex cplx_fixed_ok 6), the same fields
    QJustify lj16;
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
    int cplx_fixed_ok;       // ... and every step on the path is RND::POS_INF (or exact) + SAT::TCPL: fixed-mode variant (2: compact records; 3: ... with rounding / overflow kinds; 8 + f: ... of the branch-free feature set f; 4: one clamp for the whole loop; 5: ... on left-justified values; 6: ... in packed 16-bit halves)
    int cplx_fixed_base;     // cplx_fixed_ok 5 / 6: the form the descriptor has without the justified values (4 when the strict one-clamp form holds, else 2)
    int fast_mode;           // 0 runtime modes; 1 one format everywhere, TCPL + SAT::ZERO; 2 TCPL + SAT::TCPL; 3 / 4 / 5 per-level formats, compact steps (QFix; 3: every step clamps, 4: biased values, 5: unbiased); 6 one signed SAT::TCPL format on left-justified values, 7 ... in packed 16-bit halves, 8 ... 32-bit products, packed 16-bit nodes
    int fast_mode_base;      // fast_mode 6 / 7 (one signed SAT::TCPL format, left-justified saturating steps): the form (2 / 3) the descriptor has without it
    int lj_unsigned;         // fast_mode 6 ... 9 on an unsigned format (all operands unsigned): the unsigned saturating instructions
    int tree64_ok;           // the 2x2-outputs-per-lane 64-bit tree kernel applies (real, 5..16 levels)
    int gemv_ok;             // the one-column 32-bit tree kernel applies (N = 1, K = 2^p >= 16)
    int gemv_wide_ok;        // ... or its 64-bit-value form: elements of at most 32 storage bits, wider sums / level types
    int gemv_fixed;          // 1 / 2: every tree level has one format, no rounding shift, SAT::ZERO / SAT::TCPL (fixed-mode nodes); 3 / 5: per-level formats in compact records
    int gemv_w32;            // gemv_wide_ok descriptors whose product and every level live in ONE signed SAT::TCPL format of exactly 32 bits: values stay 32-bit words, a node is one saturating add (k_gemv<., 6>)
    int gemv_b_bit;          // ... and B is a 0/1 vector whose product with a is a itself (the Qreduce lowering)
    int wide;                // an intermediate, a level / product format or C needs more than 62 bits: 128-bit kernels (qg_ops.h: qg_step_w)
    int generic_only;        // C's WRP::TCPL_SAT can let the root through unclamped: the general kernels / the composite plan's combine pass only
    int band;                // C may hold a value outside its format (host-word containers): WRP::TCPL_SAT, or a multi-word value in [2^63, 2^64) / [-2^64, -2^63) can reach a one-word target (QStep::refcmp)
    char reason[96];
    QTreeTable tree;
    QLinearEpilogue lin;
};

void qg_analyze(const qgemul_desc* d, QAnalysis* out);

// fused element-wise epilogue (qgemul_epilogue), pre-resolved for the device: stage k combines the running value x
// with its operand e (node.sa shifts the FIRST operand of the Qop, node.sb the second), node.q rounds/overflows into
// the stage's result format; to_d is the destination tensor's converting assignment
struct QEpStage {
    int32_t op, x_first, scalar, ebytes;   // ebytes: container of a packed tensor operand (1|2|4|8)
    QNode node;
    QStep cvt;                             // assignment to the stage's tensor (identity for the last stage: to_d)
};
struct QEpTable {
    int32_t n, dbytes;                     // stages; container bytes of packed D
    int32_t bits32, max_bits;              // bits32: every value of the chain (and C itself) fits 32-bit arithmetic
    QEpStage st[QG_MAX_EW];
    QStep to_d;
};
// validates `ep` against the GEMM's C format; returns QG_OK / QG_EINVAL / QG_EUNSUPPORTED (reason filled)
int qg_analyze_ep(qfmt c, const qgemul_epilogue* ep, QEpTable* out, int* max_bits, char* reason, size_t reason_len);

// host-layout element geometry (int32/int64 per part; complex = struct {real; imag;})
struct QHostElem {
    int size, off[2], sb[2];
};
QHostElem qg_host_elem(const qfmt f[2], int is_complex);

// int8 limbs needed to hold every raw value of format f as balanced base-256 digits
int qg_limbs_for(qfmt f);
// ... of x - c for the centre c that needs the fewest (returned in *centre): an interval of width <= 256^L - 1 fits L balanced
// digits once it is moved onto [-128 S, 127 S], S = (256^L - 1) / 255 — a signed format of exactly 8 L bits is one value too wide
// for L plain balanced digits on the positive side, an unsigned format of w bits wastes a digit on its sign
int qg_limbs_centred(qfmt f, int64_t* centre);
