// qg_kernels.h — launch interface between the C-ABI layer (qg_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "qg_eltwise_args.h"
#include "qg_plan.h"

// geometry of one operand in host (reference) layout as seen by pack / fill / unpack kernels
struct QOperandGeom {
    int64_t rows, K;        // logical rows (M for A, N for B) and reduction length
    int64_t rs, ks;         // host element index of (r,k) = r*rs + k*ks   (column-major tensors)
    int32_t elem_bytes;     // host element size (complex: whole struct)
    int32_t off[2], sb[2];  // byte offset / byte width (4|8) of each part inside the element
    int32_t parts;          // 1 real, 2 complex
    int32_t W[2], S[2];     // format of each part (range check, synthetic fill)
    int32_t F[2], Q[2], O[2]; // quantise-on-load: fracBits, QuMode, OfMode of each part
    int64_t k0;             // this packed operand holds the reduction indices [k0, k0 + K) of the host tensor (a k-chunk of a
                            // composite linear plan, qg_api.hip; 0 otherwise)
};

// packed (device-private) layouts
//   tree  : [part][rows_p][K_p] containers of cbytes (4|8), K contiguous
//   limbs : int8 balanced base-256 digits, PRE-TILED and PRE-SWIZZLED for the MFMA kernel:
//           [row tile][k tile][limb][tr rows][bk bytes], where the 16-byte chunk c of row r sits at
//           slot c ^ ((r / (256/bk)) % (bk/16)).  One (row tile, k tile) block is exactly the LDS
//           image of that operand for one pipeline stage, so the kernel streams it with lane-linear
//           LDS-DMA (1 KiB per wave instruction, fully sequential in HBM; no row pitch, hence no
//           power-of-two-stride channel camping).
struct QPackedGeom {
    int64_t rows_p, K_p;    // padded extents (rows_p per part: complex limb operands stack their parts along rows)
    int32_t cbytes;         // container bytes (tree layout), 1 for limb layout
    int32_t limbs;          // 0 = tree layout, >0 = limb layout
    int32_t tr, bk;         // limb layout: rows per tile, k bytes per tile
    // limb layout with limbs > 1: the planes are followed by a 256-byte trailer whose first word is the OR over all
    // elements of (1 << l) for every limb l that is non-zero somewhere ("plane mask", written by the pack kernels).  The
    // MFMA kernel skips the products of planes that are zero everywhere: data that stays inside [-32640, 32639] never
    // touches the third int8 limb of a 17-bit format, and 4 instead of 9 products are exact for it.
    int64_t trailer;        // byte offset of the trailer (0: none)
    // Karatsuba layout (2 x 2 digits, operands of at most 12 value+sign bits): the two planes hold the UNSIGNED base-64 digits
    // of x + bias (bias = 2^W for a signed format, so the biased value is non-negative; padding holds 0), and int64
    // row_sum[rows_p] = sum_k (x + bias) follows the trailer — the kernel's epilogue takes the bias back out with it.
    int32_t digit6;
    // limb layout: the planes hold digits limb0 .. limb0 + limbs - 1 of the balanced base-256 expansion (a limb GROUP of an
    // operand of more than 3 limbs, composite linear plans; 0 otherwise)
    int32_t limb0;
    // CENTRED operands (offs != 0; qg_plan.h: qg_limbs_centred): the planes hold the balanced digits of x + bias (bias = -centre;
    // padding holds 0) and int64 row_sum[rows_p] = sum_k (x + bias) sits at rowsum_off — sum a b = sum a'b' - biasB rsA[i]
    // - biasA rsB[j] + K biasA biasB, taken back out by the MFMA kernels' epilogues / the composite plan's combine pass.
    // offs 1: this packed operand owns its row sums (the pack zeroes and fills them); 2: a sub-operand of a composite plan adding
    // its k-chunk to the operand's one array (zeroed by the caller; only the groups with limb0 == 0 add); 3: ... a group that does not
    int32_t offs, pad_;
    int64_t bias, rowsum_off;
};
enum { QG_TRAILER_BYTES = 256, QG_MASK_WORDS = 64 };

// The plane mask lives in ALL 64 words of the trailer: a pack kernel's waves OR their findings into word (wave id % 64) — on
// one word the 16 384 atomics of a 4096^2 operand serialise (0.2 ms against 0.03 ms for the pack itself, found when bench.py
// began to pack non-zero data) — and a consumer ORs the 64 words: one load per lane and a wave reduction.
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned qg_plane_mask(const uint32_t* m)
{
    if (!m) return 7u;
    unsigned v = m[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o; o >>= 1) v |= __shfl_xor(v, o);
    return (unsigned)__builtin_amdgcn_readfirstlane(v);
}
#endif

struct QCGeom {
    int64_t M, N, Mp, Np;   // logical / padded extents of packed C
    int32_t tm, tn;         // 0: row-major [part][Mp][Np]; else tiled [part][Mp/tm][Np/tn][tn cols][tm rows]
    int32_t cbytes;         // 1|2|4|8 container
    int32_t parts;
    int64_t ldc;            // host leading dimension (elements)
    int32_t elem_bytes, off[2], sb[2];
};

// element index of (part, m, n) in packed C
__host__ __device__ inline int64_t qg_c_index(const QCGeom& c, int part, int64_t m, int64_t n)
{
    if (c.tm == 0) return ((int64_t)part * c.Mp + m) * c.Np + n;
    return ((((int64_t)part * (c.Mp / c.tm) + m / c.tm) * (c.Np / c.tn) + n / c.tn) * c.tn + n % c.tn) * c.tm + m % c.tm;
}

hipError_t qg_launch_pack(const QOperandGeom& g, const QPackedGeom& p, const void* src, void* dst, int check_range,
                          int* range_flag, hipStream_t st, int generic = 0);   // generic: the any-format kernel even where a fast path exists
// the same with a column-major tensor of DOUBLES as source (complex: {re, im} pairs), quantised on load with each
// part's own QuMode / OfMode exactly as Qu_s(double) does (QuBLAS.h:2387-2393)
hipError_t qg_launch_pack_f64(const QOperandGeom& g, const QPackedGeom& p, const void* src, void* dst, hipStream_t st, int generic = 0);
hipError_t qg_launch_fill(const QOperandGeom& g, const QPackedGeom& p, uint64_t seed, int dist, void* dst, hipStream_t st);
hipError_t qg_launch_unpack_c(const QCGeom& c, const void* packed, void* dst, hipStream_t st, int generic = 0);

// complex linear class: combine the four raw dot-product blocks of D (tiled int64) into packed complex C [2][M][N]
struct QCplxCombine {
    const int64_t* D;   // [Mp/tm][Np/tn][tn][tm], Mp = 2*Mh, Np = 2*Nh
    char* C;
    int64_t M, N, Mh, Nh, Np;
    int32_t tm, tn, cbytes;
    int32_t sh[4];
    QStep to_c[2];
};
hipError_t qg_launch_cplx_combine(const QCplxCombine& g, hipStream_t st);

// composite linear plans (operands of more than 3 int8 limbs, or K beyond the int32 accumulators' exact range): the limb-group /
// k-chunk sub-GEMMs store RAW dot products (int32 for single-limb pairs, int64 otherwise) in slabs of one common packed-C
// layout; this pass forms  s = acc_in + sum_j (slab_j << sh[j])  per element, exactly, and either keeps it for the next
// k-chunk (acc_out) or rounds + overflow-handles it ONCE into C's format (out).  wide: 128-bit sums (two-word accumulator
// elements, C containers of up to 16 bytes).
enum { QG_MAX_SLABS = 9 };
struct QLinCombine {
    const void* slab[QG_MAX_SLABS];
    int32_t sh[QG_MAX_SLABS];
    int32_t n_slabs, slab_bytes;   // 4 | 8
    int64_t n;                     // elements (the padded packed-C index space)
    const void* acc_in;            // nullptr: start from 0
    void* acc_out;                 // nullptr: this is the last chunk
    void* out;                     // packed C (cbytes containers), written when acc_out == nullptr
    int32_t cbytes, wide;
    QStep to_c;
    // centred operands (QPackedGeom::offs): the correction of the last chunk, per element of the tiled packed-C index space
    // (element i: tile i / (tm tn), column (i / tm) % tn, row i % tm of that tile; tiles_n tiles per tile row)
    const int64_t* rsA;
    const int64_t* rsB;
    int64_t biasA, biasB, corr;   // corr: the reduction length K (K biasA biasB is formed in the accumulator's width)
    int32_t tm, tn;
    int64_t tiles_n;
};
hipError_t qg_launch_lin_combine(const QLinCombine& g, hipStream_t st);

// exact tree evaluation, any descriptor (real / complex, any K), 64-bit arithmetic
hipError_t qg_launch_tree_generic(const QTreeTable* dev_table, int parts, const void* A, const void* B, void* C, int64_t M,
                                  int64_t N, int64_t K, const QPackedGeom& pa, const QPackedGeom& pb, const QCGeom& pc,
                                  hipStream_t st, int wide = 0);   // wide: 128-bit values (C containers of up to 16 bytes)

// exact tree evaluation, real descriptors with K = 2^p >= 32 and 32-bit intermediates (A, B packed as int32)
hipError_t qg_launch_tree_fast(const QTreeTable* dev_table, int n_levels, int split_s, int mul24, int mode, const void* A, const void* B,
                               void* C, int64_t M, int64_t N, int64_t K, int cbytes, hipStream_t st);

// exact tree evaluation, real descriptors with 5..16 levels and 64-bit values (A, B packed as int32 or int64, K = 2^n_levels)
hipError_t qg_launch_tree64(const QTreeTable* dev_table, int n_levels, const void* A, const void* B, void* C, int64_t M, int64_t N,
                            int64_t K, int abytes, int bbytes, int cbytes, hipStream_t st);

// exact tree evaluation of ONE output column (batched Qreduce / GEMV): A [M][K] int32, B [K] int32
hipError_t qg_launch_gemv(const QTreeTable* dev_table, int n_levels, int b_is_bit, int fixed_mode, const void* A, const void* B, void* C,
                          int64_t M, int64_t K, int cbytes, hipStream_t st, int wide = 0);   // wide: 64-bit tree values (4-byte elements)

// exact tree evaluation, complex descriptors with K = 2^p >= 32 and 32-bit intermediates
hipError_t qg_launch_tree_cplx_fast(const QTreeTable* dev_table, int n_levels, int fixed_modes, int tf, const void* A, const void* B, void* C,
                                    int64_t M, int64_t N, int64_t K, int cbytes, hipStream_t st);

// linear class on int8 MFMA with LA x LB limbs
struct QMfmaCfg {
    int variant;    // 0 = no kernel for this limb combination
    int TM, TN, BK; // output tile and k-tile (bytes) the packed operands are padded to
};
QMfmaCfg qg_mfma_pick(int LA, int LB, int64_t M, int64_t N, uint32_t opt_flags = 0);
struct QMfmaArgs {
    const int8_t* A;  // [LA][Mp][Kp]
    const int8_t* B;  // [LB][Np][Kp]
    void* C;          // [Mp][Np] containers
    int64_t Mp, Np, Kp;
    int32_t cbytes;
    int32_t variant;
    QStep to_c;
    // Karatsuba variant (kara != 0): sum a*b = sum a'b' - biasB * rsA[i] - biasA * rsB[j] + corr, with a' = a + biasA etc.
    const int64_t* rsA;
    const int64_t* rsB;
    int64_t biasA, biasB, corr;
    int32_t kara, pad3_;
    const uint32_t* maskA;  // plane masks of the packed operands (QPackedGeom::trailer); nullptr: all planes
    const uint32_t* maskB;
    int32_t has_ep, pad_;   // fused element-wise epilogue: C below is then packed D (ep.dbytes containers)
    uint32_t* dbg;          // diagnostic build: in-kernel clock stamps (qg_mfma_pp.hip); nullptr otherwise
    // c_host != 0 (k_mfma_pp / k_mfma_ppl, 4- and 8-byte containers): C is the REFERENCE layout on the device — element (i, j)
    // at i + j * c_ld, c_M x c_N logical — and the epilogue's runs of 4 rows land there directly: no packed C, no unpack pass.
    // c_vec: every run of 4 rows is 16-byte aligned (base and c_ld permitting): one vector store per run.
    int32_t c_host, c_vec;
    int64_t c_ld, c_M, c_N;
    QEpTable ep;
    QEpArgs epa;
};
hipError_t qg_launch_mfma(int LA, int LB, const QMfmaArgs& a, hipStream_t st);
// single limb, 256x256 tiles, 128-byte k-tiles, two wave groups alternating on the matrix cores (qg_mfma_pp.hip; variant 9)
hipError_t qg_launch_mfma_pp(const QMfmaArgs& a, hipStream_t st);
// 3 x 3 and 2 x 2 limbs, 128x128 tiles, the two-group scheme with one A limb plane per phase (qg_mfma_ppl.hip; variant 10)
bool qg_mfma_ppl_applies(int LA, int LB, const QMfmaArgs& a);
hipError_t qg_launch_mfma_ppl(int limbs, const QMfmaArgs& a, hipStream_t st);   // limbs: 3 (3 x 3, plane masks) or 2 (2 x 2 on two-plane storage)

// hipFuncAttributeMaxDynamicSharedMemorySize is set once per DEVICE (one process may drive several: qgemul_run_sharded);
// `done` holds one bit per device ordinal of the calling thread's current device
inline hipError_t qg_lds_attr(const void* fn, int bytes, std::atomic<uint64_t>& done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

// A/B and ablation switches read from the environment exist only in the diagnostic build (libqugemm_diag.so, -DQG_DIAG,
// used by tools/); the product library ignores the environment altogether.
#ifdef QG_DIAG
#define QG_DIAG_ENV(name) (getenv(name) != nullptr)
#else
#define QG_DIAG_ENV(name) false
#endif

// element-wise epilogue as its own pass (kernels that do not fuse it) and the operand packer; see qg_eltwise.h
hipError_t qg_launch_eltwise(const QEltwiseArgs& g, hipStream_t st);
hipError_t qg_launch_pack_e(const QCGeom& c, int part, const void* src, int64_t ld, int stride, int off, int src_bytes, void* dst, int ebytes,
                            hipStream_t st);

// BitStream export of packed C (qg_pack.hip): n = M*N elements of `width` characters each
struct QBitsArgs {
    QCGeom c;
    const char* packed;
    char* out;
    int32_t width, tensor_chunk, elem_chunk, packed_bits;
};
hipError_t qg_launch_bitstream(const QBitsArgs& a, hipStream_t st);
// ... of a packed COMPLEX C: an element is the string "(re-bits, im-bits)" after its chunk reversal; tab[j] says what output
// character j of an element is: 0..63 bit k of the real part, 64..127 bit (k - 64) of the imaginary part, 128.. a literal
enum { QG_BITS_LIT_OPEN = 128, QG_BITS_LIT_COMMA = 129, QG_BITS_LIT_SPACE = 130, QG_BITS_LIT_CLOSE = 131 };
struct QBitsCplxArgs {
    QCGeom c;
    const char* packed;
    char* out;
    int32_t width, nbits, tensor_chunk, packed_bits;   // characters per element; binary characters per element (wr + wi)
    uint8_t tab[136];
};
hipError_t qg_launch_bitstream_cplx(const QBitsCplxArgs& a, hipStream_t st);
