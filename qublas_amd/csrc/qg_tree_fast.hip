// qg_tree_fast.hip — exact tree-order evaluation for the hot real-valued shapes, 32-bit VALU path
// ("i32 wavefront-MAC", BASELINE.json configuration 3).
//
// Same semantics as k_tree_generic (qg_tree.hip): every product is quantised (Qmul,
// /root/reference/include/QuBLAS.h:3152-3170) and every node of the pairwise tree is quantised (Qadd inside
// Reducer::reduce_impl, QuBLAS.h:4960-4984) in the reference's order.  What changes is the mapping:
//   * all values fit 32 bits (the planner proves it), so nodes are v_add + 2..3 VALU ops;
//   * products wider than 32 bits never materialise: with the product's rounding shift s,
//     b = bh*2^s + bl (bl unsigned), a*b = (a*bh)*2^s + a*bl, so floor(a*b / 2^s) = a*bh + ((a*bl) >> s)
//     and the discarded low bits are (a*bl) & (2^s-1) — two 32-bit multiplies, split done once per
//     staged B element instead of once per MAC;
//   * each lane owns 4 rows x 2 columns (tx and tx + 16 of the block) of outputs; the tree is streamed as a binary counter whose lower
//     four levels (16 leaves) are fully unrolled in registers and whose upper levels are a
//     statically indexed register array, so nothing spills to scratch;
//   * per-level formats and modes stay runtime values (wave-uniform scalar loads and scalar
//     branches hoisted around the 8-output loops), so one kernel serves every real descriptor
//     whose K is a power of two >= 32.
#include <hip/hip_runtime.h>

#include "qg_fix.h"
#include "qg_kernels.h"

namespace {

constexpr int KC = 32;      // k-chunk staged in LDS
constexpr int TMB = 64;     // rows of C per block   (16 thread rows x 4)
constexpr int TNB = 32;     // columns of C per block (16 thread cols x 2)
constexpr int NOUT = 8;
constexpr int PITCH = KC + 4;

// round (d > 0) NOUT values in place; x = h*2^d + l is given as (h, l) when SPLIT, else as v itself
__device__ __forceinline__ void round_all(int (&v)[NOUT], int d, int Q)
{
    if (d == 0) return;
    if (d < 0) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = (int)((unsigned)v[o] << (-d));
        return;
    }
    const int mask = (1 << d) - 1, t = 1 << (d - 1);
    switch (Q) {
    case QG_TRN_TCPL:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] >>= d;
        break;
    case QG_TRN_SMGN:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = (v[o] >> d) + ((v[o] < 0) & ((v[o] & mask) != 0));
        break;
    case QG_RND_POS_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = (v[o] >> d) + ((v[o] & mask) >= t);
        break;
    case QG_RND_NEG_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = (v[o] >> d) + ((v[o] & mask) > t);
        break;
    case QG_RND_ZERO:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { const int l = v[o] & mask; v[o] = (v[o] >> d) + ((l > t) | ((l == t) & (v[o] < 0))); }
        break;
    case QG_RND_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { const int l = v[o] & mask; v[o] = (v[o] >> d) + ((l > t) | ((l == t) & (v[o] > 0))); }
        break;
    default: // RND::CONV
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { const int l = v[o] & mask, h = v[o] >> d; v[o] = h + ((l > t) | ((l == t) & (h & 1))); }
        break;
    }
}

__device__ __forceinline__ void overflow_all(int (&v)[NOUT], const QStep& s)
{
    const int lo = (int)s.lo, hi = (int)s.hi;
    switch (s.O) {
    case QG_SAT_TCPL:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = qg_clamp_i32(v[o], lo, hi);
        break;
    case QG_SAT_ZERO: {
        const unsigned span = (unsigned)(hi - lo);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = ((unsigned)(v[o] - lo) > span) ? 0 : v[o];
        break;
    }
    case QG_SAT_SMGN: {
        const int l2 = s.S ? -hi : 0;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = qg_clamp_i32(v[o], l2, hi);
        break;
    }
    default: // WRP::TCPL
        if (s.S) {
            const int sh = 31 - s.W;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) v[o] = (int)((unsigned)v[o] << sh) >> sh;
        } else {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) v[o] &= hi;
        }
        break;
    }
}

__device__ __forceinline__ void step_all(int (&v)[NOUT], const QStep& s)
{
    if (s.identity) return;
    round_all(v, s.d, s.Q);
    overflow_all(v, s);
}

// node of level l: both children have the level's input format (no alignment shift for real GEMMs)
template <int MODE>
__device__ __forceinline__ void node_fixed(int (&v)[NOUT], const int (&x)[NOUT], int flo, int fhi, int bias, unsigned span)
{
    if (MODE == 1) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const int t = x[o] + v[o] - bias;               // biased sum (v_add3_u32)
            v[o] = ((unsigned)t > span) ? bias : t;          // out of range -> biased zero
        }
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = qg_clamp_i32(x[o] + v[o], flo, fhi);
    }
}

template <int FORM>   // 3: every step clamps; 4: values biased by -lo, the record's overflow kind decides; 5: the same on unbiased values
__device__ __forceinline__ void node_fx_rec(int (&v)[NOUT], const int (&x)[NOUT], const QFix& f)
{
#pragma unroll
    for (int o = 0; o < NOUT; ++o) v[o] = x[o] + v[o] + f.t;   // v_add3_u32 (BIASED: t also carries the change of bias)
    if (f.ls) {   // (wave-uniform, rare: the level type has MORE fraction bits than its operands)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = (int)((unsigned)v[o] << f.ls) + (FORM == 4 ? f.lo : 0);
    }
    if (FORM == 4) fx_finish_biased<NOUT>(v, f);
    else if (FORM == 5) fx_finish_any<NOUT>(v, f);
    else fx_finish<NOUT>(v, f);
}
template <int FORM>
__device__ __forceinline__ void node_fx(int (&v)[NOUT], const int (&x)[NOUT], const QTreeTable* __restrict__ t, int l)
{
    node_fx_rec<FORM>(v, x, fx_at(t, FX_OFF_ADD(0, l)));
}

__device__ __forceinline__ void node_all(int (&v)[NOUT], const int (&x)[NOUT], const QTreeTable* __restrict__ t, int l)
{
#pragma unroll
    for (int o = 0; o < NOUT; ++o) v[o] = x[o] + v[o];
    step_all(v, t->level_add[0][l].q);
    step_all(v, t->level_cvt[0][l]);
}

// SPLIT leaf rounding from (h, l): x = h*2^s + l, 0 <= l < 2^s
__device__ __forceinline__ void round_split_all(int (&h)[NOUT], const int (&l)[NOUT], int s, int Q)
{
    const int t = 1 << (s - 1);
    switch (Q) {
    case QG_TRN_TCPL: break;
    case QG_TRN_SMGN:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (h[o] < 0) & (l[o] != 0);
        break;
    case QG_RND_POS_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (l[o] >= t);
        break;
    case QG_RND_NEG_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (l[o] > t);
        break;
    case QG_RND_ZERO:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (l[o] > t) | ((l[o] == t) & (h[o] < 0));
        break;
    case QG_RND_INF:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (l[o] > t) | ((l[o] == t) & (h[o] >= 0)); // x > 0 <=> h > 0, or h == 0 with l == t > 0
        break;
    default:
#pragma unroll
        for (int o = 0; o < NOUT; ++o) h[o] += (l[o] > t) | ((l[o] == t) & (h[o] & 1));
        break;
    }
}

struct QTreeFastArgs {
    const QTreeTable* tab;
    const int32_t* A;   // [M][K]
    const int32_t* B;   // [N][K]
    char* C;            // [M][N] containers
    int64_t M, N, K;
    int32_t cbytes;
    int32_t split_s;    // SPLIT: rounding shift of the product (= bits split off B)
};

// MODE 0: per-node formats and modes are runtime values (any real descriptor the planner admits).
// MODE 1 / 2: the product and every tree level share ONE format, rounding is TRN::TCPL and the overflow mode is
//   SAT::ZERO (1) or SAT::TCPL (2) — the default-tag shapes (configurations 1 and 3 as literally configured).  Then
//   a node is 3 (ZERO, on values biased by -lo so that the range test is one unsigned compare) or 2 (TCPL:
//   v_add + v_med3) VALU instructions and the leaf needs no separate rounding step.
// MODE 3 / 4: per-level formats, every step in the compact form of qg_fix.h (QAnalysis::fast_mode): the node's record is one
//   scalar load, the node itself v_add3 (+ the rounding addend), a shift where the level has fewer fraction bits, and one
//   v_med3 (3: every step of the descriptor clamps), or — 4: SAT::ZERO / WRP::TCPL steps exist — values biased by -lo of their
//   format and the overflow by the record's kind: one unsigned compare + select, med3(u, 0, span), or u & span (qg_fix.h);
//   5: the same kinds on unbiased values (a subtraction more per range test) where a format is too wide for the biased form.
// MODE 6: one signed SAT::TCPL format for the product and every level, held LEFT-JUSTIFIED (qg_fix.h): the product is one
//   saturating v_mad_i32_i24 (operands staged with the factors that justify it, rounding addend included) + v_and, a node one
//   saturating v_add_i32, + v_and at the odd levels: 3.3 instead of 5 vector instructions per MAC, and no split product
//   (the hardware saturates from the full 48-bit product).
#define NODE(X, L)                                                         \
    do {                                                                   \
        if (MODE == 21) {                                                  \
            _Pragma("unroll") for (int o_ = 0; o_ < NOUT; ++o_) v[o_] = (int)((unsigned)X[o_] + (unsigned)v[o_]);   \
        } else if (MODE >= 17 && MODE <= 20) {                             \
            _Pragma("unroll") for (int o_ = 0; o_ < NOUT; ++o_) v[o_] = sat_add(X[o_], v[o_]);   \
            if ((MODE == 19 || MODE == 20) && ((L) & 1)) { _Pragma("unroll") for (int o_ = 0; o_ < NOUT; ++o_) v[o_] &= w_mask; }   \
        } else if (MODE == 6 || MODE == 16) {                              \
            _Pragma("unroll") for (int o_ = 0; o_ < NOUT; ++o_) v[o_] = MODE == 16 ? usat_add(X[o_], v[o_]) : sat_add(X[o_], v[o_]);   \
            if ((L) & 1) { _Pragma("unroll") for (int o_ = 0; o_ < NOUT; ++o_) v[o_] &= lj_mask; }               \
        } else if (MODE >= 3) { if ((L) < 4) node_fx_rec<MODE>(v, X, flow[(L) < 4 ? (L) : 0]); else node_fx<MODE>(v, X, tab, L); } \
        else if (MODE != 0) node_fixed<MODE>(v, X, flo, fhi, bias, span);  \
        else node_all(v, X, tab, L);                                       \
    } while (0)

template <bool SPLIT, bool MUL24, int MAXL, int MODE>
__global__ __launch_bounds__(256) void k_tree_fast(QTreeFastArgs g)
{
    __shared__ __attribute__((aligned(16))) int sA[TMB][PITCH];
    __shared__ __attribute__((aligned(16))) int sBh[TNB][PITCH];
    __shared__ __attribute__((aligned(16))) int sBl[SPLIT ? TNB : 1][PITCH];
    const QTreeTable* __restrict__ tab = g.tab;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    // XCD-aware block order (blocks b and b+8 share an XCD and its L2): every XCD gets a contiguous run of
    // tiles, walked column-major in groups of 16 tile rows, so neighbouring blocks re-use A rows and B columns in L2
    const int64_t tiles_n = (g.N + TNB - 1) / TNB, tiles_m = (g.M + TMB - 1) / TMB;
    int64_t bid = blockIdx.x;
    {
        const int64_t nwg = tiles_m * tiles_n, q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    constexpr int64_t GMT = 16;
    const int64_t grp = bid / (GMT * tiles_n), first_m = grp * GMT;
    const int64_t gsz = (tiles_m - first_m) < GMT ? (tiles_m - first_m) : GMT;
    const int64_t m0 = (first_m + (bid % (GMT * tiles_n)) % gsz) * TMB, n0 = ((bid % (GMT * tiles_n)) / gsz) * TNB;
    const int nl = tab->n_levels_k;   // (a tree shorter than 5 levels is continued with identity levels: qg_plan.h)
    const int s = g.split_s;
    const int smask = SPLIT ? ((1 << s) - 1) : 0;
    const QStep pstep = tab->mul[0].q;
    // MODE 1/2 constants: the shared format's bounds
    const int flo = (int)pstep.lo, fhi = (int)pstep.hi;
    const int bias = MODE == 1 ? -flo : 0;
    const unsigned span = (unsigned)(fhi - flo);
    const int pd = pstep.d > 0 ? pstep.d : 0;  // DIRECT: TCPL shift of the product
    // MODE 3: the product's own compact step (rounding addend, shift, clamp; the upper bound lives in a VGPR for v_med3_i32)
    const QFix fp = tab->fmul[0];
    const int phi_v = MODE >= 3 ? fx_vgpr(fp.hi) : 0;
    // ... and the records of the four lowest levels (15 of 16 nodes) stay in registers; the upper levels load theirs per node
    QFix flow[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) flow[l] = tab->fadd[0][l];
    // MODE 6 (QTreeTable::lj): shift of the justified values, their mask, the product's scaled rounding addend, the operands' factors
    // 21: a WRAPPING 32-bit word (signed WRP::TCPL — what `(int32_t)(((int64_t)a * b) >> 16)` and a plain `+=` compute): the word is
    // v_alignbit of the exact product's halves and a node a plain 32-bit add, nothing is tested or selected
    constexpr bool W32 = MODE >= 17 && MODE <= 21;   // 19 / 20: 17 / 18 on JUSTIFIED words (formats of fewer than 32 bits held as x * 2^sj, qg_plan.cpp)
    constexpr bool WJ = MODE == 19 || MODE == 20, WCMP = MODE == 17 || MODE == 19, WMAD = MODE == 18 || MODE == 20;
    const int w_d = W32 ? tab->lj.s : 0, w_t = W32 ? tab->lj.t[0] : 0;   // the product's shift and rounding addend
    unsigned w_half = WCMP ? 1u << ((w_d - 1) & 31) : 0u, w_lim = WCMP ? 1u << (w_d & 31) : 0u;
    const int w_f = WMAD ? 1 << ((32 - w_d) & 31) : 0;                   // 2^(32 - d), the weight of the product's high half in the word
    const int w_j = WJ ? tab->lj.e[0] : 0, w_mask = WJ ? (int)(~0u << w_j) : -1;   // bits below the unit in a justified word, cleared as MODE 6 clears them
    if (WCMP) asm volatile("" : "+s"(w_lim));   // (opaque: the compiler would rewrite "x < 2^d" as a shift and a compare with 0 — one instruction more per product)
    constexpr bool LJ = MODE == 6 || MODE == 16;   // (16: the unsigned counterpart — uint32 range, v_mad_u32_u24 / v_add_u32 ... clamp)
    const int lj_s = LJ ? tab->lj.s : 0, lj_mask = LJ ? (int)(~0u << lj_s) : -1, lj_t = LJ ? tab->lj.t[0] : 0;
    const int lj_ea = LJ ? tab->lj.e[0] : 0, lj_eb = LJ ? tab->lj.e[1] : 0;

    int low[4][NOUT];
    int up[MAXL - 4][NOUT];
    int v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) v[o] = 0;

    for (int64_t k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();
        // stage A: 64 rows x 32 k = 512 16-byte chunks, 2 per thread; B: 32 rows -> 256 chunks, 1 per thread
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = tid + 256 * c, r = ch >> 3, q = ch & 7;
            int4 x = make_int4(0, 0, 0, 0);
            if (m0 + r < g.M) x = *(const int4*)(g.A + (m0 + r) * g.K + k0 + q * 4);
            if (LJ) x = make_int4(x.x << lj_ea, x.y << lj_ea, x.z << lj_ea, x.w << lj_ea);
            *(int4*)&sA[r][q * 4] = x;
        }
        {
            const int r = tid >> 3, q = tid & 7;
            int4 x = make_int4(0, 0, 0, 0);
            if (n0 + r < g.N) x = *(const int4*)(g.B + (n0 + r) * g.K + k0 + q * 4);
            if (LJ) x = make_int4(x.x << lj_eb, x.y << lj_eb, x.z << lj_eb, x.w << lj_eb);
            if (SPLIT) {
                *(int4*)&sBh[r][q * 4] = make_int4(x.x >> s, x.y >> s, x.z >> s, x.w >> s);
                *(int4*)&sBl[r][q * 4] = make_int4(x.x & smask, x.y & smask, x.z & smask, x.w & smask);
            } else {
                *(int4*)&sBh[r][q * 4] = x;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; ++kb) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                int4 a4[4], bh4[2], bl4[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[i] = *(const int4*)&sA[ty * 4 + i][kb * 16 + kq * 4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // a lane's two columns are tx and tx + 16: the 16 lanes of a ds_read_b128 group then read 16 CONSECUTIVE rows,
                    // 36 dwords apart — 16 distinct 4-bank slots.  (Columns 2 tx + j, 72 dwords apart, put tx and tx + 8 on the
                    // same banks: SQ_LDS_BANK_CONFLICT = 4096^3 / 128 per launch in profiles/r03d_c3T.json.)
                    bh4[j] = *(const int4*)&sBh[tx + 16 * j][kb * 16 + kq * 4];
                    if (SPLIT) bl4[j] = *(const int4*)&sBl[tx + 16 * j][kb * 16 + kq * 4];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = kq * 4 + e; // compile-time leaf index inside the 16-leaf block
                    int av[4], bhv[2], blv[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[i] = e == 0 ? a4[i].x : e == 1 ? a4[i].y : e == 2 ? a4[i].z : a4[i].w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bhv[j] = e == 0 ? bh4[j].x : e == 1 ? bh4[j].y : e == 2 ? bh4[j].z : bh4[j].w;
                        if (SPLIT) blv[j] = e == 0 ? bl4[j].x : e == 1 ? bl4[j].y : e == 2 ? bl4[j].z : bl4[j].w;
                    }
                    // ---- leaves: 8 quantised products
                    if (MODE == 21) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const long long p = (long long)av[i] * (long long)bhv[j] + (long long)w_t;
                                v[i * 2 + j] = (int)__builtin_amdgcn_alignbit((unsigned)(p >> 32), (unsigned)p, (unsigned)w_d);
                            }
                    } else if (WMAD) {
                        // 32-bit words with a product shift of 10 ... 23 (Q15.16: 16): floor(p / 2^d) = hi * 2^(32-d) + (lo >> d), and the clamp
                        // bit of v_mad_i32_i24 saturates exactly that sum to the word (the hardware clamps the full-width result,
                        // tools/ubench/sat_semantics.hip).  hi enters as a 24-bit factor: every in-range hi is within 2^(d-1) <= 2^22, and one
                        // clamped to +-2^23 still carries the sum past the word's range (d <= 23) — 4 instructions per product where the
                        // range test and select of MODE 17 take 7
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const long long p = (long long)av[i] * (long long)bhv[j] + (long long)w_t;
                                int ph = (int)(p >> 32);
                                asm("v_med3_i32 %0, %0, %1, %2" : "+v"(ph) : "s"(-(1 << 23)), "v"((1 << 23) - 1));
                                v[i * 2 + j] = sat_mad24_vsv(ph, w_f, (int)((unsigned)p >> w_d)) & w_mask;   // (w_mask: -1 unless justified — folded away)
                            }
                    } else if (WCMP) {   // 32-bit words (fast_mode 10): floor((a b + t) / 2^d) of the exact 64-bit product, saturated to the word
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                // all in 32-bit instructions behind the one v_mad_i64_i32 (the compiler turns a 64-bit shift + clamp into
                                // v_ashrrev_i64 and two v_cmp_*_i64): the word is v_alignbit of the product's halves, and it is in range
                                // iff the high half lies in [-2^(d-1), 2^(d-1)) — one add, one unsigned compare (1 <= d <= 31, qg_plan.cpp; a wave-uniform branch for d = 0 cost 35 % at 2048^3)
                                const long long p = (long long)av[i] * (long long)bhv[j] + (long long)w_t;
                                const int ph = (int)(p >> 32);
                                const int ql = (int)__builtin_amdgcn_alignbit((unsigned)ph, (unsigned)p, (unsigned)w_d);
                                v[i * 2 + j] = ((unsigned)ph + w_half < w_lim ? ql : ((ph >> 31) ^ 0x7fffffff)) & w_mask;
                            }
                    } else if (LJ) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) v[i * 2 + j] = (MODE == 16 ? usat_mad24_vvs(av[i], bhv[j], lj_t) : sat_mad24_vvs(av[i], bhv[j], lj_t)) & lj_mask;
                    } else if (MODE == 5 && fp.skip != 0) {
                        // a product whose rounding looks at the value's sign or parity (RND::ZERO / INF / CONV, TRN::SMGN)
                        if (SPLIT) {
                            int lw[NOUT];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int t = __mul24(av[i], blv[j]);
                                    v[i * 2 + j] = __mul24(av[i], bhv[j]) + (t >> s);
                                    lw[i * 2 + j] = t & smask;
                                }
                            round_split_all(v, lw, s, fp.skip == 1 ? QG_RND_ZERO : fp.skip == 2 ? QG_RND_INF : fp.skip == 3 ? QG_RND_CONV : QG_TRN_SMGN);
                            QFix f0 = fp;
                            f0.d = 0;   // (rounded)
                            fx_finish_any<NOUT>(v, f0);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) v[i * 2 + j] = __mul24(av[i], bhv[j]);
                            fx_finish_any<NOUT>(v, fp);   // (d > 0: the rounding kind, the shift, the overflow kind)
                        }
                    } else if (MODE >= 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                if (SPLIT) {
                                    // floor((a*b + t) / 2^s) = a*bh + ((a*bl + t) >> s): the rounding addend rides in the low half
                                    const int t = mad24_vvs(av[i], blv[j], fp.t);
                                    v[i * 2 + j] = __mul24(av[i], bhv[j]) + (t >> s);
                                } else {
                                    v[i * 2 + j] = mad24_vvs(av[i], bhv[j], fp.t);
                                }
                            }
                        if (!SPLIT) {
                            if (fp.d) {
#pragma unroll
                                for (int o = 0; o < NOUT; ++o) v[o] >>= fp.d;
                            } else if (fp.ls) {
#pragma unroll
                                for (int o = 0; o < NOUT; ++o) v[o] = (int)((unsigned)v[o] << fp.ls) + (MODE == 4 ? fp.lo : 0);
                            }
                        }
                        if (MODE == 3) {
#pragma unroll
                            for (int o = 0; o < NOUT; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(v[o]) : "s"(fp.lo), "v"(phi_v));
                        } else {
                            QFix f0 = fp;
                            f0.d = 0;   // (the shift is done)
                            f0.skip = 0;
                            if (MODE == 4) fx_finish_biased<NOUT>(v, f0);
                            else fx_finish_any<NOUT>(v, f0);
                        }
                    } else if (MODE != 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                int p;
                                if (SPLIT) {
                                    // floor(a*b / 2^s) + bias = a*bh + ((a*bl + bias*2^s) >> s)
                                    const int t = __mul24(av[i], blv[j]) + (bias << s);
                                    p = __mul24(av[i], bhv[j]) + (t >> s);
                                } else {
                                    p = (__mul24(av[i], bhv[j]) + (bias << pd)) >> pd;
                                }
                                if (MODE == 1) p = ((unsigned)p > span) ? bias : p;
                                else p = qg_clamp_i32(p, flo, fhi);
                                v[i * 2 + j] = p;
                            }
                    } else if (SPLIT) {
                        int lw[NOUT];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int t = MUL24 ? __mul24(av[i], blv[j]) : av[i] * blv[j];
                                v[i * 2 + j] = (MUL24 ? __mul24(av[i], bhv[j]) : av[i] * bhv[j]) + (t >> s);
                                lw[i * 2 + j] = t & smask;
                            }
                        round_split_all(v, lw, s, pstep.Q);
                        overflow_all(v, pstep);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) v[i * 2 + j] = MUL24 ? __mul24(av[i], bhv[j]) : av[i] * bhv[j];
                        round_all(v, pstep.d, pstep.Q);
                        overflow_all(v, pstep);
                    }
                    // ---- lower four levels: binary counter on the compile-time index kk
                    if ((kk & 1) == 0) {
#pragma unroll
                        for (int o = 0; o < NOUT; ++o) low[0][o] = v[o];
                    } else {
                        NODE(low[0], 0);
                        if ((kk & 2) == 0) {
#pragma unroll
                            for (int o = 0; o < NOUT; ++o) low[1][o] = v[o];
                        } else {
                            NODE(low[1], 1);
                            if ((kk & 4) == 0) {
#pragma unroll
                                for (int o = 0; o < NOUT; ++o) low[2][o] = v[o];
                            } else {
                                NODE(low[2], 2);
                                if ((kk & 8) == 0) {
#pragma unroll
                                    for (int o = 0; o < NOUT; ++o) low[3][o] = v[o];
                                } else {
                                    NODE(low[3], 3);
                                }
                            }
                        }
                    }
                }
            }
            // ---- upper levels: v is the partial sum of one 16-leaf block (an element of list 4)
            const unsigned idx = (unsigned)((k0 >> 4) + kb);
            bool parked = false; // wave-uniform: the carry stopped at a free slot
#pragma unroll
            for (int u = 0; u < MAXL - 4; ++u) {
                if (!parked && 4 + u < nl) {
                    if (((idx >> u) & 1u) == 0) {
#pragma unroll
                        for (int o = 0; o < NOUT; ++o) up[u][o] = v[o];
                        parked = true;
                    } else {
                        NODE(up[u], 4 + u);
                    }
                }
            }
        }
    }
    // after the last block the counter has carried through every level: v holds the root
    if (MODE == 1) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] -= bias;
    }
    if (MODE == 4) {   // the root's bias
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] -= fp.ka;
    }
    if (LJ) {   // floor(v / 2^s): the value
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] = MODE == 16 ? (int)((unsigned)v[o] >> lj_s) : v[o] >> lj_s;
    }
    if (WJ) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) v[o] >>= w_j;
    }
    step_all(v, tab->c_cvt[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + ty * 4 + i, n = n0 + tx + 16 * j;
            if (m < g.M && n < g.N) {
                const int64_t idx = m * g.N + n;
                const int r = v[i * 2 + j];
                switch (g.cbytes) {
                case 1: ((int8_t*)g.C)[idx] = (int8_t)r; break;
                case 2: ((int16_t*)g.C)[idx] = (int16_t)r; break;
                case 4: ((int32_t*)g.C)[idx] = r; break;
                default: ((int64_t*)g.C)[idx] = (int64_t)r; break;
                }
            }
        }
}

// ---- packed 16-bit form (fast_mode 7): the left-justified form of MODE 6 for formats of at most 16 bits, TWO outputs per
// register.  A lane owns 4 rows x 2 columns (tx, tx + 16) as 4 registers (low half: column tx); a product pair is ONE
// v_pk_mad_i16 ... clamp (the hardware saturates a * b + t from the exact product, tools/ubench/sat_semantics.hip) whose A
// operand is broadcast to both halves by op_sel (A is staged as 16-bit values, two k per dword; B as (column tx, column
// tx + 16) pairs per k), + v_and; a node pair is one v_pk_add_i16 ... clamp, + v_and at the odd levels: 1.9 vector
// instructions per MAC where MODE 6 spends 3.8 and the v_med3 form 5.3 (BASELINE configuration 2 as literally configured).
constexpr int PKP = 18;   // dwords per sA16 row (16 + 2: 8-byte reads stay aligned)
// HYB (fast_mode 8): formats of 9 ... 16 bits whose PRODUCT does not fit the halves (format bits + the product's rounding shift
// > 16: int<7,8>, the 16-bit words of most fixed-point code).  The product is MODE 6's — one saturating v_mad_i32_i24 per output
// on x * 2^(32 - bits) — and its HIGH half is the value justified in 16 bits: one v_perm_b32 packs the high halves of a lane's two
// columns and drops the low ones — for a 16-bit format that IS the rounding's floor and no low bits exist to be cleared anywhere
// (2.0 vector instructions per MAC); narrower formats clear the rest of the fraction with one v_and per pair and then follow the
// packed form's rule (2.7 per MAC) — where MODE 6 spends 3.5.
#define NODE16(X, L)                                                                                     \
    do {                                                                                                 \
        _Pragma("unroll") for (int o_ = 0; o_ < 4; ++o_) v[o_] = UNS ? pk_add_usat(X[o_], v[o_]) : pk_add_sat(X[o_], v[o_]);   \
        if (((L) & 1) && HYB != 2) { _Pragma("unroll") for (int o_ = 0; o_ < 4; ++o_) v[o_] &= mask2; }       \
    } while (0)

template <int MAXL, int HYB = 0, bool UNS = false>   // UNS: unsigned formats (qg_fix.h); HYB 1: formats of fewer than 16 bits; 2: exactly 16 (no bits below the unit in a half: nothing to clear, ever)
__global__ __launch_bounds__(256) void k_tree_pk16(QTreeFastArgs g)
{
    __shared__ __attribute__((aligned(16))) int sA[TMB][HYB ? PITCH : PKP];              // [row][k / 2]: (k even, k odd); HYB: [row][k] of 32-bit values
    __shared__ __attribute__((aligned(16))) int sB[HYB ? TNB : TNB / 2][PITCH];          // [column pair][k]: (column p, column p + 16); HYB: [column][k]
    const QTreeTable* __restrict__ tab = g.tab;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t tiles_n = (g.N + TNB - 1) / TNB, tiles_m = (g.M + TMB - 1) / TMB;
    int64_t bid = blockIdx.x;
    {   // XCD-aware block order, as k_tree_fast
        const int64_t nwg = tiles_m * tiles_n, q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    constexpr int64_t GMT = 16;
    const int64_t grp = bid / (GMT * tiles_n), first_m = grp * GMT;
    const int64_t gsz = (tiles_m - first_m) < GMT ? (tiles_m - first_m) : GMT;
    const int64_t m0 = (first_m + (bid % (GMT * tiles_n)) % gsz) * TMB, n0 = ((bid % (GMT * tiles_n)) / gsz) * TNB;
    const int nl = tab->n_levels_k;
    const int s16 = HYB ? tab->lj.s - 16 : tab->lj16.s, ea = HYB ? tab->lj.e[0] : tab->lj16.e[0], eb = HYB ? tab->lj.e[1] : tab->lj16.e[1];
    const int m1 = (0xffff << s16) & 0xffff, mask2 = m1 | (m1 << 16);
    const int t2 = HYB ? tab->lj.t[0] : pk2(tab->lj16.t[0], tab->lj16.t[0]);   // (HYB: the 32-bit product's scaled addend)

    int low[4][4];
    int up[MAXL - 4][4];
    int v[4] = {0, 0, 0, 0};

    // the next k-chunk's operands travel in registers while the current one is computed: a small problem (configuration 2: two
    // workgroups per CU) has no other waves to cover the load latency with
    int4 ra[2], rb[2];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = tid + 256 * c, r = ch >> 3, q = ch & 7;
            ra[c] = make_int4(0, 0, 0, 0);
            if (m0 + r < g.M) ra[c] = *(const int4*)(g.A + (m0 + r) * g.K + k0 + q * 4);
        }
        rb[0] = rb[1] = make_int4(0, 0, 0, 0);
        if (HYB) {   // 32 columns x 8 chunks of 4 k: one per thread
            const int r = tid >> 3, q = tid & 7;
            if (n0 + r < g.N) rb[0] = *(const int4*)(g.B + (n0 + r) * g.K + k0 + q * 4);
        } else if (tid < 128) {
            const int p = tid >> 3, q = tid & 7;
            if (n0 + p < g.N) rb[0] = *(const int4*)(g.B + (n0 + p) * g.K + k0 + q * 4);
            if (n0 + p + 16 < g.N) rb[1] = *(const int4*)(g.B + (n0 + p + 16) * g.K + k0 + q * 4);
        }
    };
    fetch(0);
    for (int64_t k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();
        // stage A: 64 rows x 32 k, 2 chunks of 4 k per thread, as 16-bit values; B: 16 column pairs x 32 k, threads 0..127
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = tid + 256 * c, r = ch >> 3, q = ch & 7;
            const int4 x = ra[c];
            if (HYB) *(int4*)&sA[r][q * 4] = make_int4(x.x << ea, x.y << ea, x.z << ea, x.w << ea);
            else *(int2*)&sA[r][q * 2] = make_int2(pk2(x.x << ea, x.y << ea), pk2(x.z << ea, x.w << ea));
        }
        if (HYB) {
            const int r = tid >> 3, q = tid & 7;
            const int4 x = rb[0];
            *(int4*)&sB[r][q * 4] = make_int4(x.x << eb, x.y << eb, x.z << eb, x.w << eb);
        } else if (tid < 128) {
            const int p = tid >> 3, q = tid & 7;
            const int4 x = rb[0], y = rb[1];
            *(int4*)&sB[p][q * 4] = make_int4(pk2(x.x << eb, y.x << eb), pk2(x.y << eb, y.y << eb), pk2(x.z << eb, y.z << eb), pk2(x.w << eb, y.w << eb));
        }
        if (k0 + KC < g.K) fetch(k0 + KC);
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; ++kb) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                int2 a2[4];
                int4 a4[4], b4h[2];   // HYB: 32-bit operands, four k per read
                int4 b4 = make_int4(0, 0, 0, 0);
                if (HYB) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = *(const int4*)&sA[ty * 4 + i][kb * 16 + kq * 4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b4h[j] = *(const int4*)&sB[tx + 16 * j][kb * 16 + kq * 4];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a2[i] = *(const int2*)&sA[ty * 4 + i][kb * 8 + kq * 2];
                    b4 = *(const int4*)&sB[tx][kb * 16 + kq * 4];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = kq * 4 + e;
                    const int bv = e == 0 ? b4.x : e == 1 ? b4.y : e == 2 ? b4.z : b4.w;
                    if (HYB) {
                        const int b0 = e == 0 ? b4h[0].x : e == 1 ? b4h[0].y : e == 2 ? b4h[0].z : b4h[0].w;
                        const int b1 = e == 0 ? b4h[1].x : e == 1 ? b4h[1].y : e == 2 ? b4h[1].z : b4h[1].w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int av = e == 0 ? a4[i].x : e == 1 ? a4[i].y : e == 2 ? a4[i].z : a4[i].w;
                            // the high halves of the two justified products: (column tx, column tx + 16); the dropped halves are the floor
                            v[i] = UNS ? (int)__builtin_amdgcn_perm((unsigned)usat_mad24_vvs(av, b1, t2), (unsigned)usat_mad24_vvs(av, b0, t2), 0x07060302u)
                                       : (int)__builtin_amdgcn_perm((unsigned)sat_mad24_vvs(av, b1, t2), (unsigned)sat_mad24_vvs(av, b0, t2), 0x07060302u);
                            if (HYB == 1) v[i] &= mask2;   // (fewer than 16 bits: the rest of the fraction sits in the halves' low bits)
                        }
                    } else
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int av = e < 2 ? a2[i].x : a2[i].y;
                        v[i] = (UNS ? ((e & 1) ? pk_mad_usat<1>(av, bv, t2) : pk_mad_usat<0>(av, bv, t2)) : ((e & 1) ? pk_mad_sat<1>(av, bv, t2) : pk_mad_sat<0>(av, bv, t2))) & mask2;
                    }
                    if ((kk & 1) == 0) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) low[0][o] = v[o];
                    } else {
                        NODE16(low[0], 0);
                        if ((kk & 2) == 0) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) low[1][o] = v[o];
                        } else {
                            NODE16(low[1], 1);
                            if ((kk & 4) == 0) {
#pragma unroll
                                for (int o = 0; o < 4; ++o) low[2][o] = v[o];
                            } else {
                                NODE16(low[2], 2);
                                if ((kk & 8) == 0) {
#pragma unroll
                                    for (int o = 0; o < 4; ++o) low[3][o] = v[o];
                                } else {
                                    NODE16(low[3], 3);
                                }
                            }
                        }
                    }
                }
            }
            const unsigned idx = (unsigned)((k0 >> 4) + kb);
            bool parked = false;
#pragma unroll
            for (int u = 0; u < MAXL - 4; ++u) {
                if (!parked && 4 + u < nl) {
                    if (((idx >> u) & 1u) == 0) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) up[u][o] = v[o];
                        parked = true;
                    } else {
                        NODE16(up[u], 4 + u);
                    }
                }
            }
        }
    }
    // the root: floor(half / 2^s) of each half, then the conversion into C
    int r8[NOUT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r8[i * 2 + 0] = UNS ? (int)(((unsigned)v[i] & 0xffffu) >> s16) : ((int)((unsigned)v[i] << 16) >> 16) >> s16;
        r8[i * 2 + 1] = UNS ? (int)(((unsigned)v[i] >> 16) >> s16) : (v[i] >> 16) >> s16;
    }
    step_all(r8, tab->c_cvt[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + ty * 4 + i, n = n0 + tx + 16 * j;
            if (m < g.M && n < g.N) {
                const int64_t idx = m * g.N + n;
                const int r = r8[i * 2 + j];
                switch (g.cbytes) {
                case 1: ((int8_t*)g.C)[idx] = (int8_t)r; break;
                case 2: ((int16_t*)g.C)[idx] = (int16_t)r; break;
                case 4: ((int32_t*)g.C)[idx] = r; break;
                default: ((int64_t*)g.C)[idx] = (int64_t)r; break;
                }
            }
        }
}

} // namespace

template <bool SPLIT, bool MUL24, int MODE>
static void launch_tf(int n_levels, dim3 grid, hipStream_t st, const QTreeFastArgs& g)
{
    if (n_levels <= 12) hipLaunchKernelGGL((k_tree_fast<SPLIT, MUL24, 12, MODE>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_tree_fast<SPLIT, MUL24, 16, MODE>), grid, dim3(256), 0, st, g);
}

hipError_t qg_launch_tree_fast(const QTreeTable* dev_table, int n_levels, int split_s, int mul24, int mode, const void* A, const void* B,
                               void* C, int64_t M, int64_t N, int64_t K, int cbytes, hipStream_t st)
{
    if (K % KC != 0 || n_levels < 5 || n_levels > 16) return hipErrorInvalidValue;
    QTreeFastArgs g{dev_table, (const int32_t*)A, (const int32_t*)B, (char*)C, M, N, K, cbytes, split_s};
    const int64_t blocks = ((M + TMB - 1) / TMB) * ((N + TNB - 1) / TNB);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks);
    if (mode == 10) {   // 32-bit words: exact 64-bit products, saturating adds (qg_plan.cpp, fast_mode 10)
        launch_tf<false, false, 17>(n_levels, grid, st, g);
        return hipGetLastError();
    }
    if (mode == 14) {   // a wrapping 32-bit word (signed WRP::TCPL)
        launch_tf<false, false, 21>(n_levels, grid, st, g);
        return hipGetLastError();
    }
    if (mode == 12 || mode == 13) {   // ... on justified words (formats of fewer than 32 bits): 12 the compare form, 13 the multiply-add form
        if (mode == 12) launch_tf<false, false, 19>(n_levels, grid, st, g);
        else launch_tf<false, false, 20>(n_levels, grid, st, g);
        return hipGetLastError();
    }
    if (mode == 11) {   // ... with a product shift of 10 ... 23 (qg_api.hip reads it from the table): the product's word from one saturating multiply-add
        launch_tf<false, false, 18>(n_levels, grid, st, g);
        return hipGetLastError();
    }
    if (mode != 0 && !mul24) mode = 0;  // the fixed-mode variants are built for 24-bit multiplies only
    const bool uns = mode >= 16;   // (+ 16: the unsigned counterparts)
    if (uns) mode -= 16;
    if (mode >= 7 && mode <= 9) {   // packed 16-bit halves (7); 32-bit justified products, packed 16-bit nodes (8: fewer than 16 bits; 9: exactly 16)
#define QG_PK16_LAUNCH(H, U)                                                                                   \
        do {                                                                                                       \
            if (n_levels <= 12) hipLaunchKernelGGL((k_tree_pk16<12, H, U>), grid, dim3(256), 0, st, g);            \
            else hipLaunchKernelGGL((k_tree_pk16<16, H, U>), grid, dim3(256), 0, st, g);                           \
        } while (0)
        if (uns) { if (mode == 7) QG_PK16_LAUNCH(0, true); else if (mode == 8) QG_PK16_LAUNCH(1, true); else QG_PK16_LAUNCH(2, true); }
        else { if (mode == 7) QG_PK16_LAUNCH(0, false); else if (mode == 8) QG_PK16_LAUNCH(1, false); else QG_PK16_LAUNCH(2, false); }
#undef QG_PK16_LAUNCH
        return hipGetLastError();
    }
    if (mode == 6) {   // left-justified saturating form: never split (the planner checked the scaled operands against 24 bits)
        if (uns) launch_tf<false, true, 16>(n_levels, grid, st, g);
        else launch_tf<false, true, 6>(n_levels, grid, st, g);
        return hipGetLastError();
    }
    if (split_s > 0) {
        if (mode == 1) launch_tf<true, true, 1>(n_levels, grid, st, g);
        else if (mode == 2) launch_tf<true, true, 2>(n_levels, grid, st, g);
        else if (mode == 3) launch_tf<true, true, 3>(n_levels, grid, st, g);
        else if (mode == 4) launch_tf<true, true, 4>(n_levels, grid, st, g);
        else if (mode == 5) launch_tf<true, true, 5>(n_levels, grid, st, g);
        else if (mul24) launch_tf<true, true, 0>(n_levels, grid, st, g);
        else launch_tf<true, false, 0>(n_levels, grid, st, g);
    } else {
        if (mode == 1) launch_tf<false, true, 1>(n_levels, grid, st, g);
        else if (mode == 2) launch_tf<false, true, 2>(n_levels, grid, st, g);
        else if (mode == 3) launch_tf<false, true, 3>(n_levels, grid, st, g);
        else if (mode == 4) launch_tf<false, true, 4>(n_levels, grid, st, g);
        else if (mode == 5) launch_tf<false, true, 5>(n_levels, grid, st, g);
        else if (mul24) launch_tf<false, true, 0>(n_levels, grid, st, g);
        else launch_tf<false, false, 0>(n_levels, grid, st, g);
    }
    return hipGetLastError();
}
