// qg_tree_cplx.hip — exact tree-order evaluation for COMPLEX operands, 32-bit VALU path
// (BASELINE.json configuration 5: Qcomplex<int<6,3>,int<6,-3>>, TFComplexMul, RND + SAT).
//
// Each complex product runs the reference's chain of individually quantised real operations —
// TFComplexMul: A=(a+b)c, B=(c+d)b, C=(b-a)d, re=A-B, im=B-C (/root/reference/include/QuBLAS.h:3524-3529, with B
// quantised by badT, C by cdbT and (b-a) in the default-merged format, exactly as the header does);
// BasicComplexMul: re=ac-bd, im=ad+bc (QuBLAS.h:3439-3440) — and the pairwise tree reduces the real
// and imaginary parts separately with their own per-level formats (complex Qadd is part-wise,
// QuBLAS.h:3549-3564; the level buffer converts on store, :4966).  The lowering has already resolved
// every sub-operation's format into the plan's QTreeTable; this kernel only applies them.
//
// Mapping: as qg_tree_fast.hip — each lane owns a 2x2 block of complex outputs, the tree is a
// register-resident binary counter (levels 0-3 unrolled over 16 leaves, upper levels statically
// indexed).  The multiplier is a template parameter.  TF's operand-only sums (a+b), (b-a) depend on the
// A element alone and (c+d) on the B element alone: they are formed ONCE per element while the tile is
// staged (LDS planes {a+b, b, b-a} / {c, c+d, d}), not in the k loop.  The step forms (MODE) are described
// above op_addsub; compact step records are fetched with one s_load_dwordx8 each, the records of steps that
// follow each other under one wait (qg_fix.h, fx_at2 ... fx_at5).
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "qg_fix.h"
#include "qg_kernels.h"
#include "qg_step_all.h"

namespace {

constexpr int KC = 32;
constexpr int TMB = 32, TNB = 32;  // 16x16 threads x (2x2)
constexpr int PITCH = KC + 4;

// one quantising step on N values.  FIXED: the planner has shown every step of this descriptor to be the identity or
// "RND::POS_INF by d >= 0 bits, then SAT::TCPL" (QAnalysis::cplx_fixed_ok) — round-half-up is (v + 2^(d-1)) >> d and the
// saturation one clamp, 3 VALU instructions instead of the 6-7 of the runtime-mode form and no scalar branch ladder.
template <bool FIXED, int N>
__device__ __forceinline__ void step_n(int (&v)[N], const QStep& s)
{
    if constexpr (FIXED) {
        if (s.identity) return;
        const int lo = (int)s.lo, hi = (int)s.hi;
        if (s.d >= 0) {
            const int t = (1 << s.d) >> 1;
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = min(max((v[o] + t) >> s.d, lo), hi);
        } else {   // exact left shift (e.g. a product of a fracBits<-3> part brought to fracBits<3>)
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = min(max((int)((unsigned)v[o] << (-s.d)), lo), hi);
        }
    } else {
        qg_step_all<int, N>(v, s);
    }
}

template <bool FIXED, int N>
__device__ __forceinline__ void addsub_n(int (&out)[N], const int (&x)[N], const int (&y)[N], const QNode& n, bool sub)
{
#pragma unroll
    for (int o = 0; o < N; ++o) {
        const int xs = (int)((unsigned)x[o] << n.sa), ys = (int)((unsigned)y[o] << n.sb);
        out[o] = sub ? xs - ys : xs + ys;
    }
    step_n<FIXED, N>(out, n.q);
}

template <bool FIXED, int N>
__device__ __forceinline__ void mul_n(int (&out)[N], const int (&x)[N], const int (&y)[N], const QNode& n)
{
#pragma unroll
    for (int o = 0; o < N; ++o) out[o] = x[o] * y[o];
    step_n<FIXED, N>(out, n.q);
}

// tree node of level l for one part: children share a format, so no alignment shift
template <bool FIXED, int N>
__device__ __forceinline__ void node_n(int (&v)[N], const int (&x)[N], const QTreeTable* __restrict__ t, int part, int l)
{
#pragma unroll
    for (int o = 0; o < N; ++o) v[o] = x[o] + v[o];
    step_n<FIXED, N>(v, t->level_add[part][l].q);
    step_n<FIXED, N>(v, t->level_cvt[part][l]);
}

// ---- MODE 2: the compact fixed-mode steps (qg_fix.h, QFix of qg_plan.h) ----
// KIND 0: every step adds a constant and clamps; 1: the records carry rounding / overflow kinds (fx_finish_packed, a branch
// ladder per step); 8 + FEAT: kinds of the branch-free feature set FEAT (fx_finish_feat; k: the rounding's factor)
template <int KIND, int N>
__device__ __forceinline__ void fx_done(int (&v)[N], const QFix& f, int k)
{
    if constexpr (KIND >= 8) fx_finish_feat<KIND - 8, N>(v, f, k);
    else if constexpr (KIND == 1) fx_finish_packed<N>(v, f);
    else fx_finish<N>(v, f);
}
template <int KIND, int N>
__device__ __forceinline__ void fx_addsub(int (&out)[N], const int (&x)[N], const int (&y)[N], const QFix& f, bool sub)
{
    const int kb = sub ? -f.kb : f.kb;
    const int tv = fx_vgpr(f.t);
#pragma unroll
    for (int o = 0; o < N; ++o) out[o] = mad24_vsv(y[o], kb, mad24_vsv(x[o], f.ka, tv));
    fx_done<KIND, N>(out, f, f.ls);
}
template <int KIND, int N>
__device__ __forceinline__ void fx_mul(int (&out)[N], const int (&x)[N], const int (&y)[N], const QFix& f)
{
    if (f.ka != 1) {   // an exact product that is brought to MORE fraction bits: scale one factor (wave-uniform, rare)
#pragma unroll
        for (int o = 0; o < N; ++o) out[o] = mad24_vvs(__mul24(x[o], f.ka), y[o], f.t);
    } else {
#pragma unroll
        for (int o = 0; o < N; ++o) out[o] = mad24_vvs(x[o], y[o], f.t);
    }
    fx_done<KIND, N>(out, f, f.ls);
}
template <int KIND, int N>
__device__ __forceinline__ void fx_node(int (&v)[N], const int (&x)[N], const QFix& fa, const QFix& fc)
{
#pragma unroll
    for (int o = 0; o < N; ++o) v[o] = x[o] + v[o] + fa.t;   // v_add3_u32
    if (fa.ls) {      // (wave-uniform, rare: the add's result type has more fraction bits than its operands)
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (int)((unsigned)v[o] << fa.ls);
    }
    fx_done<KIND, N>(v, fa, fa.ka);
    if (!(fc.skip & 1)) {   // the level buffer's conversion (identity unless the level type differs from the add's result)
        if (fc.ls) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (int)((unsigned)v[o] << fc.ls);
        } else {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] += fc.t;
        }
        fx_done<KIND, N>(v, fc, fc.ka);
    }
}

// MODE: 0 run-time modes, 1 fixed modes read from the step table, 2 fixed modes in the compact branch-free form, 4 the same with
// ONE clamp range for the whole k loop held in registers (k_tree_cplx), 3 the compact
// form with rounding / overflow kinds (SAT::ZERO, WRP::TCPL, RND::ZERO / INF / CONV, TRN::SMGN) behind a branch per step,
// 8 + FEAT the same without the branch for the kinds of FEAT (qg_fix.h, fx_finish_feat)
template <int MODE, int N>
__device__ __forceinline__ void op_addsub(int (&out)[N], const int (&x)[N], const int (&y)[N], const QTreeTable* __restrict__ t, int slot, bool sub)
{
    if constexpr (MODE >= 2) fx_addsub<(MODE >= 8 ? MODE : MODE == 3), N>(out, x, y, fx_at(t, FX_OFF_MUL(slot)), sub);
    else addsub_n<MODE == 1, N>(out, x, y, t->mul[slot], sub);
}
template <int MODE, int N>
__device__ __forceinline__ void op_mul(int (&out)[N], const int (&x)[N], const int (&y)[N], const QTreeTable* __restrict__ t, int slot)
{
    if constexpr (MODE >= 2) fx_mul<(MODE >= 8 ? MODE : MODE == 3), N>(out, x, y, fx_at(t, FX_OFF_MUL(slot)));
    else mul_n<MODE == 1, N>(out, x, y, t->mul[slot]);
}
template <int MODE, int N>
__device__ __forceinline__ void op_node(int (&v)[N], const int (&x)[N], const QTreeTable* __restrict__ t, int part, int l)
{
    if constexpr (MODE >= 2) {
        QFix fa, fc;
        fx_at2(t, FX_OFF_ADD(part, l), FX_OFF_CVT(part, l), fa, fc);
        fx_node<(MODE >= 8 ? MODE : MODE == 3), N>(v, x, fa, fc);
    } else {
        node_n<MODE == 1, N>(v, x, t, part, l);
    }
}

// level l's node of BOTH parts: their four records in one wait
template <int MODE, int N>
__device__ __forceinline__ void op_node2(int (&v0)[N], int (&v1)[N], const int (&x0)[N], const int (&x1)[N], const QTreeTable* __restrict__ t, int l)
{
    if constexpr (MODE >= 2) {
        QFix fa0, fc0, fa1, fc1;
        fx_at4(t, FX_OFF_ADD(0, l), FX_OFF_CVT(0, l), FX_OFF_ADD(1, l), FX_OFF_CVT(1, l), fa0, fc0, fa1, fc1);
        fx_node<(MODE >= 8 ? MODE : MODE == 3), N>(v0, x0, fa0, fc0);
        fx_node<(MODE >= 8 ? MODE : MODE == 3), N>(v1, x1, fa1, fc1);
    } else {
        node_n<MODE == 1, N>(v0, x0, t, 0, l);
        node_n<MODE == 1, N>(v1, x1, t, 1, l);
    }
}

struct QTreeCplxArgs {
    const QTreeTable* tab;
    const int32_t* A;  // [2][M][K]
    const int32_t* B;  // [2][N][K]
    char* C;           // [2][M][N] containers
    int64_t M, N, K;
    int32_t cbytes;
};

// Occupancy: the compact forms are bound by how well the other waves cover a wave's scalar-load round trips, so they are compiled
// for one wave per SIMD more than their natural register count gives (4 at MAXL 12, 3 at MAXL 16; 10-20 spilled registers outside
// the hot values) — except Basic's forms whose spill count would be 30-40 (the branching kinds form and the SAT::ZERO feature).
template <int MODE, bool TF>
constexpr bool cplx_dense_waves = MODE >= 2 && (TF || (MODE != 3 && !(MODE >= 8 && ((MODE - 8) & 2))));
template <int MAXL, int MODE, bool TF>   // TF: TFComplexMul (3 multiplications), else BasicComplexMul (4)
__global__ __launch_bounds__(256, ((cplx_dense_waves<MODE, TF>) ? (MAXL == 12 ? 4 : 3) : 1)) void k_tree_cplx(QTreeCplxArgs g)
{
    constexpr int NP = TF ? 3 : 2;   // LDS planes per operand
    constexpr int RW = TF ? 2 : 4;   // leaves per LDS read: with three planes, 8-byte reads keep 24 registers of operands live where 16-byte reads keep 48
    // planes: BasicComplexMul {a, b} / {c, d}; TFComplexMul {a+b, b, b-a} / {c, c+d, d} — the three additions in front of
    // TF's multiplications depend on ONE operand element each, so they are made once per element while the tile is staged,
    // not once per output in the k loop (where they were 12 of 40 quantised values per k step and lane)
    __shared__ __attribute__((aligned(16))) int sA[NP][TMB][PITCH];
    __shared__ __attribute__((aligned(16))) int sB[NP][TNB][PITCH];
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
    // MODE 4 ("one clamp for the whole loop", qg_plan.cpp): ONE range for every value of the k loop and no shift / rounding at the
    // sums and the tree nodes: the bounds and the products' (addend, shift) live in registers for the whole launch; the products'
    // exact left shifts are factors of the operand planes, applied while the tile is staged (QTreeTable::uni)
    int u_lo = 0, u_hi = 0, u_t[4] = {0, 0, 0, 0}, u_d[4] = {0, 0, 0, 0}, u_k[4] = {1, 1, 1, 1};
    if constexpr (MODE == 4) {
        u_lo = tab->uni.lo;
        u_hi = fx_vgpr(tab->uni.hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) { u_t[i] = tab->uni.t[i]; u_d[i] = tab->uni.d[i]; u_k[i] = tab->uni.k[i]; }
    }
    // MODE 5: ... and that range is a signed SAT::TCPL format: LEFT-JUSTIFIED values (qg_fix.h, QTreeTable::lj) — the planes are
    // staged with the shifts that justify each product; the clamp bit of the multiply-add, the add and the subtract saturates
    int j_s = 0, j_mask = -1, j_t[4] = {0, 0, 0, 0}, j_e[6] = {0, 0, 0, 0, 0, 0}, j_pm[4] = {-1, -1, -1, -1};   // j_pm: a product's own mask (QJustify::g)
    if constexpr (MODE == 5) {
        j_s = tab->lj.s;
        j_mask = (int)(~0u << j_s);
#pragma unroll
        for (int i = 0; i < 4; ++i) j_pm[i] = (int)(~0u << (j_s + tab->lj.g[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) j_t[i] = tab->lj.t[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) j_e[i] = tab->lj.e[i];
    }
    auto jmask4 = [&](int (&x)[4]) {
#pragma unroll
        for (int o = 0; o < 4; ++o) x[o] &= j_mask;
    };
    auto uclamp4 = [&](int (&x)[4]) {
#pragma unroll
        for (int o = 0; o < 4; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(x[o]) : "s"(u_lo), "v"(u_hi));
    };

    int low[2][4][4];
    int up[2][MAXL - 4][4];
    int v[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int o = 0; o < 4; ++o) v[p][o] = 0;

    for (int64_t k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();
        // stage: A and B, 2 parts x 32 rows x 32 k = 512 16-byte chunks each; a thread takes both parts of one chunk
        {
            const int r = (tid >> 3) & 31, q = tid & 7;
            int4 x[2], y[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                x[p] = make_int4(0, 0, 0, 0);
                y[p] = make_int4(0, 0, 0, 0);
                if (m0 + r < g.M) x[p] = *(const int4*)(g.A + ((int64_t)p * g.M + m0 + r) * g.K + k0 + q * 4);
                if (n0 + r < g.N) y[p] = *(const int4*)(g.B + ((int64_t)p * g.N + n0 + r) * g.K + k0 + q * 4);
            }
            if constexpr (TF) {
                const int ar4[4] = {x[0].x, x[0].y, x[0].z, x[0].w}, ai4[4] = {x[1].x, x[1].y, x[1].z, x[1].w};
                const int br4[4] = {y[0].x, y[0].y, y[0].z, y[0].w}, bi4[4] = {y[1].x, y[1].y, y[1].z, y[1].w};
                int ab[4], ba[4], cd[4];
                constexpr int SM = (MODE == 4 || MODE == 5) ? 2 : MODE;   // (the staged sums keep their own compact records)
                op_addsub<SM, 4>(ab, ar4, ai4, tab, QG_T_AB, false);  // (a+b), per A element
                op_addsub<SM, 4>(ba, ai4, ar4, tab, QG_T_BA, true);   // (b-a), per A element
                op_addsub<SM, 4>(cd, br4, bi4, tab, QG_T_CD, false);  // (c+d), per B element
                if constexpr (MODE == 4) {   // the products' exact left shifts, once per element: A = (a+b) c, B = (c+d) b, C = (b-a) d
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ab[e] = __mul24(ab[e], u_k[0]); cd[e] = __mul24(cd[e], u_k[1]); ba[e] = __mul24(ba[e], u_k[2]); }
                }
                if constexpr (MODE == 5) {   // planes (a+b), b, (b-a) / c, (c+d), d with the shifts that justify A, B, C
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ab[e] <<= j_e[0]; ba[e] <<= j_e[2]; cd[e] <<= j_e[4]; }
                    x[1] = make_int4(x[1].x << j_e[1], x[1].y << j_e[1], x[1].z << j_e[1], x[1].w << j_e[1]);
                    y[0] = make_int4(y[0].x << j_e[3], y[0].y << j_e[3], y[0].z << j_e[3], y[0].w << j_e[3]);
                    y[1] = make_int4(y[1].x << j_e[5], y[1].y << j_e[5], y[1].z << j_e[5], y[1].w << j_e[5]);
                }
                *(int4*)&sA[0][r][q * 4] = make_int4(ab[0], ab[1], ab[2], ab[3]);
                *(int4*)&sA[1][r][q * 4] = x[1];
                *(int4*)&sA[2][r][q * 4] = make_int4(ba[0], ba[1], ba[2], ba[3]);
                *(int4*)&sB[0][r][q * 4] = y[0];
                *(int4*)&sB[1][r][q * 4] = make_int4(cd[0], cd[1], cd[2], cd[3]);
                *(int4*)&sB[2][r][q * 4] = y[1];
            } else {
                if constexpr (MODE == 5) {   // planes a, b, c, d with the shifts that justify ac, bd, ad, bc
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        x[p] = make_int4(x[p].x << j_e[p], x[p].y << j_e[p], x[p].z << j_e[p], x[p].w << j_e[p]);
                        y[p] = make_int4(y[p].x << j_e[2 + p], y[p].y << j_e[2 + p], y[p].z << j_e[2 + p], y[p].w << j_e[2 + p]);
                    }
                }
                if constexpr (MODE == 4) {   // planes a, b, c, d with their factors
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        x[p] = make_int4(__mul24(x[p].x, u_k[p]), __mul24(x[p].y, u_k[p]), __mul24(x[p].z, u_k[p]), __mul24(x[p].w, u_k[p]));
                        y[p] = make_int4(__mul24(y[p].x, u_k[2 + p]), __mul24(y[p].y, u_k[2 + p]), __mul24(y[p].z, u_k[2 + p]), __mul24(y[p].w, u_k[2 + p]));
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    *(int4*)&sA[p][r][q * 4] = x[p];
                    *(int4*)&sB[p][r][q * 4] = y[p];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; ++kb) {
#pragma unroll
            for (int kq = 0; kq < 16 / RW; ++kq) {
                int a4[NP][2][RW], b4[NP][2][RW];  // [plane][row/col of the 2x2 block][leaf]
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int* pa = &sA[p][ty * 2 + i][kb * 16 + kq * RW];
                        const int* pb = &sB[p][tx + 16 * i][kb * 16 + kq * RW];   // columns tx and tx + 16: 16 consecutive rows per read group, no bank conflict (qg_tree_fast.hip)
                        if constexpr (RW == 4) {
                            const int4 x = *(const int4*)pa, y = *(const int4*)pb;
                            a4[p][i][0] = x.x; a4[p][i][1] = x.y; a4[p][i][2] = x.z; a4[p][i][3] = x.w;
                            b4[p][i][0] = y.x; b4[p][i][1] = y.y; b4[p][i][2] = y.z; b4[p][i][3] = y.w;
                        } else {
                            const int2 x = *(const int2*)pa, y = *(const int2*)pb;
                            a4[p][i][0] = x.x; a4[p][i][1] = x.y;
                            b4[p][i][0] = y.x; b4[p][i][1] = y.y;
                        }
                    }
#pragma unroll
                for (int e = 0; e < RW; ++e) {
                    const int kk = kq * RW + e;   // compile-time leaf index inside the 16-leaf block
                    int a0[2], a1[2], a2[2], b0[2], b1[2], b2[2];  // planes of the A rows / B columns of this lane's 2 x 2 outputs
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a0[i] = a4[0][i][e]; a1[i] = a4[1][i][e]; a2[i] = a4[NP - 1][i][e];
                        b0[i] = b4[0][i][e]; b1[i] = b4[1][i][e]; b2[i] = b4[NP - 1][i][e];
                    }
                    // ---- one complex product per output (x = a+bi from A, y = c+di from B)
                    if constexpr (TF) {   // planes {a+b, b, b-a} x {c, c+d, d}
                        int ab[4], xi[4], ba[4], yr[4], cd[4], yi[4], PA[4], PB[4], PC[4];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                ab[i * 2 + j] = a0[i]; xi[i * 2 + j] = a1[i]; ba[i * 2 + j] = a2[i];
                                yr[i * 2 + j] = b0[j]; cd[i * 2 + j] = b1[j]; yi[i * 2 + j] = b2[j];
                            }
                        if constexpr (MODE == 5) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                PA[o] = sat_mad24_vvs(ab[o], yr[o], j_t[0]) & j_pm[0];
                                PB[o] = sat_mad24_vvs(cd[o], xi[o], j_t[1]) & j_pm[1];
                                PC[o] = sat_mad24_vvs(ba[o], yi[o], j_t[2]) & j_pm[2];
                            }
#pragma unroll
                            for (int o = 0; o < 4; ++o) { v[0][o] = sat_sub(PA[o], PB[o]); v[1][o] = sat_sub(PB[o], PC[o]); }   // (clean subtrahends)
                        } else if constexpr (MODE == 4) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) { PA[o] = mad24_vvs(ab[o], yr[o], u_t[0]); PB[o] = mad24_vvs(cd[o], xi[o], u_t[1]); PC[o] = mad24_vvs(ba[o], yi[o], u_t[2]); }
#pragma unroll
                            for (int o = 0; o < 4; ++o) { PA[o] >>= u_d[0]; PB[o] >>= u_d[1]; PC[o] >>= u_d[2]; }   // (a shift by 0 costs less than a select around it)
                            uclamp4(PA);
                            uclamp4(PB);
                            uclamp4(PC);
#pragma unroll
                            for (int o = 0; o < 4; ++o) { v[0][o] = PA[o] - PB[o]; v[1][o] = PB[o] - PC[o]; }
                            uclamp4(v[0]);
                            uclamp4(v[1]);
                        } else if constexpr (MODE >= 2) {   // records of steps that follow each other share a wait (qg_fix.h)
                            constexpr int KIND = MODE >= 8 ? MODE : MODE == 3;
                            QFix fA, fB, fC, fR, fI;
                            fx_at5(tab, FX_OFF_MUL(QG_T_A), FX_OFF_MUL(QG_T_B), FX_OFF_MUL(QG_T_C), FX_OFF_MUL(QG_T_RE), FX_OFF_MUL(QG_T_IM), fA, fB, fC, fR, fI);
                            fx_mul<KIND, 4>(PA, ab, yr, fA);
                            fx_mul<KIND, 4>(PB, cd, xi, fB);
                            fx_mul<KIND, 4>(PC, ba, yi, fC);
                            fx_addsub<KIND, 4>(v[0], PA, PB, fR, true);
                            fx_addsub<KIND, 4>(v[1], PB, PC, fI, true);
                        } else {
                            op_mul<MODE, 4>(PA, ab, yr, tab, QG_T_A);
                            op_mul<MODE, 4>(PB, cd, xi, tab, QG_T_B);
                            op_mul<MODE, 4>(PC, ba, yi, tab, QG_T_C);
                            op_addsub<MODE, 4>(v[0], PA, PB, tab, QG_T_RE, true);
                            op_addsub<MODE, 4>(v[1], PB, PC, tab, QG_T_IM, true);
                        }
                    } else {
                        int xr[4], xi[4], yr[4], yi[4];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                xr[i * 2 + j] = a0[i]; xi[i * 2 + j] = a1[i];
                                yr[i * 2 + j] = b0[j]; yi[i * 2 + j] = b1[j];
                            }
                        int ac[4], bd[4], ad[4], bc[4];
                        if constexpr (MODE == 5) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                ac[o] = sat_mad24_vvs(xr[o], yr[o], j_t[0]) & j_pm[0];
                                bd[o] = sat_mad24_vvs(xi[o], yi[o], j_t[1]) & j_pm[1];
                                ad[o] = sat_mad24_vvs(xr[o], yi[o], j_t[2]) & j_pm[2];
                                bc[o] = sat_mad24_vvs(xi[o], yr[o], j_t[3]) & j_pm[3];
                            }
#pragma unroll
                            for (int o = 0; o < 4; ++o) { v[0][o] = sat_sub(ac[o], bd[o]); v[1][o] = sat_add(ad[o], bc[o]); }
                        } else if constexpr (MODE == 4) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                ac[o] = mad24_vvs(xr[o], yr[o], u_t[0]) >> u_d[0];
                                bd[o] = mad24_vvs(xi[o], yi[o], u_t[1]) >> u_d[1];
                                ad[o] = mad24_vvs(xr[o], yi[o], u_t[2]) >> u_d[2];
                                bc[o] = mad24_vvs(xi[o], yr[o], u_t[3]) >> u_d[3];
                            }
                            uclamp4(ac);
                            uclamp4(bd);
                            uclamp4(ad);
                            uclamp4(bc);
#pragma unroll
                            for (int o = 0; o < 4; ++o) { v[0][o] = ac[o] - bd[o]; v[1][o] = ad[o] + bc[o]; }
                            uclamp4(v[0]);
                            uclamp4(v[1]);
                        } else if constexpr (MODE >= 2) {
                            constexpr int KIND = MODE >= 8 ? MODE : MODE == 3;
                            QFix f0, f1, f2, f3;
                            fx_at2(tab, FX_OFF_MUL(QG_B_AC), FX_OFF_MUL(QG_B_BD), f0, f1);
                            fx_mul<KIND, 4>(ac, xr, yr, f0);
                            fx_mul<KIND, 4>(bd, xi, yi, f1);
                            fx_at2(tab, FX_OFF_MUL(QG_B_AD), FX_OFF_MUL(QG_B_BC), f2, f3);
                            fx_mul<KIND, 4>(ad, xr, yi, f2);
                            fx_mul<KIND, 4>(bc, xi, yr, f3);
                            fx_at2(tab, FX_OFF_MUL(QG_B_RE), FX_OFF_MUL(QG_B_IM), f0, f1);
                            fx_addsub<KIND, 4>(v[0], ac, bd, f0, true);
                            fx_addsub<KIND, 4>(v[1], ad, bc, f1, false);
                        } else {
                            op_mul<MODE, 4>(ac, xr, yr, tab, QG_B_AC);
                            op_mul<MODE, 4>(bd, xi, yi, tab, QG_B_BD);
                            op_mul<MODE, 4>(ad, xr, yi, tab, QG_B_AD);
                            op_mul<MODE, 4>(bc, xi, yr, tab, QG_B_BC);
                            op_addsub<MODE, 4>(v[0], ac, bd, tab, QG_B_RE, true);
                            op_addsub<MODE, 4>(v[1], ad, bc, tab, QG_B_IM, false);
                        }
                    }
                    // ---- lower four levels (compile-time leaf index)
                    {
                        bool parked_low = false;
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            if (!parked_low) {
                                if (((kk >> l) & 1) == 0) {
#pragma unroll
                                    for (int p = 0; p < 2; ++p)
#pragma unroll
                                        for (int o = 0; o < 4; ++o) low[p][l][o] = v[p][o];
                                    parked_low = true;
                                } else if constexpr (MODE == 5) {
                                    // RE / IM come from one saturating operation on clean values: 2^31 - 1 or clean.  One more
                                    // saturating add keeps floor(v / 2^s) right; before the next one the low bits are cleared
#pragma unroll
                                    for (int o = 0; o < 4; ++o) { v[0][o] = sat_add(low[0][l][o], v[0][o]); v[1][o] = sat_add(low[1][l][o], v[1][o]); }
                                    if ((l & 1) == 0) { jmask4(v[0]); jmask4(v[1]); }
                                } else if constexpr (MODE == 4) {
#pragma unroll
                                    for (int o = 0; o < 4; ++o) { v[0][o] += low[0][l][o]; v[1][o] += low[1][l][o]; }
                                    uclamp4(v[0]);
                                    uclamp4(v[1]);
                                } else {
                                    op_node2<MODE, 4>(v[0], v[1], low[0][l], low[1][l], tab, l);
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
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int o = 0; o < 4; ++o) up[p][u][o] = v[p][o];
                        parked = true;
                    } else if constexpr (MODE == 5) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) { v[0][o] = sat_add(up[0][u][o], v[0][o]); v[1][o] = sat_add(up[1][u][o], v[1][o]); }
                        if ((u & 1) == 0) { jmask4(v[0]); jmask4(v[1]); }   // (level 4 + u: the even ones)
                    } else if constexpr (MODE == 4) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) { v[0][o] += up[0][u][o]; v[1][o] += up[1][u][o]; }
                        uclamp4(v[0]);
                        uclamp4(v[1]);
                    } else {
                        op_node2<MODE, 4>(v[0], v[1], up[0][u], up[1][u], tab, 4 + u);
                    }
                }
            }
        }
    }
    if constexpr (MODE == 5) {   // floor(v / 2^s): the value
#pragma unroll
        for (int o = 0; o < 4; ++o) { v[0][o] >>= j_s; v[1][o] >>= j_s; }
    }
    qg_step_all<int, 4>(v[0], tab->c_cvt[0]);
    qg_step_all<int, 4>(v[1], tab->c_cvt[1]);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t m = m0 + ty * 2 + i, n = n0 + tx + 16 * j;
                if (m < g.M && n < g.N) {
                    const int64_t idx = ((int64_t)p * g.M + m) * g.N + n;
                    const int r = v[p][i * 2 + j];
                    switch (g.cbytes) {
                    case 1: ((int8_t*)g.C)[idx] = (int8_t)r; break;
                    case 2: ((int16_t*)g.C)[idx] = (int16_t)r; break;
                    case 4: ((int32_t*)g.C)[idx] = r; break;
                    default: ((int64_t*)g.C)[idx] = (int64_t)r; break;
                    }
                }
            }
}

// ---- packed 16-bit form (cplx_fixed_ok 6): MODE 5's left-justified steps for a common format of at most 16 bits, TWO outputs
// per register.  A lane owns 4 rows x 2 columns (tx, tx + 16; low half: column tx) of both parts; the A-side planes are staged
// as 16-bit values (two k per dword), the B-side planes as (column tx, column tx + 16) pairs per k; a pair of products is one
// v_pk_mad_i16 ... clamp whose A operand is broadcast by op_sel, + v_and; RE / IM one v_pk_sub_i16 / v_pk_add_i16 ... clamp; a
// node pair one v_pk_add_i16 ... clamp, + v_and at the even levels: 6.3 vector instructions per complex MAC (TF) where MODE 5
// spends 12.9 (BASELINE configuration 5 as literally configured).
constexpr int TM16 = 64;    // rows of C per block (16 thread rows x 4)
constexpr int PKP = 18;     // dwords per A-plane row: 16 + 2 (8-byte reads stay aligned)

template <int MAXL, bool TF>
__global__ __launch_bounds__(256, 3) void k_tree_cplx_pk16(QTreeCplxArgs g)
{
    constexpr int NP = TF ? 3 : 2;
    __shared__ __attribute__((aligned(16))) int sA[NP][TM16][PKP];       // [plane][row][k / 2]: (k even, k odd)
    __shared__ __attribute__((aligned(16))) int sB[NP][TNB / 2][PITCH];  // [plane][column pair][k]: (column p, column p + 16)
    const QTreeTable* __restrict__ tab = g.tab;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t tiles_n = (g.N + TNB - 1) / TNB, tiles_m = (g.M + TM16 - 1) / TM16;
    int64_t bid = blockIdx.x;
    {   // XCD-aware block order, as k_tree_cplx
        const int64_t nwg = tiles_m * tiles_n, q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    constexpr int64_t GMT = 16;
    const int64_t grp = bid / (GMT * tiles_n), first_m = grp * GMT;
    const int64_t gsz = (tiles_m - first_m) < GMT ? (tiles_m - first_m) : GMT;
    const int64_t m0 = (first_m + (bid % (GMT * tiles_n)) % gsz) * TM16, n0 = ((bid % (GMT * tiles_n)) / gsz) * TNB;
    const int nl = tab->n_levels_k;
    const int s16 = tab->lj16.s;
    const int m1 = (0xffff << s16) & 0xffff, mask2 = m1 | (m1 << 16);
    int t2[4], je[6], pm2[4];   // pm2: a product's own mask (QJustify::g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t2[i] = pk2(tab->lj16.t[i], tab->lj16.t[i]);
        const int mi = (0xffff << (s16 + tab->lj16.g[i])) & 0xffff;
        pm2[i] = mi | (mi << 16);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) je[i] = tab->lj16.e[i];

    int low[2][4][4];
    int up[2][MAXL - 4][4];
    int v[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int o = 0; o < 4; ++o) v[p][o] = 0;
#define NODE16C(X0, X1, L)                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int o_ = 0; o_ < 4; ++o_) { v[0][o_] = pk_add_sat(X0[o_], v[0][o_]); v[1][o_] = pk_add_sat(X1[o_], v[1][o_]); } \
        if (((L) & 1) == 0) { _Pragma("unroll") for (int o_ = 0; o_ < 4; ++o_) { v[0][o_] &= mask2; v[1][o_] &= mask2; } }                   \
    } while (0)

    for (int64_t k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();
        // stage A: 64 rows x 32 k of both parts, 2 chunks of 4 k per thread
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = tid + 256 * c, r = ch >> 3, q = ch & 7;
            int4 x[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
            if (m0 + r < g.M) {
                x[0] = *(const int4*)(g.A + ((int64_t)0 * g.M + m0 + r) * g.K + k0 + q * 4);
                x[1] = *(const int4*)(g.A + ((int64_t)1 * g.M + m0 + r) * g.K + k0 + q * 4);
            }
            const int ar4[4] = {x[0].x, x[0].y, x[0].z, x[0].w}, ai4[4] = {x[1].x, x[1].y, x[1].z, x[1].w};
            if constexpr (TF) {   // planes (a+b), b, (b-a)
                int ab[4], ba[4];
                op_addsub<2, 4>(ab, ar4, ai4, tab, QG_T_AB, false);
                op_addsub<2, 4>(ba, ai4, ar4, tab, QG_T_BA, true);
                *(int2*)&sA[0][r][q * 2] = make_int2(pk2(ab[0] << je[0], ab[1] << je[0]), pk2(ab[2] << je[0], ab[3] << je[0]));
                *(int2*)&sA[1][r][q * 2] = make_int2(pk2(ai4[0] << je[1], ai4[1] << je[1]), pk2(ai4[2] << je[1], ai4[3] << je[1]));
                *(int2*)&sA[2][r][q * 2] = make_int2(pk2(ba[0] << je[2], ba[1] << je[2]), pk2(ba[2] << je[2], ba[3] << je[2]));
            } else {              // planes a, b
                *(int2*)&sA[0][r][q * 2] = make_int2(pk2(ar4[0] << je[0], ar4[1] << je[0]), pk2(ar4[2] << je[0], ar4[3] << je[0]));
                *(int2*)&sA[1][r][q * 2] = make_int2(pk2(ai4[0] << je[1], ai4[1] << je[1]), pk2(ai4[2] << je[1], ai4[3] << je[1]));
            }
        }
        // stage B: 16 column pairs x 32 k of both parts, threads 0..127
        if (tid < 128) {
            const int p = tid >> 3, q = tid & 7;
            int4 y[2][2];   // [column of the pair][part]
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    y[c][part] = make_int4(0, 0, 0, 0);
                    if (n0 + p + 16 * c < g.N) y[c][part] = *(const int4*)(g.B + ((int64_t)part * g.N + n0 + p + 16 * c) * g.K + k0 + q * 4);
                }
            int br[2][4], bi[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                br[c][0] = y[c][0].x; br[c][1] = y[c][0].y; br[c][2] = y[c][0].z; br[c][3] = y[c][0].w;
                bi[c][0] = y[c][1].x; bi[c][1] = y[c][1].y; bi[c][2] = y[c][1].z; bi[c][3] = y[c][1].w;
            }
            if constexpr (TF) {   // planes c, (c+d), d
                int cd[2][4];
                op_addsub<2, 4>(cd[0], br[0], bi[0], tab, QG_T_CD, false);
                op_addsub<2, 4>(cd[1], br[1], bi[1], tab, QG_T_CD, false);
                *(int4*)&sB[0][p][q * 4] = make_int4(pk2(br[0][0] << je[3], br[1][0] << je[3]), pk2(br[0][1] << je[3], br[1][1] << je[3]),
                                                     pk2(br[0][2] << je[3], br[1][2] << je[3]), pk2(br[0][3] << je[3], br[1][3] << je[3]));
                *(int4*)&sB[1][p][q * 4] = make_int4(pk2(cd[0][0] << je[4], cd[1][0] << je[4]), pk2(cd[0][1] << je[4], cd[1][1] << je[4]),
                                                     pk2(cd[0][2] << je[4], cd[1][2] << je[4]), pk2(cd[0][3] << je[4], cd[1][3] << je[4]));
                *(int4*)&sB[2][p][q * 4] = make_int4(pk2(bi[0][0] << je[5], bi[1][0] << je[5]), pk2(bi[0][1] << je[5], bi[1][1] << je[5]),
                                                     pk2(bi[0][2] << je[5], bi[1][2] << je[5]), pk2(bi[0][3] << je[5], bi[1][3] << je[5]));
            } else {              // planes c, d
                *(int4*)&sB[0][p][q * 4] = make_int4(pk2(br[0][0] << je[2], br[1][0] << je[2]), pk2(br[0][1] << je[2], br[1][1] << je[2]),
                                                     pk2(br[0][2] << je[2], br[1][2] << je[2]), pk2(br[0][3] << je[2], br[1][3] << je[2]));
                *(int4*)&sB[1][p][q * 4] = make_int4(pk2(bi[0][0] << je[3], bi[1][0] << je[3]), pk2(bi[0][1] << je[3], bi[1][1] << je[3]),
                                                     pk2(bi[0][2] << je[3], bi[1][2] << je[3]), pk2(bi[0][3] << je[3], bi[1][3] << je[3]));
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; ++kb) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                int2 a2[NP][4];
                int4 b4[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a2[p][i] = *(const int2*)&sA[p][ty * 4 + i][kb * 8 + kq * 2];
                    b4[p] = *(const int4*)&sB[p][tx][kb * 16 + kq * 4];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = kq * 4 + e;
                    int bv[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) bv[p] = e == 0 ? b4[p].x : e == 1 ? b4[p].y : e == 2 ? b4[p].z : b4[p].w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int av[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) av[p] = e < 2 ? a2[p][i].x : a2[p][i].y;
                        if constexpr (TF) {   // A = (a+b) c, B = (c+d) b, C = (b-a) d; re = A - B, im = B - C
                            const int PA = ((e & 1) ? pk_mad_sat<1>(av[0], bv[0], t2[0]) : pk_mad_sat<0>(av[0], bv[0], t2[0])) & pm2[0];
                            const int PB = ((e & 1) ? pk_mad_sat<1>(av[1], bv[1], t2[1]) : pk_mad_sat<0>(av[1], bv[1], t2[1])) & pm2[1];
                            const int PC = ((e & 1) ? pk_mad_sat<1>(av[2], bv[2], t2[2]) : pk_mad_sat<0>(av[2], bv[2], t2[2])) & pm2[2];
                            v[0][i] = pk_sub_sat(PA, PB);
                            v[1][i] = pk_sub_sat(PB, PC);
                        } else {              // re = ac - bd, im = ad + bc
                            const int ac = ((e & 1) ? pk_mad_sat<1>(av[0], bv[0], t2[0]) : pk_mad_sat<0>(av[0], bv[0], t2[0])) & pm2[0];
                            const int bd = ((e & 1) ? pk_mad_sat<1>(av[1], bv[1], t2[1]) : pk_mad_sat<0>(av[1], bv[1], t2[1])) & pm2[1];
                            const int ad = ((e & 1) ? pk_mad_sat<1>(av[0], bv[1], t2[2]) : pk_mad_sat<0>(av[0], bv[1], t2[2])) & pm2[2];
                            const int bc = ((e & 1) ? pk_mad_sat<1>(av[1], bv[0], t2[3]) : pk_mad_sat<0>(av[1], bv[0], t2[3])) & pm2[3];
                            v[0][i] = pk_sub_sat(ac, bd);
                            v[1][i] = pk_add_sat(ad, bc);
                        }
                    }
                    bool parked_low = false;
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        if (!parked_low) {
                            if (((kk >> l) & 1) == 0) {
#pragma unroll
                                for (int p = 0; p < 2; ++p)
#pragma unroll
                                    for (int o = 0; o < 4; ++o) low[p][l][o] = v[p][o];
                                parked_low = true;
                            } else {
                                NODE16C(low[0][l], low[1][l], l);
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
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int o = 0; o < 4; ++o) up[p][u][o] = v[p][o];
                        parked = true;
                    } else {
                        NODE16C(up[0][u], up[1][u], 4 + u);
                    }
                }
            }
        }
    }
#undef NODE16C
    // the roots: floor(half / 2^s) of each half, then the conversion into C
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int r8[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r8[i * 2 + 0] = ((int)((unsigned)v[p][i] << 16) >> 16) >> s16;
            r8[i * 2 + 1] = (v[p][i] >> 16) >> s16;
        }
        qg_step_all<int, 8>(r8, tab->c_cvt[p]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t m = m0 + ty * 4 + i, n = n0 + tx + 16 * j;
                if (m < g.M && n < g.N) {
                    const int64_t idx = ((int64_t)p * g.M + m) * g.N + n;
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
}

} // namespace

hipError_t qg_launch_tree_cplx_fast(const QTreeTable* dev_table, int n_levels, int fixed, int tf, const void* A, const void* B, void* C, int64_t M,
                                    int64_t N, int64_t K, int cbytes, hipStream_t st)
{
    if (K % KC != 0 || n_levels < 5 || n_levels > 16) return hipErrorInvalidValue;
    QTreeCplxArgs g{dev_table, (const int32_t*)A, (const int32_t*)B, (char*)C, M, N, K, cbytes};
    const int64_t blocks = ((M + TMB - 1) / TMB) * ((N + TNB - 1) / TNB);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    // fixed: 0 run-time modes; 1 fixed modes, steps read from the step table; 2 fixed modes, compact branch-free steps; 3 compact
    // steps with a branch on the rounding / overflow kind; 8 + f compact steps, branch-free kinds of the feature set f (1 R, 2 Z,
    // 3 RZ, 4 W, 5 RW; the other ones run the full set)
#define QG_CPLX_LAUNCH(MODE)                                                                                                      \
    do {                                                                                                                          \
        if (tf) {                                                                                                                 \
            if (n_levels <= 12) hipLaunchKernelGGL((k_tree_cplx<12, MODE, true>), dim3((unsigned)blocks), dim3(256), 0, st, g);   \
            else hipLaunchKernelGGL((k_tree_cplx<16, MODE, true>), dim3((unsigned)blocks), dim3(256), 0, st, g);                  \
        } else {                                                                                                                  \
            if (n_levels <= 12) hipLaunchKernelGGL((k_tree_cplx<12, MODE, false>), dim3((unsigned)blocks), dim3(256), 0, st, g);  \
            else hipLaunchKernelGGL((k_tree_cplx<16, MODE, false>), dim3((unsigned)blocks), dim3(256), 0, st, g);                 \
        }                                                                                                                         \
    } while (0)
    const int base = fixed >> 8;   // (forms 5 / 6: 4 when the strict one-clamp form holds as well, else 2)
    fixed &= 255;
    switch (fixed) {
    case 0: QG_CPLX_LAUNCH(0); break;
    case 1: QG_CPLX_LAUNCH(1); break;
    case 2: QG_CPLX_LAUNCH(2); break;
    case 3: QG_CPLX_LAUNCH(3); break;
    case 6:     // ... in packed 16-bit halves
    case 5:     // ... on left-justified values
    case 4: {   // one clamp for the whole loop (qg_plan.cpp)
        static const bool no_uniform = QG_DIAG_ENV("QG_NO_UNIFORM_CLAMP");   // A/B switch (diagnostic library): the compact form such a descriptor had before
        static const bool no_lj = QG_DIAG_ENV("QG_NO_LEFT_JUSTIFIED");
        static const bool no_pk = QG_DIAG_ENV("QG_NO_PACKED16");
        if (no_uniform) QG_CPLX_LAUNCH(2);
        else if (fixed == 6 && !no_lj && !no_pk) {
            const int64_t blocks16 = ((M + TM16 - 1) / TM16) * ((N + TNB - 1) / TNB);
            if (tf) {
                if (n_levels <= 12) hipLaunchKernelGGL((k_tree_cplx_pk16<12, true>), dim3((unsigned)blocks16), dim3(256), 0, st, g);
                else hipLaunchKernelGGL((k_tree_cplx_pk16<16, true>), dim3((unsigned)blocks16), dim3(256), 0, st, g);
            } else {
                if (n_levels <= 12) hipLaunchKernelGGL((k_tree_cplx_pk16<12, false>), dim3((unsigned)blocks16), dim3(256), 0, st, g);
                else hipLaunchKernelGGL((k_tree_cplx_pk16<16, false>), dim3((unsigned)blocks16), dim3(256), 0, st, g);
            }
        } else if (fixed >= 5 && !no_lj) QG_CPLX_LAUNCH(5);
        else if (fixed >= 5 && base != 4) QG_CPLX_LAUNCH(2);
        else QG_CPLX_LAUNCH(4);
        break;
    }
    case 9: QG_CPLX_LAUNCH(9); break;
    case 10: QG_CPLX_LAUNCH(10); break;
    case 11: QG_CPLX_LAUNCH(11); break;
    case 12: QG_CPLX_LAUNCH(12); break;
    case 13: QG_CPLX_LAUNCH(13); break;
    case 14:
    case 15: QG_CPLX_LAUNCH(15); break;
    default: return hipErrorInvalidValue;
    }
#undef QG_CPLX_LAUNCH
    return hipGetLastError();
}
