// qg_gemv.hip — exact tree-order evaluation for ONE output column (N = 1): the batched Qreduce / fixed-point GEMV of
// SURVEY.md §8-f #1 (Qreduce<L…>(v), /root/reference/include/QuBLAS.h:4960-4990, :5014-5018; as a Qgemul the
// operand B is the vector, for Qreduce the constant 1).  HBM-bound byte work: every row of packed A (K int32 values) is
// read once, so the kernel is laid out for coalesced reads and for as few quantising instructions per leaf as the exact
// tree allows; the GEMM tree kernels would spend a 32-column tile on the single column.
//
// One wave owns one row at a time.  A row is consumed in segments of SEG = 64 * CH leaves (CH = 2^q <= 32 per lane):
//   * the segment is fetched with lane-linear 16-byte loads (fully coalesced) one segment AHEAD into registers, written
//     to the wave's private LDS image and read back chunk-wise, so that lane j holds the CH CONSECUTIVE leaves
//     [j*CH, (j+1)*CH) — the image pads every chunk by 16 bytes, which makes the chunk reads bank-conflict free;
//   * products are quantised per leaf (Qmul, QuBLAS.h:3152-3170) — for a 0/1-valued B of the Qreduce lowering the
//     product is a select;
//   * q tree levels run inside the lane on register arrays (the wave-uniform mode switches hoisted around the arrays,
//     qg_step_all.h), 6 more across the lanes (lane 2s*i takes its partner s lanes up), which leaves the segment's
//     partial result — a node of level q+6 — in lane 0; segments are combined by a binary counter over the remaining
//     levels.  Every node is the reference's Qadd of two equal-format children in the reference's order.
// Rows of 16 .. 128 leaves take k_gemv_short (below), which needs no LDS.
// Requirements (planner): real descriptor, N = 1, at least 4 tree levels (the operands are zero-padded to 2^n_levels leaves),
// every value except the unrounded product within 31 bits.  When the whole vector is one segment, B is staged once per workgroup in the same padded image; longer
// vectors are re-read per segment from L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>

#include "qg_kernels.h"
#include "qg_fix.h"
#include "qg_step_all.h"

namespace {

struct QGemvArgs {
    const QTreeTable* tab;
    const int32_t* A;   // [M][K], K contiguous
    const int32_t* B;   // [K]
    char* C;            // [M] containers of cbytes
    int64_t M, K;
    int32_t cbytes, n_levels, b_is_bit;
    int32_t pad_;       // fixed-mode selector: 0 run-time modes, 1 one format + SAT::ZERO, 2 one format + SAT::TCPL (QAnalysis::gemv_fixed)
};

typedef int v4i __attribute__((ext_vector_type(4)));

// The per-level table is read through the CONSTANT address space: the kernel stores a result per row, and behind a
// possibly-aliasing global store the compiler would otherwise re-read every (wave-uniform) table field with per-lane
// vector loads instead of scalar loads.  The table is written once by the host before the launch.
typedef const __attribute__((address_space(4))) QTreeTable* CTab;
typedef const __attribute__((address_space(4))) QStep* CStep;
__device__ __forceinline__ QStep load_step(CStep p)
{
    // the opaque zero keeps the (loop-invariant) scalar loads of ALL levels from being hoisted out of the row loop, where
    // they would hold some 10 scalar registers per level and spill; each use re-reads its 40 bytes through the scalar cache
    int zero = 0;
    asm volatile("" : "+s"(zero));
    p += zero;
    QStep s;
    s.d = p->d; s.Q = p->Q; s.O = p->O; s.W = p->W; s.S = p->S; s.identity = p->identity; s.lo = p->lo; s.hi = p->hi;
    return s;
}

constexpr int WAVES = 16;  // 16 waves x one 8 KB segment prefetched in registers = 128 KB in flight per CU
constexpr int MAXUP = 20;   // levels above a segment (K < 2^31)

// node of level l: acc (LEFT child) + x, quantised by the level's add format, then stored into the level buffer's type
// (children share one format: no alignment shift in a real GEMM tree)
template <int N, class T = int>
__device__ __forceinline__ void node_all(T (&acc)[N], const T (&x)[N], CTab tab, int level)
{
#pragma unroll
    for (int o = 0; o < N; ++o) acc[o] += x[o];
    qg_step_all<T, N>(acc, load_step(&tab->level_add[0][level].q));   // (level_cvt is the identity in a real GEMM tree)
}

// in-lane levels: v[0..CNT) -> v[0..CNT/2) ... -> v[0]
// MODE 1 / 2: every level has ONE format with no rounding shift and SAT::ZERO / SAT::TCPL overflow (what default tags
// produce; QAnalysis::fast_mode) — a node is an add and a range test / a clamp against two constants, with no per-level
// table read and no scalar branch ladder.  The counters of the run-time-mode kernel showed more scalar than vector
// instructions per launch (65 M SALU vs 48 M VALU at 65536 x 4096) and 56 % of the wave cycles waiting.
template <int MODE>
__device__ __forceinline__ int node_fixed(int a, int b, int lo, int hi)
{
    if (MODE == 6 || MODE == 7) return sat_add(a, b);   // 32-bit words (QAnalysis::gemv_w32): the format's range IS the int32 range — one v_add_i32 ... clamp
    const int t = a + b;
    if (MODE == 1) return ((unsigned)(t - lo) > (unsigned)(hi - lo)) ? 0 : t;
    return qg_clamp_i32(t, lo, hi);
}

// MODE 3 / 5: per-level formats in the compact records of qg_plan.h (QFix; QAnalysis::gemv_fixed): acc = acc + x + the rounding
// addend, a left shift where the level has more fraction bits, then a right shift and one clamp (3: every level clamps) or what
// the record's overflow kind asks for (5: SAT::ZERO / WRP::TCPL levels exist) — no mode ladder, one scalar load per level.
template <int MODE, int N>
__device__ __forceinline__ void node_rec(int (&acc)[N], const int (&x)[N], const QTreeTable* t, int level)
{
    const QFix f = fx_at(t, FX_OFF_ADD(0, level));
#pragma unroll
    for (int o = 0; o < N; ++o) acc[o] = acc[o] + x[o] + f.t;
    if (f.ls) {
#pragma unroll
        for (int o = 0; o < N; ++o) acc[o] = (int)((unsigned)acc[o] << f.ls);
    }
    if (MODE == 5) fx_finish_any<N>(acc, f);
    else fx_finish<N>(acc, f);
}

template <int CNT, int MODE, class T = int>
__device__ __forceinline__ T lane_tree(T (&v)[CNT], CTab tab, int level, int lo, int hi)
{
    if constexpr (CNT == 1) {
        return v[0];
    } else {
        T h[CNT / 2];   // (arrays of exact size, indexed only by unrolled loops: registers, never scratch)
        if constexpr (MODE == 0) {
#pragma unroll
            for (int o = 0; o < CNT / 2; ++o) h[o] = v[2 * o] + v[2 * o + 1];   // left child + right child, in the tree's order
            qg_step_all<T, CNT / 2>(h, load_step(&tab->level_add[0][level].q));
        } else if constexpr (MODE == 3 || MODE == 5) {
            int r[CNT / 2];
#pragma unroll
            for (int o = 0; o < CNT / 2; ++o) { h[o] = v[2 * o]; r[o] = v[2 * o + 1]; }
            node_rec<MODE, CNT / 2>(h, r, (const QTreeTable*)tab, level);
        } else {
#pragma unroll
            for (int o = 0; o < CNT / 2; ++o) h[o] = node_fixed<MODE>(v[2 * o], v[2 * o + 1], lo, hi);
        }
        return lane_tree<CNT / 2, MODE, T>(h, tab, level + 1, lo, hi);
    }
}

// T = int64_t (MODE 0 only): the tree's values need more than 31 bits — sums of 32-bit words, wide level types — while the
// elements still come in 4-byte containers; the same streaming structure on 64-bit nodes (still HBM-bound by far: the general
// 64-bit tree kernel, built for square output tiles, ran a 65 536 x 4096 reduction of Q15.16 at 140 GB/s).
template <int CH, int MODE, class T = int>   // leaves per lane and segment; MODE: 0 run-time modes, 1 / 2 fixed (see node_fixed)
__global__ __launch_bounds__(64 * WAVES) void k_gemv(QGemvArgs g)
{
    static_assert(sizeof(T) == 4 || MODE == 0, "64-bit values: run-time modes");
    constexpr int Q = CH == 32 ? 5 : CH == 16 ? 4 : CH == 8 ? 3 : 2;
    static_assert(CH <= 32 && CH >= 4, "leaves per lane");
    constexpr int SEG = 64 * CH;                 // leaves per segment
    constexpr int LOADS = CH / 4;                // 16-byte loads per lane and segment
    constexpr int CHUNK = CH * 4 + 16;           // padded chunk pitch in bytes
    constexpr int IMG = 64 * CHUNK;              // one padded segment image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const CTab tab = (CTab)g.tab;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* bimg = smem;                                    // B image when the whole vector is one segment; else B is read from L2
    const int64_t nseg = g.K / SEG;
    const bool b_in_lds = nseg == 1;
    char* mine = smem + (b_in_lds ? IMG : 0) + wave * IMG;

    // padded image offset of the 16-byte piece p (= 4 leaves) of a segment: chunk p / LOADS, slot p % LOADS
    auto img_off = [&](int p) { return (p / LOADS) * CHUNK + (p % LOADS) * 16; };

    if (b_in_lds) {
        for (int64_t p = threadIdx.x; p < nseg * (SEG / 4); p += 64 * WAVES) {
            const int64_t s = p / (SEG / 4);
            *(v4i*)(bimg + s * IMG + img_off((int)(p % (SEG / 4)))) = *(const v4i*)(g.B + p * 4);
        }
        __syncthreads();
    }

    // binary counter over the levels above a segment (only vectors longer than one segment use it): per wave in LDS, so
    // that it can be indexed at run time; lane 0 holds the real partial results and is the only writer
    __shared__ T upbuf[WAVES][MAXUP];
    T* up = &upbuf[wave][0];
    const int64_t wstride = (int64_t)gridDim.x * WAVES;
    int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    v4i nxt[LOADS];
    auto fetch = [&](int64_t r, int64_t s) {
        const int32_t* src = g.A + r * g.K + s * SEG + lane * 4;
#pragma unroll
        for (int t = 0; t < LOADS; ++t) nxt[t] = *(const v4i*)(src + t * 256);
    };
    if (row < g.M) fetch(row, 0);
    QNode pnode;
    pnode.sa = pnode.sb = 0;
    pnode.q = load_step(&tab->mul[0].q);
    const QStep c_cvt = load_step(&tab->c_cvt[0]);
    const int flo = (int)tab->level_add[0][0].q.lo, fhi = (int)tab->level_add[0][0].q.hi;   // MODE 1 / 2: the one level format
    // MODE 7: the product's shift, its rounding addend (TRN::TCPL 0, RND::POS_INF 2^(d-1), RND::NEG_INF 2^(d-1) - 1) and the range test's constants
    const int w_d = MODE == 7 ? pnode.q.d : 0;
    const int w_t = MODE == 7 ? (pnode.q.Q == QG_RND_POS_INF ? 1 << ((w_d - 1) & 31) : pnode.q.Q == QG_RND_NEG_INF ? (1 << ((w_d - 1) & 31)) - 1 : 0) : 0;
    unsigned w_half = MODE == 7 ? 1u << ((w_d - 1) & 31) : 0u, w_lim = MODE == 7 ? 1u << (w_d & 31) : 0u;
    if (MODE == 7) asm volatile("" : "+s"(w_lim));   // (opaque: see k_tree_fast)
    for (; row < g.M; row += wstride) {
        T root = 0;
        for (int64_t s = 0; s < nseg; ++s) {
            // publish the prefetched segment in the wave's image, start fetching the next one
#pragma unroll
            for (int t = 0; t < LOADS; ++t) *(v4i*)(mine + img_off(t * 64 + lane)) = nxt[t];
            {
                int64_t nr = row, ns = s + 1;
                if (ns == nseg) { ns = 0; nr = row + wstride; }
                if (nr < g.M) fetch(nr, ns);
            }
            T v[CH];
            if (g.b_is_bit) {
#pragma unroll
                for (int t = 0; t < LOADS; ++t) {
                    const v4i a = *(const v4i*)(mine + lane * CHUNK + t * 16);
                    v4i b;
                    if (b_in_lds) b = *(const v4i*)(bimg + lane * CHUNK + t * 16);
                    else b = *(const v4i*)(g.B + s * SEG + lane * CH + t * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * t + e] = b[e] ? a[e] : 0;   // Qmul(a, 0|1) into a's own format: exact
                }
            } else {
#pragma unroll
                for (int t = 0; t < LOADS; ++t) {
                    const v4i a = *(const v4i*)(mine + lane * CHUNK + t * 16);
                    v4i b;
                    if (b_in_lds) b = *(const v4i*)(bimg + lane * CHUNK + t * 16);
                    else b = *(const v4i*)(g.B + s * SEG + lane * CH + t * 4);
                    if constexpr (MODE == 7) {
                        // 32-bit words, product "add a constant, shift right by 1 ... 31, saturate to the word" (QAnalysis::gemv_fixed 7):
                        // the steps of k_tree_fast<., 17> — v_mad_i64_i32, v_alignbit, range test of the high half, select
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const long long pr = (long long)a[e] * (long long)b[e] + (long long)w_t;
                            const int ph = (int)(pr >> 32);
                            const int ql = (int)__builtin_amdgcn_alignbit((unsigned)ph, (unsigned)pr, (unsigned)w_d);
                            v[4 * t + e] = (unsigned)ph + w_half < w_lim ? ql : ((ph >> 31) ^ 0x7fffffff);
                        }
                    } else {
                        int64_t p[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) p[e] = (int64_t)a[e] * (int64_t)b[e];
                        qg_step_all<int64_t, 4>(p, pnode.q);                          // Qmul: round + overflow into the product format
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * t + e] = (T)p[e];
                    }
                }
            }
            // across the lanes: level Q + i pairs lane j (left) with lane j + 2^i
            T x[1] = {lane_tree<CH, MODE, T>(v, tab, 0, flo, fhi)};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T y[1] = {__shfl_down(x[0], 1 << i)};
                if constexpr (MODE == 0) node_all<1, T>(x, y, tab, Q + i);
                else if constexpr (MODE == 3 || MODE == 5) node_rec<MODE, 1>(x, y, g.tab, Q + i);
                else x[0] = node_fixed<MODE>(x[0], y[0], flo, fhi);
            }
            // x[0] in lane 0 = the segment's node of level Q + 6; carry it into the counter
            const int base = Q + 6;
            bool parked = false;   // wave-uniform
            for (int u = 0; u < MAXUP && !parked; ++u) {
                if (base + u >= g.n_levels) { root = x[0]; parked = true; }
                else if (((s >> u) & 1) == 0) { if (lane == 0) up[u] = x[0]; parked = true; }
                else if constexpr (MODE == 0) {
                    T l[1] = {up[u]};
                    node_all<1, T>(l, x, tab, base + u);
                    x[0] = l[0];
                } else if constexpr (MODE == 3 || MODE == 5) {
                    int l[1] = {up[u]};
                    node_rec<MODE, 1>(l, x, g.tab, base + u);
                    x[0] = l[0];
                } else {
                    x[0] = node_fixed<MODE>(up[u], x[0], flo, fhi);
                }
            }
        }
        typename std::conditional<MODE == 6 || MODE == 7, int64_t, T>::type r[1] = {root};   // (MODE 6 / 7: a rounding addend on a full 32-bit word needs the 64-bit step)
        qg_step_all<typename std::conditional<MODE == 6 || MODE == 7, int64_t, T>::type, 1>(r, c_cvt);
        if (lane == 0) {
            switch (g.cbytes) {
            case 1: ((int8_t*)g.C)[row] = (int8_t)r[0]; break;
            case 2: ((int16_t*)g.C)[row] = (int16_t)r[0]; break;
            case 4: ((int32_t*)g.C)[row] = (int32_t)r[0]; break;
            default: ((int64_t*)g.C)[row] = (int64_t)r[0]; break;
            }
        }
    }
}

// Short rows (K = 16 .. 128): a row is K/4 lanes wide, so one 16-byte load per lane covers 256/K whole rows and the lane's
// four values are already consecutive leaves — no LDS.  Two levels inside the lane, log2(K) - 2 across the lanes of the
// row's group; U row groups are processed together so that the (wave-uniform) mode switches are paid once per U values.
template <int KK, int MODE, class T = int>   // MODE: see node_fixed; T = int64_t (MODE 0): see k_gemv
__global__ __launch_bounds__(256) void k_gemv_short(QGemvArgs g)
{
    static_assert(sizeof(T) == 4 || MODE == 0, "64-bit values: run-time modes");
    constexpr int LPR = KK / 4;        // lanes per row
    constexpr int RPL = 64 / LPR;      // rows per wave-wide load
    constexpr int U = 4;               // loads in flight per lane
    constexpr int XL = KK == 16 ? 2 : KK == 32 ? 3 : KK == 64 ? 4 : 5;   // levels across lanes
    const CTab tab = (CTab)g.tab;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int sub = lane % LPR, rl = lane / LPR;
    const v4i b = *(const v4i*)(g.B + sub * 4);
    QNode pnode;
    pnode.sa = pnode.sb = 0;
    pnode.q = load_step(&tab->mul[0].q);
    const QStep c_cvt = load_step(&tab->c_cvt[0]);
    const int flo = (int)tab->level_add[0][0].q.lo, fhi = (int)tab->level_add[0][0].q.hi;   // MODE 1 / 2: the one level format
    const int64_t groups = (g.M + RPL - 1) / RPL;            // groups of RPL rows
    for (int64_t g0 = wave * U; g0 < groups; g0 += nwaves * U) {
        v4i a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (g0 + u) * RPL + rl;
            a[u] = row < g.M ? *(const v4i*)(g.A + row * KK + sub * 4) : v4i{0, 0, 0, 0};
        }
        T l0[2 * U];   // level-0 inputs pairwise: (p0 + p1), (p2 + p3) per load
        {
            T p[4 * U];
            if (g.b_is_bit) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[4 * u + e] = b[e] ? a[u][e] : 0;
            } else {
                int64_t w[4 * U];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[4 * u + e] = (int64_t)a[u][e] * (int64_t)b[e];
                qg_step_all<int64_t, 4 * U>(w, pnode.q);
#pragma unroll
                for (int o = 0; o < 4 * U; ++o) p[o] = (T)w[o];
            }
            if constexpr (MODE >= 3) {
                int r0[2 * U];
#pragma unroll
                for (int o = 0; o < 2 * U; ++o) { l0[o] = p[2 * o]; r0[o] = p[2 * o + 1]; }
                node_rec<MODE, 2 * U>(l0, r0, g.tab, 0);
            } else {
#pragma unroll
                for (int o = 0; o < 2 * U; ++o) {
                    if constexpr (MODE == 0) l0[o] = p[2 * o] + p[2 * o + 1];
                    else l0[o] = node_fixed<(MODE == 1 || MODE == 2) ? MODE : 1>(p[2 * o], p[2 * o + 1], flo, fhi);
                }
            }
        }
        if constexpr (MODE == 0) qg_step_all<T, 2 * U>(l0, load_step(&tab->level_add[0][0].q));
        T x[U];
        if constexpr (MODE >= 3) {
            int r1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { x[u] = l0[2 * u]; r1[u] = l0[2 * u + 1]; }
            node_rec<MODE, U>(x, r1, g.tab, 1);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (MODE == 0) x[u] = l0[2 * u] + l0[2 * u + 1];
                else x[u] = node_fixed<(MODE == 1 || MODE == 2) ? MODE : 1>(l0[2 * u], l0[2 * u + 1], flo, fhi);
            }
        }
        if constexpr (MODE == 0) qg_step_all<T, U>(x, load_step(&tab->level_add[0][1].q));
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            T y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = __shfl_down(x[u], 1 << i);
            if constexpr (MODE == 0) node_all<U, T>(x, y, tab, 2 + i);
            else if constexpr (MODE >= 3) node_rec<MODE, U>(x, y, g.tab, 2 + i);
            else {
#pragma unroll
                for (int u = 0; u < U; ++u) x[u] = node_fixed<(MODE == 1 || MODE == 2) ? MODE : 1>(x[u], y[u], flo, fhi);
            }
        }
        qg_step_all<T, U>(x, c_cvt);
        if (sub == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = (g0 + u) * RPL + rl;
                if (row < g.M) {
                    switch (g.cbytes) {
                    case 1: ((int8_t*)g.C)[row] = (int8_t)x[u]; break;
                    case 2: ((int16_t*)g.C)[row] = (int16_t)x[u]; break;
                    case 4: ((int32_t*)g.C)[row] = x[u]; break;
                    default: ((int64_t*)g.C)[row] = (int64_t)x[u]; break;
                    }
                }
            }
        }
    }
}

template <int KK, int MODE = 0>
hipError_t launch_gemv_short(const QGemvArgs& g, hipStream_t st)
{
    if constexpr (MODE == 0) {
        if (g.pad_ == 1) return launch_gemv_short<KK, 1>(g, st);
        if (g.pad_ == 2) return launch_gemv_short<KK, 2>(g, st);
        if (g.pad_ == 3) return launch_gemv_short<KK, 3>(g, st);
        if (g.pad_ == 5) return launch_gemv_short<KK, 5>(g, st);
    }
    constexpr int RPL = 64 / (KK / 4);
    const int64_t groups = (g.M + RPL - 1) / RPL;
    int64_t blocks = (groups + 15) / 16;          // 4 waves x U = 4 groups per block pass
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((k_gemv_short<KK, MODE>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError();
}

template <int KK>
hipError_t launch_gemv_short_wide(const QGemvArgs& g, hipStream_t st)   // 64-bit values (k_gemv_short<KK, 0, int64_t>)
{
    constexpr int RPL = 64 / (KK / 4);
    const int64_t groups = (g.M + RPL - 1) / RPL;
    int64_t blocks = (groups + 15) / 16;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((k_gemv_short<KK, 0, int64_t>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError();
}

template <int CH>
hipError_t launch_gemv_wide(const QGemvArgs& g, hipStream_t st)   // 64-bit values (k_gemv<CH, 0, int64_t>)
{
    constexpr int IMG = 64 * (CH * 4 + 16);
    const int64_t nseg = g.K / (64 * CH);
    const int lds = IMG * ((nseg == 1 ? 1 : 0) + WAVES);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = qg_lds_attr((const void*)k_gemv<CH, 0, int64_t>, IMG * (1 + WAVES), attr_done); e != hipSuccess) return e;
    int64_t blocks = (g.M + WAVES - 1) / WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL((k_gemv<CH, 0, int64_t>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, g);
    return hipGetLastError();
}

template <int CH, int MODE = 0>
hipError_t launch_gemv(const QGemvArgs& g, hipStream_t st)
{
    if constexpr (MODE == 0) {
        if (g.pad_ == 1) return launch_gemv<CH, 1>(g, st);
        if (g.pad_ == 2) return launch_gemv<CH, 2>(g, st);
        if (g.pad_ == 3) return launch_gemv<CH, 3>(g, st);
        if (g.pad_ == 5) return launch_gemv<CH, 5>(g, st);
        if (g.pad_ == 6) return launch_gemv<CH, 6>(g, st);
        if (g.pad_ == 7) return launch_gemv<CH, 7>(g, st);
    }
    constexpr int IMG = 64 * (CH * 4 + 16);
    const int64_t nseg = g.K / (64 * CH);
    const int lds = IMG * ((nseg == 1 ? 1 : 0) + WAVES);
    static std::atomic<uint64_t> attr_done{0};   // one bit per device
    if (hipError_t e = qg_lds_attr((const void*)k_gemv<CH, MODE>, IMG * (1 + WAVES), attr_done); e != hipSuccess) return e;
    int64_t blocks = (g.M + WAVES - 1) / WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;   // rows beyond that are walked by the grid-stride loop
    hipLaunchKernelGGL((k_gemv<CH, MODE>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, g);
    return hipGetLastError();
}

} // namespace

hipError_t qg_launch_gemv(const QTreeTable* dev_table, int n_levels, int b_is_bit, int fixed_mode, const void* A, const void* B, void* C,
                          int64_t M, int64_t K, int cbytes, hipStream_t st, int wide)
{
    if (M <= 0) return hipSuccess;
    if (K < 16 || (K & (K - 1)) || n_levels < 4 || n_levels > 10 + MAXUP) return hipErrorInvalidValue;
    QGemvArgs g{dev_table, (const int32_t*)A, (const int32_t*)B, (char*)C, M, K, cbytes, n_levels, b_is_bit, wide ? 0 : fixed_mode};
    if (wide) {   // 64-bit tree values on 4-byte elements: 8 leaves per lane and segment (16 would hold 64 registers of values alone)
        if (K >= 1024) return launch_gemv_wide<16>(g, st);
        if (K >= 512) return launch_gemv_wide<8>(g, st);
        if (K >= 256) return launch_gemv_wide<4>(g, st);
        switch (K) {
        case 16: return launch_gemv_short_wide<16>(g, st);
        case 32: return launch_gemv_short_wide<32>(g, st);
        case 64: return launch_gemv_short_wide<64>(g, st);
        default: return launch_gemv_short_wide<128>(g, st);
        }
    }
    if ((fixed_mode == 6 || fixed_mode == 7) && K < 256) return hipErrorInvalidValue;   // (32-bit words: long rows only — qg_api.hip keeps short rows on the 64-bit form)
    switch (K) {
    case 16: return launch_gemv_short<16>(g, st);
    case 32: return launch_gemv_short<32>(g, st);
    case 64: return launch_gemv_short<64>(g, st);
    case 128: return launch_gemv_short<128>(g, st);
    default: break;
    }
    // (64 leaves per lane were measured first: v[64] plus the 64 prefetch registers spill, 0.68 ms for 65536 x 4096)
    // leaves per lane: 16 measured best at 65536 x 4096 (0.277 ms, 3.9 TB/s; 32 leaves 0.48 ms: the prefetch registers
    // spill under the 128-register cap of 16 waves); QG_GEMV_CH overrides for A/B runs (tools/measure_reduce.py)
#ifdef QG_DIAG
    static const int force_ch = getenv("QG_GEMV_CH") ? atoi(getenv("QG_GEMV_CH")) : 0;
#else
    constexpr int force_ch = 0;
#endif
    const int ch = force_ch ? force_ch : 16;
    if (ch >= 32 && K >= 2048) return launch_gemv<32>(g, st);
    if (ch >= 16 && K >= 1024) return launch_gemv<16>(g, st);
    if (ch >= 8 && K >= 512) return launch_gemv<8>(g, st);
    return launch_gemv<4>(g, st);
}
