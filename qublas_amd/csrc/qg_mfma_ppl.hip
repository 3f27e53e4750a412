// qg_mfma_ppl.hip — multi-limb linear class (operands of 9..23 storage bits as balanced base-256 int8 limbs) on 128x128 output
// tiles, 64-byte k-tiles, with the two wave groups of a workgroup taking turns on the matrix cores — the scheme of
// qg_mfma_pp.hip carried over to the limb products.
//
// Arithmetic as k_mfma16<LA,LB> (qg_mfma.hip): LA*LB MFMAs (v_mfma_i32_16x16x64_i8) per k-tile and 16x16 output tile into
// LA+LB-1 int32 accumulator sets, one per limb weight; 64-bit recombination sum_w acc_w 256^w and ONE round + overflow into C's
// format (converting constructor, /root/reference/include/QuBLAS.h:2398-2411) in the epilogue.
//
// A workgroup is 8 waves (2 x 4), a wave owns 64 x 32 outputs (4 x 2 tiles of 16 x 16).  A k-tile is walked in LA phases, ONE
// LIMB PLANE OF A each:
//     phase l   LOAD : ds_read_b128 the 4 fragments of A's plane l (phase 0 also the 2*LB fragments of all planes of B),
//                      issue 2 LDS-DMA pieces, counted vmcnt, lgkmcnt(0), s_barrier
//               MFMA : A_l x (B_0 .. B_{LB-1}) on the wave's 8 tiles = 8*LB MFMAs back to back,             s_barrier
// and waves 4-7 run one barrier interval behind waves 0-3: on every SIMD one wave feeds the matrix pipe while its partner reads
// LDS and issues DMA.  A phase reads ONLY plane l of A (and, in phase 0, B), so the LDS slot of a plane is free once its phase
// has passed in both groups and is refilled at once with the same plane of k-tile kt+2 (two buffers of LA+LB planes of
// 128 rows x 64 bytes = 8 KB each: 96 KB for 3 x 3).  Per wave and k-tile: LA + LB pieces of 1 KiB, 2 per phase (3 x 3 and 2 x 2).
//     3 x 3:  phase 0 issues A1, A2 of k-tile kt+1;  phase 1: A0, B0 of kt+2;  phase 2: B1, B2 of kt+2
//             waits (all but the N youngest):         phase 0: 7 (A1 of kt);   phase 1: 8 (A2 of kt);   phase 2: 6 (A0, B of kt+1)
//     2 x 2:  phase 0 issues B1, A1 of k-tile kt+1;  phase 1: A0, B0 of kt+2
//             waits:                                  phase 0: 4 (A1 of kt);   phase 1: 3 (A0, B0, B1 of kt+1)
// (Two phases per k-tile — planes 0 and 1 of A, then plane 2: four barriers instead of six — measured the same within noise,
// 0.364 against 0.361 ms at 4096^3, profiles/r04_ppl_two_vs_three_phases.jsonl: the k-loop already runs 2 304 MFMA-issue cycles in
// about 2 500.  Not kept.)
// Ordering rules as in qg_mfma_pp.hip: a wave waits for its own pieces of a plane BEFORE the barrier that closes the LOAD
// interval preceding the first read of that plane by group 0, and retires its fragment reads (lgkmcnt(0)) BEFORE the barrier
// that closes its LOAD interval.  Workgroups are persistent (one per CU, a list of tiles each) and the LDS-DMA pipeline runs
// across tile boundaries; both groups run a tile's epilogue at the same time.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "qg_kernels.h"
#include "qg_step_all.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

#define QG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define QG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int TM = 128, TN = 128, BK = 64;
constexpr int PLANE = TM * BK;   // one limb plane of a (row tile, k tile) block: 128 rows x 64 bytes = 8 pieces of 1 KiB

__device__ __forceinline__ void tile_of(int w, int tiles_m, int tiles_n, int& tile_m, int& tile_n)
{
    constexpr int GM = 8;
    const int grp = w / (GM * tiles_n);
    const int first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int rem = w - grp * (GM * tiles_n);
    tile_m = first_m + rem % gsz;
    tile_n = rem / gsz;
}

// LA x LB limbs computed, SA x SB planes stored per operand (3 x 3 storage with empty third planes runs as 2 x 2: plane masks,
// k_mfma_ppl below).  FAST: truncation (TRN::TCPL, right shift d >= 0) + SAT::TCPL as a 64-bit shift and a clamp; otherwise the
// general routine.  CB: container bytes of C (4 or 8; narrower C takes the lock-step kernel).
template <int LA, int LB, int SA, int SB, bool FAST, int CB>
__device__ __forceinline__ void ppl_body(const QMfmaArgs& g)
{
    static_assert((LA == 3 && LB == 3) || (LA == 2 && LB == 2), "issue schedule written out for 3 x 3 and 2 x 2 limbs");
    constexpr int NW = LA + LB - 1;
    constexpr int BUF = (LA + LB) * PLANE;   // A planes 0..LA-1, then B planes 0..LB-1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // group 0 = waves 0-3 (rows 0-63 of the tile), group 1 = waves 4-7

    const int tiles_m = (int)(g.Mp / TM), tiles_n = (int)(g.Np / TN);
    const int nwg = tiles_m * tiles_n;
    int w_first, w_step, n_my;
    {
        const int q = nwg / 8, r = nwg % 8, x = blockIdx.x % 8;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0), j = blockIdx.x / 8, P = gridDim.x / 8;
        w_first = start + j;
        w_step = P;
        n_my = j < cnt ? (cnt - j + P - 1) / P : 0;
    }
    if (n_my == 0) return;

    const int nk = (int)(g.Kp / BK);
    const int64_t panel_a = (int64_t)nk * SA * PLANE, panel_b = (int64_t)nk * SB * PLANE;
    const uint32_t lane_off = (uint32_t)(wave * 1024 + lane * 16);   // this wave's piece of a plane
    char* const lds_w = smem + wave * 1024;
    struct Cursor { const int8_t* a; const int8_t* b; int kt, ti; };
    auto cursor_at_tile = [&](int ti) {
        int tm, tn;
        tile_of(w_first + ti * w_step, tiles_m, tiles_n, tm, tn);
        return Cursor{g.A + tm * panel_a, g.B + tn * panel_b, 0, ti};
    };
    auto advance = [&](Cursor c) {
        if (c.kt + 1 < nk) return Cursor{c.a + SA * PLANE, c.b + SB * PLANE, c.kt + 1, c.ti};
        if (c.ti + 1 < n_my) return cursor_at_tile(c.ti + 1);
        return c;   // past the end: the last k-tile again, into slots nobody reads (branch-free issue keeps the vmcnt counts)
    };
    auto issue_a = [&](int buf_off, int l, const Cursor& c) {
        __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(c.a + l * PLANE + lane_off), QG_LDS_PTR(lds_w + buf_off + l * PLANE), 16, 0, 0);
    };
    auto issue_b = [&](int buf_off, int l, const Cursor& c) {
        __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(c.b + l * PLANE + lane_off), QG_LDS_PTR(lds_w + buf_off + (LA + l) * PLANE), 16, 0, 0);
    };

    v4i acc[NW][4][2];
    // fragment of v_mfma_i32_16x16x64_i8: lane l holds row (l & 15), bytes [16 (l >> 4), +16) of the 64-byte k-step; LDS image:
    // 64-byte rows, chunk c of row r at slot c ^ {0,2,3,1}[(r / 4) % 4] (swz<64>, qg_mfma.hip) — a lane constant here
    const int fr = lane & 15, fq = lane >> 4;
    const int chunk = (fq ^ ((0x78 >> (2 * (fr >> 2))) & 3)) * 16;
    const int a_lane = (wm * 64 + fr) * BK + chunk;
    const int b_lane = LA * PLANE + (wn * 32 + fr) * BK + chunk;
    v4i fa[4], fb[LB][2];
    auto read_a = [&](int buf_off, int l) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const v4i*)(smem + buf_off + l * PLANE + i * (16 * BK) + a_lane);
    };
    auto read_b = [&](int buf_off) {
#pragma unroll
        for (int l = 0; l < LB; ++l)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[l][j] = *(const v4i*)(smem + buf_off + l * PLANE + j * (16 * BK) + b_lane);
    };
    auto mfmas = [&](int l) {
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[l + lb][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[lb][j], acc[l + lb][i][j], 0, 0, 0);
    };
#define QG_LOAD_DONE(N)                                             \
    do {                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);                          \
        __builtin_amdgcn_s_barrier();                               \
        __builtin_amdgcn_sched_barrier(0);                          \
    } while (0)
    auto mfma_phase = [&](int l) {
        __builtin_amdgcn_s_setprio(1);
        mfmas(l);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue (once per workgroup): k-tile 0 whole and, of k-tile 1, the planes that the loop does not issue in its first phase
    Cursor cur1 = cursor_at_tile(0);
    issue_a(0, 0, cur1);
#pragma unroll
    for (int l = 0; l < LB; ++l) issue_b(0, l, cur1);
#pragma unroll
    for (int l = 1; l < LA; ++l) issue_a(0, l, cur1);
    cur1 = advance(cur1);
    issue_a(BUF, 0, cur1);
    issue_b(BUF, 0, cur1);
    if constexpr (LA == 3) {
        issue_b(BUF, 1, cur1);
        issue_b(BUF, 2, cur1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A0 and all of B of k-tile 0 (10 issued)
    } else {
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // A0, B0, B1 of k-tile 0 (6 issued)
    }
    Cursor cur2 = advance(cur1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur = 0;
    for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[w][i][j][e] = 0;
        if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind
        __builtin_amdgcn_sched_barrier(0);

        for (int kt = 0; kt < nk; ++kt) {
            const int oth = BUF - cur;
            if constexpr (LA == 3) {
                // phase 0: A plane 0, all of B
                read_a(cur, 0);
                read_b(cur);
                issue_a(oth, 1, cur1);
                issue_a(oth, 2, cur1);
                QG_LOAD_DONE(7);      // A1 of this k-tile is in
                mfma_phase(0);
                // phase 1
                read_a(cur, 1);
                issue_a(cur, 0, cur2);
                issue_b(cur, 0, cur2);
                QG_LOAD_DONE(8);      // A2 of this k-tile is in
                mfma_phase(1);
                // phase 2
                read_a(cur, 2);
                issue_b(cur, 1, cur2);
                issue_b(cur, 2, cur2);
                QG_LOAD_DONE(6);      // A0 and B of the next k-tile are in
                mfma_phase(2);
            } else {
                read_a(cur, 0);
                read_b(cur);
                issue_b(oth, 1, cur1);
                issue_a(oth, 1, cur1);
                QG_LOAD_DONE(4);      // A1 of this k-tile is in
                mfma_phase(0);
                read_a(cur, 1);
                issue_a(cur, 0, cur2);
                issue_b(cur, 0, cur2);
                QG_LOAD_DONE(3);      // A0, B0, B1 of the next k-tile are in
                mfma_phase(1);
            }
            cur1 = cur2;
            cur2 = advance(cur2);
            cur = oth;
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last barrier: both groups are level again
        __builtin_amdgcn_sched_barrier(0);

        // epilogue: recombine the limb weights in 64 bits, one round + overflow, stores of 4 consecutive rows
        // C/D of the 16x16 MFMA: col = lane & 15, rows 4 (lane >> 4) + e; packed C is column-major inside the tile
        int tile_m, tile_n;
        tile_of(w_first + ti * w_step, tiles_m, tiles_n, tile_m, tile_n);
        const QStep st = g.to_c;
        char* C = (char*)g.C;
        const int64_t tile_base = ((int64_t)tile_m * tiles_n + tile_n) * TM * TN;
        [[maybe_unused]] const int sh = st.d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t s[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int64_t x = (int64_t)acc[NW - 1][i][j][e];
#pragma unroll
                    for (int w = NW - 2; w >= 0; --w) x = x * 256 + (int64_t)acc[w][i][j][e];
                    if (g.rsA)   // centred operands (QPackedGeom::offs; k_mfma): the centres go out with the row sums (wrapping arithmetic)
                        x = (int64_t)((uint64_t)x + (uint64_t)g.corr - (uint64_t)g.biasA * (uint64_t)g.rsB[(int64_t)tile_n * TN + wn * 32 + j * 16 + fr] -
                                      (uint64_t)g.biasB * (uint64_t)g.rsA[(int64_t)tile_m * TM + wm * 64 + i * 16 + 4 * fq + e]);
                    s[j * 4 + e] = x;
                }
            if constexpr (FAST) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const int64_t x = s[o] >> sh, y = x < st.lo ? st.lo : x;
                    s[o] = y > st.hi ? st.hi : y;
                }
            } else {
                qg_step_all<int64_t, 8>(s, st);
            }
            const int row0 = wm * 64 + i * 16 + 4 * fq;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 32 + j * 16 + fr;
                const int64_t* q = s + j * 4;
                if (g.c_host) {   // the reference layout itself (wave-uniform choice): element (r, c) at r + c * ld
                    const int64_t gr = (int64_t)tile_m * TM + row0, gc = (int64_t)tile_n * TN + col;
                    if (gc < g.c_N) {
                        using E = std::conditional_t<CB == 4, int32_t, int64_t>;
                        E* dst = (E*)C + gc * g.c_ld + gr;
                        if (gr + 3 < g.c_M && g.c_vec) {
                            if constexpr (CB == 4) *(int4*)dst = make_int4((int)q[0], (int)q[1], (int)q[2], (int)q[3]);
                            else { *(longlong2*)dst = make_longlong2(q[0], q[1]); *(longlong2*)(dst + 2) = make_longlong2(q[2], q[3]); }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (gr + e < g.c_M) dst[e] = (E)q[e];
                        }
                    }
                    continue;
                }
                const int64_t base = tile_base + (int64_t)col * TM + row0;
                if constexpr (CB == 4) {
                    *(int4*)(C + base * 4) = make_int4((int)q[0], (int)q[1], (int)q[2], (int)q[3]);
                } else {
                    int64_t* p = (int64_t*)(C + base * 8);
                    *(longlong2*)p = make_longlong2(q[0], q[1]);
                    *(longlong2*)(p + 2) = make_longlong2(q[2], q[3]);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped refills must land before the LDS is handed on
#undef QG_LOAD_DONE
}

// Plane masks (qg_mfma.hip): a 3-plane operand whose third plane is zero everywhere needs only 2 x 2 limb products.  The
// lock-step kernels launch a pair of kernels of which one returns at once (2-3 us per call for the idle partner); here ONE
// persistent kernel reads the masks and takes one of two complete code paths — a wave-uniform branch at the very top, not a
// masked MFMA sequence (that was measured 2.4x slower: it spills).
template <bool FAST, int CB>
__global__ __launch_bounds__(512) void k_mfma_ppl(QMfmaArgs g)
{
    const unsigned ma = qg_plane_mask(g.maskA);
    const unsigned mb = qg_plane_mask(g.maskB);
    if (((ma | mb) & 4u) == 0) ppl_body<2, 2, 3, 3, FAST, CB>(g);
    else ppl_body<3, 3, 3, 3, FAST, CB>(g);
}

// two-limb operands on two-plane storage (13..15 storage bits that are not Karatsuba-eligible, and the stacked real GEMM of the
// complex linear class): the 2 x 2 path on its own
template <bool FAST, int CB>
__global__ __launch_bounds__(512) void k_mfma_ppl22(QMfmaArgs g)
{
    ppl_body<2, 2, 2, 2, FAST, CB>(g);
}

template <bool FAST, int CB>
hipError_t launch_ppl22(const QMfmaArgs& a, unsigned grid, hipStream_t st)
{
    constexpr int lds = 2 * 4 * PLANE;
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (qg_lds_attr)
    if (hipError_t e = qg_lds_attr((const void*)k_mfma_ppl22<FAST, CB>, lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mfma_ppl22<FAST, CB>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}

template <bool FAST, int CB>
hipError_t launch_ppl(const QMfmaArgs& a, unsigned grid, hipStream_t st)
{
    constexpr int lds = 2 * 6 * PLANE;   // the 3 x 3 path's two buffers
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (qg_lds_attr)
    if (hipError_t e = qg_lds_attr((const void*)k_mfma_ppl<FAST, CB>, lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mfma_ppl<FAST, CB>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_ppl_modes(int limbs, const QMfmaArgs& a, unsigned grid, hipStream_t st)
{
    const QStep& q = a.to_c;
    const bool fast = !q.identity && q.O == QG_SAT_TCPL && q.Q == QG_TRN_TCPL && q.d >= 0;
    if (limbs == 2) {
        if (a.cbytes == 4) return fast ? launch_ppl22<true, 4>(a, grid, st) : launch_ppl22<false, 4>(a, grid, st);
        if (a.cbytes == 8) return fast ? launch_ppl22<true, 8>(a, grid, st) : launch_ppl22<false, 8>(a, grid, st);
        return hipErrorInvalidValue;
    }
    if (a.cbytes == 4) return fast ? launch_ppl<true, 4>(a, grid, st) : launch_ppl<false, 4>(a, grid, st);
    if (a.cbytes == 8) return fast ? launch_ppl<true, 8>(a, grid, st) : launch_ppl<false, 8>(a, grid, st);
    return hipErrorInvalidValue;
}

} // namespace

bool qg_mfma_ppl_applies(int LA, int LB, const QMfmaArgs& a)
{
    if (!((LA == 3 && LB == 3) || (LA == 2 && LB == 2)) || a.has_ep || a.kara || a.variant != 10) return false;
    if (a.cbytes != 4 && a.cbytes != 8) return false;
    return (a.Mp / TM) * (a.Np / TN) >= 256;   // persistent: one workgroup per CU with at least a tile each
}

hipError_t qg_launch_mfma_ppl(int limbs, const QMfmaArgs& a, hipStream_t st)
{
    const int64_t blocks = (a.Mp / TM) * (a.Np / TN);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll || a.Kp % BK || a.Mp % TM || a.Np % TN) return hipErrorInvalidValue;
    int dev = 0, cus = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
    int64_t grid = cus / 8 * 8;
    if (grid < 8) grid = 8;
    if (grid > blocks) grid = (blocks + 7) / 8 * 8;
    return launch_ppl_modes(limbs, a, (unsigned)grid, st);
}
