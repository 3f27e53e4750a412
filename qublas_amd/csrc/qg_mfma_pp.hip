// qg_mfma_pp.hip — single-limb linear class (operands of at most 8 storage bits) on 256x256 output tiles with 128-byte
// k-tiles: the two wave groups of a workgroup take turns on the matrix cores ("ping-pong").
//
// Same arithmetic as k_mfma16<1,1> (qg_mfma.hip): the exact int32 dot product on v_mfma_i32_16x16x64_i8 and ONE round +
// overflow into C's format (converting constructor, /root/reference/include/QuBLAS.h:2398-2411) — what differs is who
// issues what when.  A workgroup is 8 waves; waves 0-3 (group 0) and waves 4-7 (group 1) sit pairwise on the CU's four SIMDs.
// A k-tile is walked in four phases, one 64x32 quadrant of the wave's 128x64 outputs each:
//     LOAD  : ds_read_b128 the quadrant's fragments, issue 2 LDS-DMA pieces, wait (counted vmcnt, lgkmcnt(0)), s_barrier
//     MFMA  : 16 MFMAs back to back (256 matrix-pipe cycles),                                                      s_barrier
// and group 1 runs ONE barrier interval behind group 0, so in every interval one wave of each SIMD feeds the matrix pipe while
// its partner reads LDS and issues DMA: the pipe never waits for a fragment read, a DMA issue or a barrier of its own wave.
// (The lock-step kernel it replaces for large problems ran both waves of a SIMD through the same sequence at the same time and
// measured 0.42 matrix-pipe utilisation at 16384^2 x 4096, profiles/r03m_c2L.json.)
//
// LDS: 2 buffers x 4 half-tiles (A rows 0-127, A rows 128-255, B rows 0-127, B rows 128-255 of the tile; 16 KB each = 128 rows x
// 128 bytes) = 128 KB.  Wave (wr, wc) owns output rows {q*128 + wr*64 + [0,64)} and columns {q*128 + wc*32 + [0,32)}, q = 0, 1:
// a phase with row half qi / column half qj reads ONLY half-tiles A[qi] / B[qj], so a half-tile's slot is free as soon as
// the phase that reads it has passed in both groups, and is refilled (for k-tile kt+2) right then:
//     phase  quadrant  reads (ds_read_b128 per wave)   LDS-DMA issued (2 pieces per wave)
//       0    (0,0)     A[0] (8)  B[0] (4)              A[1] of k-tile kt+1
//       1    (0,1)     B[1] (4)                        A[0] of k-tile kt+2
//       2    (1,1)     A[1] (8)                        B[0] of k-tile kt+2
//       3    (1,0)     -  (B[0] stays in registers)    B[1] of k-tile kt+2
// Five half-tiles stay in flight per wave (vmcnt(10)); a half-tile has more than a k-tile of MFMA time to land.
// Ordering rules (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"): every wave waits for its own
// pieces of a half-tile (counted vmcnt) BEFORE the barrier that ends the LOAD interval preceding the first read of that
// half-tile by group 0; every wave retires its fragment reads (lgkmcnt(0)) BEFORE the barrier that ends its LOAD interval, so
// a slot read in phase p is provably idle two intervals later, when group 0 refills it.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "qg_kernels.h"
#include "qg_step_all.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

#define QG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define QG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int TM = 256, TN = 256, BK = 128;
constexpr int HALF = 128 * BK;     // one half-tile: 128 rows of 128 bytes
constexpr int BUF = 4 * HALF;      // A[0] A[1] B[0] B[1]
constexpr int TILE_BYTES = TM * BK;   // a (row tile, k tile) block of a packed operand
constexpr int PP_DMA_SPLIT = 1;       // PH2: LDS-DMA pieces per wave in phase A / phase B: 0 = 4 / 4, 1 = 2 / 6 (shipped), 2 = 0 / 8
constexpr bool PP_TWO_PHASES = true;  // k_mfma_pp's default phase structure (PH2): measured 2-6 % faster than four phases

// the walk over the output tiles (as k_mfma16): groups of 8 tile rows, column by column inside a group
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

// PERSIST: one workgroup per CU walks a list of tiles, and the LDS-DMA pipeline runs straight across tile boundaries — the
// refills issued during a tile's last two k-tiles fetch the NEXT tile's first k-tiles, so there is no prologue latency, no
// workgroup launch and no pipeline drain between tiles; what remains at a boundary is the epilogue, which both wave groups
// run at the same time (group 0 waits one barrier for group 1's last MFMA interval, group 1 re-opens its one-interval lag
// behind group 0 afterwards).  Without PERSIST a workgroup's list has one tile (the launch-per-tile form, kept for A/B).
// STAMP (diagnostic build only): shader-clock stamps of the first 8 tiles of every wave, kept in the 32 KB of LDS behind
// the two buffers and copied to g.dbg at the end (no vector-memory traffic while the kernel runs); tools/stamps_pp.py.
//   per tile: 0 before the k-loop, 1 after phase 0's LOAD interval of the first k-tile, 2 after the k-loop, 3 after the
//   levelling barrier (epilogue start), 4 after the epilogue's last store was issued, 5 after phase 3's LOAD interval of the
//   first k-tile (the first counted vmcnt wait that has the epilogue's stores in front of it is the one of phase 0)
// FAST, CB (chosen by the launcher from the descriptor): the one conversion into C's format is specialised at compile time —
// FAST = truncation (TRN::TCPL, right shift d >= 0) + SAT::TCPL, a shift and a clamp per value (what a default-mode C type
// asks for: configurations 2 and 4); otherwise the general routine with its wave-uniform mode switches — and so is the
// container size CB (1, 2, 4 or 8 bytes), so that the epilogue is straight-line code.
// (Measured and dropped, profiles/r2h_measure_pp.jsonl: issuing the last 2 / 4 / 8 MFMAs of an interval BEHIND its closing
// barrier, beside the partner group's first MFMAs, to cover the barrier release: 0.91 ms against 0.77 ms at 16384^2 x 4096 —
// two waves feeding one SIMD's matrix pipe at the same time cost far more than the release does.  A start offset between the
// workgroups of an XCD, to spread the epilogues' stores in time: no gain for 1- and 2-byte C once the epilogue was straight-
// line code, profiles/r2g_measure_pp.jsonl.)
// PH2 (the default): two phases of 32 MFMAs per k-tile instead of the four of 16 described at the top of this file — half the
// barriers, intervals of 512 matrix-pipe cycles: phase A = row half 0 (reads A[0], B[0], B[1]: 16 ds_read_b128; quadrants (0,0),
// (0,1)), phase B = row half 1 (reads A[1]: 8; quadrants (1,1), (1,0)).  LDS-DMA (DM = 1, shipped): phase A, which carries
// 16 of the 24 fragment reads, issues only A[1] of k-tile kt+1 (2 pieces per wave), phase B issues A[0], B[0], B[1] of kt+2 (6);
// waits vmcnt(8) in both (A[1] of this k-tile; A[0], B[0], B[1] of the next).  (DM = 0: 4 + 4 pieces — B[1], A[1] of kt+1 in phase A —
// with waits 8 / 6; DM = 2: all 8 in phase B.  16384^2 x 4096, same process: 0.776 / 0.767 / 0.774 ms for 4+4 / 2+6 / 0+8,
// profiles/r04_pp_dma_split.jsonl.)  Same ordering rules.  16384^2 x 4096: 0.781 against 0.800 ms, 8192^2 x 4096: 0.193 against 0.204 ms, a 2048-row shard
// 0.099 against 0.106 ms on four phases, same process (profiles/r04_pp_two_phases.jsonl; QG_PP_PH4 in the diagnostic build).
template <bool PERSIST, bool FAST, int CB, bool STAMP = false, bool PH2 = PP_TWO_PHASES, int DM = PP_DMA_SPLIT>
__global__ __launch_bounds__(512) void k_mfma_pp(QMfmaArgs g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware tile order: block ids b and b + 8 share an XCD (observed dispatch; speed only), so each of the 8 residue
    // classes gets a contiguous run of the walk.  PERSIST: the P = gridDim.x / 8 workgroups of a class take the run's tiles
    // round-robin, i.e. in every round a class works on P consecutive tiles of the walk (8 tile rows x P / 8 columns).
    const int tiles_m = (int)(g.Mp / TM), tiles_n = (int)(g.Np / TN);
    const int nwg = tiles_m * tiles_n;
    int w_first, w_step, n_my;
    {
        const int q = nwg / 8, r = nwg % 8, x = blockIdx.x % 8;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        if constexpr (PERSIST) {
            const int cnt = q + (x < r ? 1 : 0), j = blockIdx.x / 8, P = gridDim.x / 8;
            w_first = start + j;
            w_step = P;
            n_my = j < cnt ? (cnt - j + P - 1) / P : 0;
        } else {
            w_first = start + blockIdx.x / 8;
            w_step = 0;
            n_my = 1;
        }
    }
    if (n_my == 0) return;   // (the whole workgroup, before any barrier)

    const int nk = (int)(g.Kp / BK);
    const int64_t panel = (int64_t)nk * TILE_BYTES;   // all k-tiles of one row tile
    // this wave's two pieces of a half-tile: 1 KiB at wave * 1024 and at (wave + 8) * 1024
    const uint32_t lane_off = (uint32_t)(wave * 1024 + lane * 16);
    char* const lds_w = smem + wave * 1024;
    // the issue cursor: k-tile `kt` of tile number `ti` of this workgroup's list; past the end of the list it stays on the last
    // k-tile, which is then fetched again into slots nobody reads (branch-free issue: the vmcnt counts assume every phase issues)
    struct Cursor { const int8_t* a; const int8_t* b; int kt, ti; };
    auto cursor_at_tile = [&](int ti) {
        int tm, tn;
        tile_of(w_first + ti * w_step, tiles_m, tiles_n, tm, tn);
        return Cursor{g.A + tm * panel, g.B + tn * panel, 0, ti};
    };
    auto advance = [&](Cursor c) {
        if (c.kt + 1 < nk) return Cursor{c.a + TILE_BYTES, c.b + TILE_BYTES, c.kt + 1, c.ti};
        if (c.ti + 1 < n_my) return cursor_at_tile(c.ti + 1);
        return c;
    };
    // slot: 0 A[0], 1 A[1], 2 B[0], 3 B[1]
    auto issue = [&](int buf_off, int slot, const Cursor& c) {
        const int8_t* src = (slot < 2 ? c.a : c.b) + (slot & 1) * HALF + lane_off;
        char* dst = lds_w + buf_off + slot * HALF;
        __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(src), QG_LDS_PTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(src + 8192), QG_LDS_PTR(dst + 8192), 16, 0, 0);
    };

    v4i acc[2][2][4][2];   // [row half][column half][16-row tile][16-column tile]
    [[maybe_unused]] uint32_t* const stamps = (uint32_t*)(smem + 2 * BUF) + (threadIdx.x >> 6) * 64;
    [[maybe_unused]] auto stamp = [&](int ti, int idx) {
        if constexpr (STAMP) {
            if (ti < 8) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0) stamps[ti * 8 + idx] = t;
            }
        }
    };
    if constexpr (STAMP) stamps[threadIdx.x & 63] = 0;

    // fragment of v_mfma_i32_16x16x64_i8: lane l holds row (l & 15), bytes [16 (l >> 4), +16) of the 64-byte k-step.
    // LDS image: 128-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (swz<128>, qg_mfma.hip; the pack kernels
    // write it); every row this lane reads is (a multiple of 16) + fr, so its swizzle term is a lane constant.
    const int fr = lane & 15, fq = lane >> 4;
    const int c0 = ((fq ^ (fr >> 1)) & 7) * 16;   // k-step 0: chunk fq; k-step 1 (chunk 4 + fq): c0 ^ 64
    const int a_lane = (wr * 64 + fr) * BK + c0;
    const int b_lane = 2 * HALF + (wc * 32 + fr) * BK + c0;
    v4i fa[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](int buf_off, int qi) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fa[t][ks] = *(const v4i*)(smem + buf_off + qi * HALF + t * (16 * BK) + (a_lane ^ (ks * 64)));
    };
    auto read_b = [&](v4i (&fb)[2][2], int buf_off, int qj) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fb[u][ks] = *(const v4i*)(smem + buf_off + qj * HALF + u * (16 * BK) + (b_lane ^ (ks * 64)));
    };
    auto mfmas = [&](v4i (&c)[4][2], const v4i (&fb)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    c[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[t][ks], fb[u][ks], c[t][u], 0, 0, 0);
    };
    // end of a LOAD interval: this wave's pieces of the half-tile read next are in (all but the 10 youngest vector-memory
    // operations: the epilogue's stores of the previous tile count too and are older than any piece that may still fly, so
    // the wait is at worst early), its own fragment reads are back, then the workgroup barrier
    auto load_done_n = [&](auto n) {   // PH2: counts differ per phase
        if constexpr (decltype(n)::value == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_done = [&](bool wait_dma) {
        if (wait_dma) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue (once per workgroup): k-tile 0 whole and A[0], B[0], B[1] of k-tile 1 in flight (7 half-tiles, in the order
    // they are read)
    Cursor cur1 = cursor_at_tile(0);   // k-tile g + 1 (its A[1] is issued in phase 0 of k-tile g)
    issue(0, 0, cur1);
    issue(0, 2, cur1);
    issue(0, 3, cur1);
    issue(0, 1, cur1);
    cur1 = advance(cur1);
    issue(BUF, 0, cur1);
    issue(BUF, 2, cur1);
    constexpr bool D26 = PH2 && DM == 1, D08 = PH2 && DM == 2;
    if constexpr (!PH2 || D26 || D08) issue(BUF, 3, cur1);
    Cursor cur2 = advance(cur1);       // k-tile g + 2 (A[0], B[0], B[1] issued in phases 1, 2, 3 of k-tile g)
    if constexpr (D26 || D08) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A[0], B[0], B[1] of k-tile 0 (14 pieces issued)
    else if constexpr (PH2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A[0], B[0], B[1] of k-tile 0 (12 pieces issued)
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");               // A[0], B[0] of k-tile 0 (14 issued)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur = 0;   // LDS buffer of the k-tile being computed
    for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[a][b][t][u][e] = 0;
        if (wr == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one interval behind
        __builtin_amdgcn_sched_barrier(0);
        stamp(ti, 0);

        for (int kt = 0; kt < nk; ++kt) {
            const int oth = BUF - cur;
            if constexpr (PH2) {
                // phase A: row half 0
                read_a(cur, 0);
                read_b(fb0, cur, 0);
                read_b(fb1, cur, 1);
                if constexpr (!D26 && !D08) issue(oth, 3, cur1);
                if constexpr (!D08) issue(oth, 1, cur1);
                if constexpr (D08) load_done_n(std::integral_constant<int, 6>{});   // A[1] of this k-tile is in (6 younger pieces)
                else load_done_n(std::integral_constant<int, 8>{});                 // A[1] of this k-tile is in
                if constexpr (STAMP) { if (kt == 0) stamp(ti, 1); }
                __builtin_amdgcn_s_setprio(1);
                mfmas(acc[0][0], fb0);
                mfmas(acc[0][1], fb1);
                __builtin_amdgcn_s_setprio(0);
                mfma_done();
                // phase B: row half 1
                read_a(cur, 1);
                if constexpr (D08) issue(oth, 1, cur1);           // A[1] of k-tile kt+1 (first: it is needed first)
                issue(cur, 0, cur2);
                issue(cur, 2, cur2);
                if constexpr (D26 || D08) { issue(cur, 3, cur2); load_done_n(std::integral_constant<int, 8>{}); }
                else load_done_n(std::integral_constant<int, 6>{});   // A[0], B[0], B[1] of the next k-tile are in
                if constexpr (STAMP) { if (kt == 0) stamp(ti, 5); }
                __builtin_amdgcn_s_setprio(1);
                mfmas(acc[1][1], fb1);
                mfmas(acc[1][0], fb0);
                __builtin_amdgcn_s_setprio(0);
                mfma_done();
                cur1 = cur2;
                cur2 = advance(cur2);
                cur = oth;
                continue;
            }
            // phase 0: quadrant (0, 0)
            read_a(cur, 0);
            read_b(fb0, cur, 0);
            issue(oth, 1, cur1);
            load_done(true);                                  // B[1] of this k-tile is in
            if constexpr (STAMP) { if (kt == 0) stamp(ti, 1); }
            __builtin_amdgcn_s_setprio(1);
            mfmas(acc[0][0], fb0);
            __builtin_amdgcn_s_setprio(0);
            mfma_done();
            // phase 1: quadrant (0, 1)
            read_b(fb1, cur, 1);
            issue(cur, 0, cur2);
            load_done(true);                                  // A[1] of this k-tile is in
            __builtin_amdgcn_s_setprio(1);
            mfmas(acc[0][1], fb1);
            __builtin_amdgcn_s_setprio(0);
            mfma_done();
            // phase 2: quadrant (1, 1)
            read_a(cur, 1);
            issue(cur, 2, cur2);
            load_done(false);
            __builtin_amdgcn_s_setprio(1);
            mfmas(acc[1][1], fb1);
            __builtin_amdgcn_s_setprio(0);
            mfma_done();
            // phase 3: quadrant (1, 0)
            issue(cur, 3, cur2);
            load_done(true);                                  // A[0], B[0] of the next k-tile are in
            if constexpr (STAMP) { if (kt == 0) stamp(ti, 5); }
            __builtin_amdgcn_s_setprio(1);
            mfmas(acc[1][0], fb0);
            __builtin_amdgcn_s_setprio(0);
            mfma_done();
            cur1 = cur2;
            cur2 = advance(cur2);
            cur = oth;
        }
        stamp(ti, 2);
        if (wr == 0) __builtin_amdgcn_s_barrier();           // pairs with group 1's last barrier: both groups are level again
        __builtin_amdgcn_sched_barrier(0);
        stamp(ti, 3);

        // epilogue: one round + overflow, stored as runs of 4 rows (packed C is column-major inside the tile)
        // C/D of the 16x16 MFMA: col = lane & 15, rows 4 (lane >> 4) + e
        int tile_m, tile_n;
        tile_of(w_first + ti * w_step, tiles_m, tiles_n, tile_m, tile_n);
        const QStep st = g.to_c;
        [[maybe_unused]] const int sh = st.d;
        [[maybe_unused]] const int32_t clo = (int32_t)st.lo, chi = (int32_t)st.hi;
        auto convert16 = [&](int32_t (&v)[16]) {
            if constexpr (FAST) {
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    v[o] = qg_clamp_i32(v[o] >> sh, clo, chi);
                }
            } else {
                qg_step_all<int32_t, 16>(v, st);
            }
        };
        // centred operands (QPackedGeom::offs; k_mfma): sum a b = acc - biasB rsA[row] - biasA rsB[col] + K biasA biasB needs more than
        // 32 bits, its image in C's format (at most 31 bits on this path: qg_api.hip) does not.  rowf / colf: tile-local row and
        // column of value o of the 16
        auto convert16c = [&](int32_t (&v)[16], auto rowf, auto colf) {
            if (!g.rsA) { convert16(v); return; }
            const int64_t* ra = g.rsA + (int64_t)tile_m * TM;
            const int64_t* rb = g.rsB + (int64_t)tile_n * TN;
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const int64_t x = (int64_t)((uint64_t)(int64_t)v[o] + (uint64_t)g.corr - (uint64_t)g.biasB * (uint64_t)ra[rowf(o)] - (uint64_t)g.biasA * (uint64_t)rb[colf(o)]);
                if constexpr (FAST) {
                    const int64_t y = x >> sh;
                    v[o] = (int32_t)(y < (int64_t)clo ? (int64_t)clo : y > (int64_t)chi ? (int64_t)chi : y);
                } else {
                    v[o] = (int32_t)qg_step<int64_t>(x, st);
                }
            }
        };
        char* C = (char*)g.C;
        const int64_t tile_base = ((int64_t)tile_m * tiles_n + tile_n) * TM * TN;
        if constexpr (CB <= 2) {
            // 1- and 2-byte containers: a lane's run of 4 rows is only 4 / 8 bytes, and 32 such stores per lane made the
            // epilogue store-issue bound.  The four lanes that share a column (lane rows fq = 0..3) hold rows 4 fq .. 4 fq + 3 of
            // each of the wave's four 16-row tiles t: a 4 x 4 transpose of packed dwords across the lane rows (2
            // v_permlane32_swap + 2 v_permlane16_swap) leaves lane row q with all 16 rows of tile t = q — one 16-byte store
            // per lane and column tile instead of four 4-byte ones, and each group of four lanes writes 64 (128) contiguous
            // bytes of the column.
            auto xpose4 = [](uint32_t (&x)[4]) {
                auto r = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false);   // rows 2, 3 of x0 <-> rows 0, 1 of x2
                x[0] = r[0]; x[2] = r[1];
                r = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false);
                x[1] = r[0]; x[3] = r[1];
                r = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false);         // rows 1, 3 of x0 <-> rows 0, 2 of x1
                x[0] = r[0]; x[1] = r[1];
                r = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false);
                x[2] = r[0]; x[3] = r[1];
            };
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int qj = 0; qj < 2; ++qj)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        int32_t s[16];   // [t][e]
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[t * 4 + e] = acc[qi][qj][t][u][e];
                        convert16c(s, [&](int o) { return qi * 128 + wr * 64 + (o >> 2) * 16 + 4 * fq + (o & 3); },
                                   [&](int) { return qj * 128 + wc * 32 + u * 16 + fr; });
                        const int col = qj * 128 + wc * 32 + u * 16 + fr;
                        const int64_t base = tile_base + (int64_t)col * TM + qi * 128 + wr * 64 + fq * 16;   // 16 rows of tile t = fq
                        if constexpr (CB == 1) {
                            uint32_t x[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {   // low bytes of four values: three v_perm_b32
                                const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)s[4 * t + 1], (uint32_t)s[4 * t], 0x0c0c0400u);
                                const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)s[4 * t + 3], (uint32_t)s[4 * t + 2], 0x0c0c0400u);
                                x[t] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                            }
                            xpose4(x);
                            *(uint4*)(C + base) = make_uint4(x[0], x[1], x[2], x[3]);
                        } else {
                            uint32_t lo[4], hi[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                lo[t] = __builtin_amdgcn_perm((uint32_t)s[4 * t + 1], (uint32_t)s[4 * t], 0x05040100u);       // low halves of two values
                                hi[t] = __builtin_amdgcn_perm((uint32_t)s[4 * t + 3], (uint32_t)s[4 * t + 2], 0x05040100u);
                            }
                            xpose4(lo);
                            xpose4(hi);
                            *(uint4*)(C + base * 2) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
                            *(uint4*)(C + base * 2 + 16) = make_uint4(lo[2], hi[2], lo[3], hi[3]);
                        }
                    }
        } else {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int32_t s[16];
#pragma unroll
                    for (int qj = 0; qj < 2; ++qj)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[(qj * 2 + u) * 4 + e] = acc[qi][qj][t][u][e];
                    convert16c(s, [&](int o) { return qi * 128 + wr * 64 + t * 16 + 4 * fq + (o & 3); },
                               [&](int o) { return (o >> 3) * 128 + wc * 32 + ((o >> 2) & 1) * 16 + fr; });
                    const int row0 = qi * 128 + wr * 64 + t * 16 + 4 * fq;
#pragma unroll
                    for (int qj = 0; qj < 2; ++qj)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int col = qj * 128 + wc * 32 + u * 16 + fr;
                            const int32_t* q = s + (qj * 2 + u) * 4;
                            if (g.c_host) {   // the reference layout itself (wave-uniform choice): element (r, c) at r + c * ld
                                const int64_t gr = (int64_t)tile_m * TM + row0, gc = (int64_t)tile_n * TN + col;
                                if (gc < g.c_N) {
                                    using E = std::conditional_t<CB == 4, int32_t, int64_t>;
                                    E* dst = (E*)C + gc * g.c_ld + gr;
                                    if (gr + 3 < g.c_M && g.c_vec) {
                                        if constexpr (CB == 4) *(int4*)dst = make_int4(q[0], q[1], q[2], q[3]);
                                        else { *(longlong2*)dst = make_longlong2((int64_t)q[0], (int64_t)q[1]); *(longlong2*)(dst + 2) = make_longlong2((int64_t)q[2], (int64_t)q[3]); }
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
                                *(int4*)(C + base * 4) = make_int4(q[0], q[1], q[2], q[3]);
                            } else {
                                int64_t* p = (int64_t*)(C + base * 8);
                                *(longlong2*)p = make_longlong2((int64_t)q[0], (int64_t)q[1]);
                                *(longlong2*)(p + 2) = make_longlong2((int64_t)q[2], (int64_t)q[3]);
                            }
                        }
                }
        }
        stamp(ti, 4);
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (g.dbg) g.dbg[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)] = stamps[threadIdx.x & 63];
    }
    // the clamped refills are still in flight: they must land before this workgroup ends and its LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

} // namespace

namespace {
template <bool PERSIST, bool FAST, int CB, bool STAMP = false>
hipError_t launch_pp(const QMfmaArgs& a, unsigned grid, int lds, hipStream_t st)
{
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (qg_lds_attr)
    if (hipError_t e = qg_lds_attr((const void*)k_mfma_pp<PERSIST, FAST, CB, STAMP>, lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mfma_pp<PERSIST, FAST, CB, STAMP>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}
template <bool PERSIST, bool STAMP = false>
hipError_t launch_pp_modes(const QMfmaArgs& a, unsigned grid, int lds, hipStream_t st)
{
    const QStep& q = a.to_c;
    const bool fast = !q.identity && q.O == QG_SAT_TCPL && q.Q == QG_TRN_TCPL && q.d >= 0;
    switch (a.cbytes) {
    case 1: return fast ? launch_pp<PERSIST, true, 1, STAMP>(a, grid, lds, st) : launch_pp<PERSIST, false, 1, STAMP>(a, grid, lds, st);
    case 2: return fast ? launch_pp<PERSIST, true, 2, STAMP>(a, grid, lds, st) : launch_pp<PERSIST, false, 2, STAMP>(a, grid, lds, st);
    case 4: return fast ? launch_pp<PERSIST, true, 4, STAMP>(a, grid, lds, st) : launch_pp<PERSIST, false, 4, STAMP>(a, grid, lds, st);
    case 8: return launch_pp<PERSIST, false, 8, STAMP>(a, grid, lds, st);   // (C beyond 31 value bits: raw dot products for the 64-bit pass)
    default: return hipErrorInvalidValue;
    }
}
} // namespace

hipError_t qg_launch_mfma_pp(const QMfmaArgs& a, hipStream_t st)
{
    if (a.has_ep || a.kara) return hipErrorInvalidValue;
    const int64_t blocks = (a.Mp / TM) * (a.Np / TN);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll || a.Kp % BK || a.Mp % TM || a.Np % TN) return hipErrorInvalidValue;
    constexpr int lds = 2 * BUF;
    const QMfmaArgs& b = a;
#ifdef QG_DIAG
    if (QG_DIAG_ENV("QG_PP_LAUNCH_PER_TILE")) return launch_pp_modes<false>(b, (unsigned)blocks, lds, st);   // A/B: one workgroup per tile
#endif
    // one workgroup per CU (128 KB of LDS each), a multiple of 8 so that every XCD residue class has the same number
    int dev = 0, cus = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
    int64_t grid = cus / 8 * 8;
    if (grid < 8) grid = 8;
    if (grid > blocks) grid = (blocks + 7) / 8 * 8;   // (fewer tiles than CUs: surplus workgroups find their list empty)
#ifdef QG_DIAG
    if (a.dbg) return launch_pp_modes<true, true>(b, (unsigned)grid, lds + 4096, st);
    if (const char* dm = getenv("QG_PP_DMA")) {   // A/B: DMA pieces per phase (fast 1-byte variant only)
        const QStep& q = a.to_c;
        if (a.cbytes == 1 && !q.identity && q.O == QG_SAT_TCPL && q.Q == QG_TRN_TCPL && q.d >= 0) {
            static std::atomic<uint64_t> d0{0}, d1{0}, d2{0};
            switch (atoi(dm)) {
            case 0:
                if (hipError_t er = qg_lds_attr((const void*)k_mfma_pp<true, true, 1, false, true, 0>, lds, d0); er != hipSuccess) return er;
                hipLaunchKernelGGL((k_mfma_pp<true, true, 1, false, true, 0>), dim3((unsigned)grid), dim3(512), lds, st, b);
                return hipGetLastError();
            case 1:
                if (hipError_t er = qg_lds_attr((const void*)k_mfma_pp<true, true, 1, false, true, 1>, lds, d1); er != hipSuccess) return er;
                hipLaunchKernelGGL((k_mfma_pp<true, true, 1, false, true, 1>), dim3((unsigned)grid), dim3(512), lds, st, b);
                return hipGetLastError();
            case 2:
                if (hipError_t er = qg_lds_attr((const void*)k_mfma_pp<true, true, 1, false, true, 2>, lds, d2); er != hipSuccess) return er;
                hipLaunchKernelGGL((k_mfma_pp<true, true, 1, false, true, 2>), dim3((unsigned)grid), dim3(512), lds, st, b);
                return hipGetLastError();
            default: break;
            }
        }
    }
    if (QG_DIAG_ENV("QG_PP_PH2") || QG_DIAG_ENV("QG_PP_PH4")) {   // A/B of the phase structure (fast 1-byte variant only)
        const QStep& q = a.to_c;
        if (a.cbytes == 1 && !q.identity && q.O == QG_SAT_TCPL && q.Q == QG_TRN_TCPL && q.d >= 0) {
            static std::atomic<uint64_t> d2{0}, d4{0};
            if (QG_DIAG_ENV("QG_PP_PH2")) {
                if (hipError_t er = qg_lds_attr((const void*)k_mfma_pp<true, true, 1, false, true>, lds, d2); er != hipSuccess) return er;
                hipLaunchKernelGGL((k_mfma_pp<true, true, 1, false, true>), dim3((unsigned)grid), dim3(512), lds, st, b);
            } else {
                if (hipError_t er = qg_lds_attr((const void*)k_mfma_pp<true, true, 1, false, false>, lds, d4); er != hipSuccess) return er;
                hipLaunchKernelGGL((k_mfma_pp<true, true, 1, false, false>), dim3((unsigned)grid), dim3(512), lds, st, b);
            }
            return hipGetLastError();
        }
    }
#endif
    return launch_pp_modes<true>(b, (unsigned)grid, lds, st);
}
