// qg_mfma.hip — the linear class ("class L") on the gfx950 matrix cores.
//
// When every intermediate conversion of the reference expression is provably the identity, the
// whole tree equals the exact integer dot product and ONE round + overflow into C's format is
// the entire epilogue (converting constructor, /root/reference/include/QuBLAS.h:2398-2411).  The dot product runs on
// v_mfma_i32_32x32x32_i8 (limb kernels) or v_mfma_i32_16x16x64_i8 (single limb).  Operands wider than 8 storage bits are split on the host-facing pack
// step into balanced base-256 int8 limbs (x = sum_l d_l * 256^l, d_l in [-128,127]); the kernel
// then issues LA*LB MFMAs per k-step into LA+LB-1 int32 accumulators, one per limb weight, and
// recombines them in 64-bit in the epilogue.  int32 accumulation is exact: |d*d'| <= 2^14 and
// the planner bounds K * min(LA,LB) <= 2^17.
//
// Tiling (wave64, CDNA4): 128x128 or 256x256 output tile per workgroup of 4 or 8 waves, each wave a
// TI x TJ grid of 32x32 MFMA tiles (template parameters; qg_mfma_pick chooses per shape).  A and B k-tiles are copied HBM -> LDS with
// global_load_lds_dwordx4 (16 B per lane, no VGPR round trip) into a double-buffered LDS image
// whose 16-byte chunks are XOR-swizzled on the SOURCE address (the LDS-DMA destination is
// lane-linear) so that the ds_read_b128 fragment reads are bank-conflict free.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>
#include <utility>

#include "qg_eltwise.h"
#include "qg_kernels.h"
#include "qg_step_all.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define QG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define QG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS image of one k-tile: rows of BK bytes; chunk c (16 B) of row r is stored at slot c ^ swz(r).
// swz(r) = (r / rows_per_bank_row) % chunks_per_row makes the 16 lanes of every ds_read_b128 group
// of a 32x32x32 fragment read (rows l&31, one chunk column) land on 16 distinct 16-byte bank slots.
template <int BK>
__device__ __forceinline__ int swz(int r)
{
    constexpr int CPR = BK / 16;   // chunks per row
    constexpr int RPB = 256 / BK;  // rows per 256-byte bank row
    const int q = (r / RPB) % CPR;
    // For 64-byte rows q -> {0,2,3,1}[q] keeps the 32x32x32 reads conflict free (any bijection does) AND makes the
    // 16x16x64 fragment reads (rows l&15, chunk l>>4) conflict free; k_pack applies the same function.
    return BK == 64 ? ((0x78 >> (2 * q)) & 3) : q;
}

// Issue-order hint for one basic block that holds NM MFMAs, NDS LDS reads and NVM LDS-DMA issues: the reads and DMA issues
// are spread evenly between the MFMAs (sched_group_barrier masks: 0x008 MFMA, 0x100 DS read, 0x020 VMEM read) instead of the
// compiler's default of a burst of reads / a burst of DMA issues and then the MFMAs back to back.  Both waves of a SIMD leave
// the k-tile barrier together, so bursts collide and the matrix pipe idles while both issue ~100-cycle DMA instructions;
// interleaved, one wave's MFMAs cover the other's issue slots.  Measured on the 3x3 kernel at 4096^3: 0.432 vs 0.4525 ms.
template <int NM, int NDS, int NVM, int... M>
__device__ __forceinline__ void interleave_hint(std::integer_sequence<int, M...>)
{
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0),
      __builtin_amdgcn_sched_group_barrier(0x100, (M + 1) * NDS / NM - M * NDS / NM, 0),
      __builtin_amdgcn_sched_group_barrier(0x020, (M + 1) * NVM / NM - M * NVM / NM, 0)),
     ...);
}

// LA, LB : int8 limbs per A / B element          BK       : k-tile in bytes
// WGM x WGN waves per workgroup                  TI x TJ  : 32x32 MFMA tiles per wave
// SA, SB: limb planes STORED per operand (>= LA, LB).  Multi-limb operands carry a plane mask in their trailer
// (QPackedGeom::trailer): a plane that is zero in the whole operand contributes nothing, so 17-bit data that stays inside
// [-32640, 32639] never touches the third int8 limb and its product needs 4 limb products instead of 9 — exactly.  For 3 x 3
// stored planes the launcher starts TWO kernels back to back, <3,3> and <2,2 on 3,3-plane storage>; each reads the masks and
// returns in its first instructions unless it is the one that fits, so the choice is made on the device from data that
// travels with the packed operand (no host read-back, no stale host state).  (One kernel with a run-time-masked MFMA
// sequence was measured: it spills and runs 2.4x slower than the full sequence.)
// KARA (LA = LB = 2, operands of at most 12 value+sign bits): the planes hold UNSIGNED base-64 digits d0, d1 of x + bias, and
// the kernel issues THREE products per k-step instead of four (Karatsuba): P0 = A0.B0, P1 = A1.B1, P01 = (A0+A1).(B0+B1) —
// digit sums <= 126 still fit the MFMA's signed int8 operands, and with non-negative digits the sum of two packed-byte
// registers is one v_add_u32 (no carry crosses a byte).  sum a'b' = P0 + 64 (P01 - P0 - P1) + 4096 P1, and the bias goes
// out with the operands' row sums: sum ab = sum a'b' - biasB rsA[i] - biasA rsB[j] + K biasA biasB.  Exact: every P is an
// int32 as long as K * 126^2 < 2^31 (planner).  (The 3 x 3-digit form needs 6 accumulator sets — 384 registers for a
// 64 x 64 wave tile — which hipcc cannot allocate without spilling)
// KS > 1 (small problems): KS wave groups of WGM x WGN waves split the k-tiles of ONE output tile among them — each group has its
// own LDS ring and walks nk / KS consecutive k-tiles, twice the LDS-DMA transfers in flight and half the k-loop per wave —
// and the groups' accumulators are summed through LDS before the one epilogue (no atomics, no memset, no second launch).
template <int LA, int LB, int BK, int WGM, int WGN, int TI, int TJ, int NSTAGE, int ABL, bool EP, int SA = LA, int SB = LB, bool KARA = false, int KS = 1>   // EP: fused element-wise epilogue
__global__ __launch_bounds__(64 * WGM * WGN * KS) void k_mfma(QMfmaArgs g)
{
    if constexpr (SA == 3 && SB == 3 && ABL == 0) {
        const unsigned ma = qg_plane_mask(g.maskA);
        const unsigned mb = qg_plane_mask(g.maskB);
        const bool two_planes_suffice = ((ma | mb) & 4u) == 0;
        if (two_planes_suffice != (LA == 2)) return;   // the other kernel of this launch pair does the work
    }
    constexpr int TM = WGM * TI * 32, TN = WGN * TJ * 32;
    constexpr int NWAVES = WGM * WGN;
    constexpr int NW = LA + LB - 1;            // limb weights
    constexpr int ROWS = LA * TM + LB * TN;    // LDS rows per stage
    constexpr int STAGE = ROWS * BK;           // bytes per stage
    constexpr int PIECES = STAGE / 1024;       // 1-KiB LDS-DMA pieces per stage
    constexpr int PPW = PIECES / NWAVES;       // pieces (LDS-DMA instructions) per wave per stage
    constexpr int KSTEPS = BK / 32;            // MFMA k-steps per tile
    static_assert(PIECES % NWAVES == 0, "every wave issues the same number of LDS-DMA pieces");
    static_assert(NSTAGE >= 2 && (NSTAGE - 2) * PPW < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = KS > 1 ? wave_all / NWAVES : 0;          // k group of this wave (wave-uniform)
    const int wave = KS > 1 ? wave_all % NWAVES : wave_all;
    char* smem = smem_all + kg * (NSTAGE * STAGE);           // the group's own ring
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: consecutive block ids go to different XCDs, so give each XCD a
    // contiguous run of tiles, walked in column-major groups of 8 tile-rows for L2 reuse.
    const int tiles_m = (int)(g.Mp / TM), tiles_n = (int)(g.Np / TN);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    constexpr int GM = 8;
    const int grp = bid / (GM * tiles_n);
    const int first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + (bid % (GM * tiles_n)) % gsz;
    const int tile_n = (bid % (GM * tiles_n)) / gsz;

    // operands are pre-tiled: block (row tile, k tile) of A is LA*TM*BK contiguous bytes that are
    // already the swizzled LDS image; same for B.  A stage is two linear copies.
    const int nk_all = (int)(g.Kp / BK);
    const int nk = nk_all / KS;                                         // k-tiles of this group (the launcher checks divisibility)
    constexpr int A_BYTES = LA * TM * BK, B_BYTES = LB * TN * BK;       // copied per stage (the first LA / LB planes)
    constexpr int A_STORED = SA * TM * BK, B_STORED = SB * TN * BK;     // stride of a (row tile, k tile) block in memory
    constexpr int A_PIECES = A_BYTES / 1024;
    const int8_t* Ag = g.A + ((int64_t)tile_m * nk_all + (int64_t)kg * nk) * A_STORED + lane * 16;
    const int8_t* Bg = g.B + ((int64_t)tile_n * nk_all + (int64_t)kg * nk) * B_STORED + lane * 16;

    auto issue = [&](int stage, int kt) {
        char* sbase = smem + stage * STAGE;
        const int kts = ABL == 6 ? 0 : kt;  // diagnostic: re-read k-tile 0 (cache hits) to separate DMA cost from traffic cost
        const int8_t* a = Ag + (int64_t)kts * A_STORED;
        const int8_t* b = Bg + (int64_t)kts * B_STORED;
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi) {
            const int p = wave + NWAVES * pi;       // wave-uniform piece id
            const int8_t* src = p < A_PIECES ? a + p * 1024 : b + (p - A_PIECES) * 1024;
            __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(src), QG_LDS_PTR(sbase + p * 1024), 16, 0, 0);
        }
    };

    v16i acc[NW][TI][TJ];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[w][i][j][e] = 0;

    const int fr = lane & 31, fh = lane >> 5;
    // Software pipeline (needs NSTAGE == 3 stages, KSTEPS even):
    //   * fragments are double-buffered in registers: while the MFMAs of k-step s run, the
    //     ds_read_b128s of k-step s+1 are already in flight, so LDS latency is never exposed;
    //   * ONE barrier per k-tile, placed before the LAST k-step of tile kt: it publishes tile kt+1
    //     (every wave has waited for its own LDS-DMA pieces of that tile), after which the first
    //     fragments of tile kt+1 are prefetched and the stage last read in iteration kt-1 is
    //     refilled with tile kt+2 (raw s_barrier: __syncthreads() would also drain the DMA queue).
    static_assert((NSTAGE == 3 || (NSTAGE > 3 && ABL == 0)) && KSTEPS % 2 == 0, "pipeline shape");
    v4i fa[2][LA][TI], fb[2][LB][TJ];
    auto load_frags = [&](int set, const char* stage_base, int ks) {
        const char* sA = stage_base;
        const char* sB = stage_base + LA * TM * BK;
        const int c = 2 * ks + fh;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ra = (wm * TI + i) * 32 + fr;
#pragma unroll
            for (int l = 0; l < LA; ++l) fa[set][l][i] = *(const v4i*)(sA + (l * TM + ra) * BK + ((c ^ swz<BK>(ra)) * 16));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int rb = (wn * TJ + j) * 32 + fr;
#pragma unroll
            for (int l = 0; l < LB; ++l) fb[set][l][j] = *(const v4i*)(sB + (l * TN + rb) * BK + ((c ^ swz<BK>(rb)) * 16));
        }
    };
    // the LA*LB*TI*TJ MFMAs of one k-step, optionally only those with index in [first, last)
    auto mfmas = [&](int set, int first, int last) {
        if constexpr (KARA) {
            static_assert(!KARA || (LA == 2 && LB == 2), "Karatsuba variant: 2 x 2 digits");
            v4i sa[TI], sb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) sa[i] = fa[set][0][i] + fa[set][1][i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) sb[j] = fb[set][0][j] + fb[set][1][j];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][0][i], fb[set][0][j], acc[0][i][j], 0, 0, 0);   // P0
                    acc[2][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][1][i], fb[set][1][j], acc[2][i][j], 0, 0, 0);   // P1
                    acc[1][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(sa[i], sb[j], acc[1][i][j], 0, 0, 0);                   // P01
                }
            return;
        }
        int n = 0;
#pragma unroll
        for (int la = 0; la < LA; ++la)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j, ++n)
                        if (n >= first && n < last)
                            acc[la + lb][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][la][i], fb[set][lb][j], acc[la + lb][i][j], 0, 0, 0);
    };
    constexpr int NM = (KARA ? 3 : LA * LB) * TI * TJ;  // MFMAs per k-step per wave

    // prologue: tiles 0 and 1 in flight, tile 0 published, its first fragments loaded
    issue(0, 0);
    if constexpr (NSTAGE == 3) {
        if (nk > 1) issue(1, 1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // NSTAGE - 1 tiles in flight (small problems: a k-tile of a 64x64-tile kernel is ~100 issue cycles, a fraction of one
        // trip to L2, so with two tiles in flight every k-tile waits for its data); past the end the last tile is fetched again
#pragma unroll
        for (int t = 1; t < NSTAGE - 1; ++t) issue(t, t < nk ? t : nk - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PPW) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, smem, 0);
    int cur = 0;  // stage of tile kt
    for (int kt = 0; kt < nk; ++kt) {
        const char* sc = smem + cur * STAGE;
        const int nx = cur + 1 == NSTAGE ? 0 : cur + 1;   // stage of tile kt+1
        const int rf = cur == 0 ? NSTAGE - 1 : cur - 1;   // stage of tile kt-1: refilled with tile kt+NSTAGE-1
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + 1 < KSTEPS) {
                if (ABL != 2 && ABL != 3 && ABL != 4) load_frags((ks + 1) & 1, sc, ks + 1);
                if constexpr (ABL == 0) {
                    mfmas(ks & 1, 0, NM);
                    interleave_hint<NM, LA * TI + LB * TJ, 0>(std::make_integer_sequence<int, NM>{});
                    continue;
                }
            } else {
                // tile kt+1 was issued one iteration ago: wait for this wave's pieces, then publish
                if (ABL != 4) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 3) * PPW) : "memory");   // tile kt+1 is in; kt+2 ... may fly on
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // (staggering the DMA issue between the two waves of a SIMD was measured: no gain for
                // the limb kernel, 2-4 % slower for the single-limb one — all waves issue here)
                if constexpr (ABL == 0) {
                    // branch-free refill (past the end: the last tile once more, into a stage nobody reads again) so that the
                    // LDS-DMA issues, the fragment reads of the next tile and this k-step's MFMAs share one basic block
                    issue(rf, kt + NSTAGE - 1 < nk ? kt + NSTAGE - 1 : nk - 1);
                    load_frags(0, smem + nx * STAGE, 0);
                    mfmas(ks & 1, 0, NM);
                    interleave_hint<NM, LA * TI + LB * TJ, PPW>(std::make_integer_sequence<int, NM>{});
                    continue;
                }
                if (kt + 2 < nk && ABL != 1 && ABL != 3 && ABL != 4) issue(rf, kt + 2);
                if (kt + 1 < nk && ABL != 2 && ABL != 3 && ABL != 4) load_frags(0, smem + nx * STAGE, 0);
            }
            mfmas(ks & 1, 0, NM);
        }
        cur = nx;
    }
    // the branch-free refill leaves LDS-DMA transfers of the clamped tile in flight: they must have landed before this
    // workgroup can end and its LDS be handed to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if constexpr (KS > 1) {
        // sum the groups' accumulators: groups 1 .. KS-1 park theirs in LDS (the rings are free now), group 0 adds them
        static_assert(KS == 1 || (NW == 1 && !KARA && !EP), "k split: single-limb kernels");
        __syncthreads();
        int* red = (int*)smem_all;   // [KS - 1][NWAVES][TI * TJ][16][64]
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((((kg - 1) * NWAVES + wave) * (TI * TJ) + i * TJ + j) * 16 + e) * 64 + lane] = acc[0][i][j][e];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][i][j][e] += red[(((q * NWAVES + wave) * (TI * TJ) + i * TJ + j) * 16 + e) * 64 + lane];
    }

    // epilogue: recombine limb weights, ONE round + overflow into C, store.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5),
    // i.e. every lane owns runs of 4 consecutive rows of one column.  Packed C is tiled
    // [tile_m][tile_n][col][row] (column-major inside the tile, the order of the host tensor), so a
    // run of 4 rows is one 4/8/16/32-byte store per lane.
    const QStep st = g.to_c;
    char* C = (char*)g.C;
    const int64_t tile_base = ((int64_t)tile_m * tiles_n + tile_n) * TM * TN;
    using S = std::conditional_t<(NW == 1), int32_t, int64_t>;  // one limb pair: the int32 accumulator is the dot product
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            S s[16];
            if constexpr (KARA) {
                const int64_t cj = g.corr - g.biasA * g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 32 + fr];
                const int64_t* ra = g.rsA + (int64_t)tile_m * TM + (wm * TI + i) * 32 + 4 * fh;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t p0 = acc[0][i][j][e], p01 = acc[1][i][j][e], p1 = acc[2][i][j][e];
                    s[e] = (S)(p0 + 64 * (p01 - p0 - p1) + 4096 * p1 - g.biasB * ra[(e & 3) + 8 * (e >> 2)] + cj);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S x = (S)acc[NW - 1][i][j][e];
#pragma unroll
                    for (int w = NW - 2; w >= 0; --w) x = x * 256 + (S)acc[w][i][j][e];
                    s[e] = x;
                }
                if constexpr (NW > 1) {
                    if (g.rsA) {   // centred operands (QPackedGeom::offs): sum a b = sum a'b' - biasB rsA[row] - biasA rsB[col] + K biasA biasB (wrapping)
                        const uint64_t cj = (uint64_t)g.corr - (uint64_t)g.biasA * (uint64_t)g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 32 + fr];
                        const int64_t* ra = g.rsA + (int64_t)tile_m * TM + (wm * TI + i) * 32 + 4 * fh;
#pragma unroll
                        for (int e = 0; e < 16; ++e) s[e] = (S)((uint64_t)s[e] + cj - (uint64_t)g.biasB * (uint64_t)ra[(e & 3) + 8 * (e >> 2)]);
                    }
                }
            }
            bool converted = false;
            if constexpr (NW == 1 && !KARA) {
                if (g.rsA) {   // a centred single-limb pair: the sum needs 64 bits, its image in C (at most 31 bits: qg_api.hip) does not
                    const uint64_t cj = (uint64_t)g.corr - (uint64_t)g.biasA * (uint64_t)g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 32 + fr];
                    const int64_t* ra = g.rsA + (int64_t)tile_m * TM + (wm * TI + i) * 32 + 4 * fh;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        s[e] = (S)qg_step<int64_t>((int64_t)((uint64_t)(int64_t)s[e] + cj - (uint64_t)g.biasB * (uint64_t)ra[(e & 3) + 8 * (e >> 2)]), st);
                    converted = true;
                }
            }
            if (!converted) qg_step_all<S, 16>(s, st);
            if (ABL == 5 && s[0] != (S)0x7ead1234) continue; // diagnostic: keep the arithmetic, drop the stores
            const int col = (wn * TJ + j) * 32 + fr;
            const int row0 = (wm * TI + i) * 32 + 4 * fh;
            const int64_t base = tile_base + (int64_t)col * TM + row0;
            if constexpr (EP) {
                // the value just converted into C's element type goes through the element-wise chain and is stored
                // as D: C itself never reaches memory (qg_eltwise.h); 32-bit arithmetic when the planner allows it
                // (fused only for chains the planner has bounded by 32 bits: qg_api.hip, fuses_epilogue)
                int32_t v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = (int32_t)s[e];
                qg_ep_apply_runs<int32_t, 4>(v, g.ep, g.epa, base, 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) qg_ep_store_run<int32_t>(C, base + 8 * q, g.ep.dbytes, v + 4 * q);
            } else
            switch (g.cbytes) {
            case 1:
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(uint32_t*)(C + base + 8 * q) = (uint32_t)(s[4 * q] & 0xff) | ((uint32_t)(s[4 * q + 1] & 0xff) << 8) |
                                                     ((uint32_t)(s[4 * q + 2] & 0xff) << 16) | ((uint32_t)(s[4 * q + 3] & 0xff) << 24);
                break;
            case 2:
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(uint2*)(C + (base + 8 * q) * 2) = make_uint2((uint32_t)(s[4 * q] & 0xffff) | ((uint32_t)(s[4 * q + 1] & 0xffff) << 16),
                                                                 (uint32_t)(s[4 * q + 2] & 0xffff) | ((uint32_t)(s[4 * q + 3] & 0xffff) << 16));
                break;
            case 4:
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(int4*)(C + (base + 8 * q) * 4) = make_int4((int)s[4 * q], (int)s[4 * q + 1], (int)s[4 * q + 2], (int)s[4 * q + 3]);
                break;
            default:
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int64_t* p = (int64_t*)(C + (base + 8 * q) * 8);
                    *(longlong2*)p = make_longlong2((int64_t)s[4 * q], (int64_t)s[4 * q + 1]);
                    *(longlong2*)(p + 2) = make_longlong2((int64_t)s[4 * q + 2], (int64_t)s[4 * q + 3]);
                }
                break;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Single-limb variant on v_mfma_i32_16x16x64_i8: same tiles, LDS image and pipeline idea, but one MFMA
// k-step covers a whole 64-byte k-tile, so the fragment double-buffer alternates per K-TILE and the one
// barrier per k-tile sits at the top of the iteration.  (A/B of the two MFMA shapes: DESIGN.md §9.)
// Fragment map: lane l holds row (l & 15), bytes [16*(l>>4), +16) of the 64-byte k-step for A and B alike;
// C/D: col = lane & 15, rows 4*(lane>>4) .. +3 in the 4 result registers.

typedef int v4acc __attribute__((ext_vector_type(4)));

// LA x LB limbs; DBUF: fragments double-buffered across k-tiles (only when the registers allow it)
template <int LA, int LB, int WGM, int WGN, int TI, int TJ, bool DBUF, bool EP, bool HINT = true, int PV = 1, int SA = LA, int SB = LB, bool KARA = false>   // TI x TJ tiles of 16x16 per wave; SA, SB, KARA: see k_mfma
__global__ __launch_bounds__(64 * WGM * WGN) void k_mfma16(QMfmaArgs g)
{
    if constexpr (SA == 3 && SB == 3) {   // plane masks: see k_mfma; a 3 x 3 launch is the pair <3,3> + <2,2 on 3-plane storage>
        const unsigned ma = qg_plane_mask(g.maskA);
        const unsigned mb = qg_plane_mask(g.maskB);
        const bool two_planes_suffice = ((ma | mb) & 4u) == 0;
        if (two_planes_suffice != (LA == 2)) return;   // the other kernel of this launch pair does the work
    }
    constexpr int BK = 64, NSTAGE = 3;
    constexpr int TM = WGM * TI * 16, TN = WGN * TJ * 16;
    constexpr int NWAVES = WGM * WGN;
    constexpr int NW = LA + LB - 1;
    constexpr int STAGE = (LA * TM + LB * TN) * BK;
    constexpr int PIECES = STAGE / 1024;
    constexpr int PPW = PIECES / NWAVES;
    constexpr int NSET = DBUF ? 2 : 1;
    static_assert(PIECES % NWAVES == 0, "every wave issues the same number of LDS-DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_m = (int)(g.Mp / TM), tiles_n = (int)(g.Np / TN);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    constexpr int GM = 8;
    const int grp = bid / (GM * tiles_n);
    const int first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + (bid % (GM * tiles_n)) % gsz;
    const int tile_n = (bid % (GM * tiles_n)) / gsz;

    const int nk = (int)(g.Kp / BK);
    constexpr int A_BYTES = LA * TM * BK, B_BYTES = LB * TN * BK, A_PIECES = A_BYTES / 1024;   // copied per stage (the first LA / LB planes)
    constexpr int A_STRIDE = SA * TM * BK, B_STRIDE = SB * TN * BK;                            // stored per k-tile
    const int8_t* Ag = g.A + (int64_t)tile_m * nk * A_STRIDE + lane * 16;
    const int8_t* Bg = g.B + (int64_t)tile_n * nk * B_STRIDE + lane * 16;
    auto issue = [&](int stage, int kt) {
        char* sbase = smem + stage * STAGE;
        const int8_t* a = Ag + (int64_t)kt * A_STRIDE;
        const int8_t* b = Bg + (int64_t)kt * B_STRIDE;
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi) {
            const int p = wave + NWAVES * pi;
            const int8_t* src = p < A_PIECES ? a + p * 1024 : b + (p - A_PIECES) * 1024;
            __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(src), QG_LDS_PTR(sbase + p * 1024), 16, 0, 0);
        }
    };

    auto issue_part = [&](int stage, int kt, int p0, int p1) {   // pieces [p0, p1) of this wave's share
        char* sbase = smem + stage * STAGE;
        const int8_t* a = Ag + (int64_t)kt * A_STRIDE;
        const int8_t* b = Bg + (int64_t)kt * B_STRIDE;
#pragma unroll
        for (int pi = p0; pi < p1; ++pi) {
            const int p = wave + NWAVES * pi;
            const int8_t* src = p < A_PIECES ? a + p * 1024 : b + (p - A_PIECES) * 1024;
            __builtin_amdgcn_global_load_lds(QG_GLOBAL_PTR(src), QG_LDS_PTR(sbase + p * 1024), 16, 0, 0);
        }
    };

    v4acc acc[NW][TI][TJ];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[w][i][j][e] = 0;

    const int fr = lane & 15, fq = lane >> 4;
    v4i fa[NSET][LA][TI], fb[NSET][LB][TJ];
    auto load_frags = [&](int set, const char* sA) {
        const char* sB = sA + LA * TM * BK;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ra = (wm * TI + i) * 16 + fr;
#pragma unroll
            for (int l = 0; l < LA; ++l) fa[set][l][i] = *(const v4i*)(sA + (l * TM + ra) * BK + ((fq ^ swz<BK>(ra)) * 16));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int rb = (wn * TJ + j) * 16 + fr;
#pragma unroll
            for (int l = 0; l < LB; ++l) fb[set][l][j] = *(const v4i*)(sB + (l * TN + rb) * BK + ((fq ^ swz<BK>(rb)) * 16));
        }
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int la = 0; la < LA; ++la)
#pragma unroll
            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[la + lb][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][la][i], fb[set][lb][j], acc[la + lb][i][j], 0, 0, 0);
    };
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    constexpr bool DEEP = DBUF && HINT && PV != 0;   // two k-tiles in flight (below)
    // tiles 0, 1 (and 2) in flight; tile 0 published
    issue(0, 0);
    if constexpr (DEEP) {
        issue(1, nk > 1 ? 1 : nk - 1);
        issue(2, nk > 2 ? 2 : nk - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    } else {
        if (nk > 1) issue(1, 1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!DBUF) {
        // Limb variant: the fragment double-buffer of a whole k-tile does not fit beside NW accumulator sets, so the k-tile is
        // walked in TI row steps: A fragments of row step i+1 (LA reads) are fetched under the MFMAs of row step i into the
        // other of two small buffers, B fragments of the next k-tile under the second-to-last row step.
        static_assert(TI % 2 == 0, "the A buffer parity must return to 0 at the end of a k-tile");
        v4i pa[2][LA], pb[2][LB][TJ];
        auto load_a = [&](int buf, const char* sA, int i) {
            const int ra = (wm * TI + i) * 16 + fr;
#pragma unroll
            for (int l = 0; l < LA; ++l) pa[buf][l] = *(const v4i*)(sA + (l * TM + ra) * BK + ((fq ^ swz<BK>(ra)) * 16));
        };
        auto load_b = [&](int buf, const char* sA) {
            const char* sB = sA + LA * TM * BK;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int rb = (wn * TJ + j) * 16 + fr;
#pragma unroll
                for (int l = 0; l < LB; ++l) pb[buf][l][j] = *(const v4i*)(sB + (l * TN + rb) * BK + ((fq ^ swz<BK>(rb)) * 16));
            }
        };
        load_b(0, smem);
        load_a(0, smem, 0);
        int st0 = 0, st1 = 1, st2 = 2;   // stages of tiles k, k+1, k+2
        for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kt + h;
                if (k < nk) {
                    publish();
                    const int kn = k + 2 < nk ? k + 2 : nk - 1;
                    const char* cur = smem + st0 * STAGE;
                    const char* nxt = smem + st1 * STAGE;
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        // reads first, fenced: their first use is a whole row step (LA*LB*TJ MFMAs) away
                        if (i + 1 < TI) load_a((i + 1) & 1, cur, i + 1);
                        else load_a(0, nxt, 0);
                        if (i == TI - 2) load_b(h ^ 1, nxt);
                        if constexpr (HINT) __builtin_amdgcn_sched_barrier(0);
                        // LDS-DMA of tile k+2 in the first two row steps: a piece issued late in the iteration has no time to
                        // land before the next publish (spread over all four row steps, PV = 0: 0.414 vs 0.390 ms at 4096^3)
                        if constexpr (PV == 0) issue_part(st2, kn, PPW * i / TI, PPW * (i + 1) / TI);
                        else { if (i < 2) issue_part(st2, kn, PPW * i / 2, PPW * (i + 1) / 2); }
                        if constexpr (KARA) {   // three products of unsigned base-64 digits (k_mfma): P0, P1, P01 = (a0+a1)(b0+b1)
                            static_assert(!KARA || (LA == 2 && LB == 2), "Karatsuba variant: 2 x 2 digits");
                            const v4i sa = pa[i & 1][0] + pa[i & 1][1];
#pragma unroll
                            for (int j = 0; j < TJ; ++j) {
                                const v4i sb = pb[h][0][j] + pb[h][1][j];
                                acc[0][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pa[i & 1][0], pb[h][0][j], acc[0][i][j], 0, 0, 0);
                                acc[2][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pa[i & 1][1], pb[h][1][j], acc[2][i][j], 0, 0, 0);
                                acc[1][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(sa, sb, acc[1][i][j], 0, 0, 0);
                            }
                        } else
#pragma unroll
                        for (int la = 0; la < LA; ++la)
#pragma unroll
                            for (int lb = 0; lb < LB; ++lb)
#pragma unroll
                                for (int j = 0; j < TJ; ++j)
                                    acc[la + lb][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pa[i & 1][la], pb[h][lb][j], acc[la + lb][i][j], 0, 0, 0);
                        if constexpr (HINT) {
                            interleave_hint<(KARA ? 3 : LA * LB) * TJ, 0, PV == 0 ? (PPW + TI - 1) / TI : (PPW + 1) / 2>(std::make_integer_sequence<int, (KARA ? 3 : LA * LB) * TJ>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    st0 = st1;
                    st1 = st2;
                    st2 = (st2 + 1) % 3;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // iteration k: publish tile k+1, refill the stage of tile k-1 with tile k+2, prefetch fragments of k+1, MFMA k
        load_frags(0, smem);
        int st1 = 1, st2 = 2;  // stages of tiles k+1, k+2
        for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kt + h;
                if (k < nk) {
                    if constexpr (DEEP) {
                        // The fragments of tile k were read out of LDS during iteration k-1, so the stage of tile k is free as
                        // soon as every wave is here: refill it with tile k+3 now.  Two tiles (k+2, k+3) stay in flight, each
                        // with two iterations to land, on the same three stages; the wait is counted (tile k+1 must be in, the
                        // PPW pieces of tile k+2 may fly on).  A k-tile of this kernel is 32 MFMAs = 512 issue cycles, less than
                        // one trip to L2.
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        issue((st1 + 2) % 3, k + 3 < nk ? k + 3 : nk - 1);
                        load_frags(h ^ 1, smem + st1 * STAGE);
                        mfmas(h);
                        interleave_hint<LA * LB * TI * TJ, LA * TI + LB * TJ, PPW>(std::make_integer_sequence<int, LA * LB * TI * TJ>{});
                        st1 = st2;
                        st2 = (st1 + 1) % 3;
                        continue;
                    }
                    // branch-free body (past the end: the last tile is fetched once more into a stage nobody reads again, and
                    // fragments are read from it in vain) so that DMA issues, fragment reads and MFMAs share one basic block
                    // and interleave_hint can spread them
                    if constexpr (HINT) {
                        publish();
                        issue(st2, k + 2 < nk ? k + 2 : nk - 1);
                        load_frags(DBUF ? (h ^ 1) : 0, smem + st1 * STAGE);
                        mfmas(DBUF ? h : 0);
                        interleave_hint<LA * LB * TI * TJ, LA * TI + LB * TJ, PPW>(std::make_integer_sequence<int, LA * LB * TI * TJ>{});
                    } else {   // the compiler's own issue order (A/B: QG_NO_HINT16)
                        if (k + 1 < nk) {
                            publish();
                            if (k + 2 < nk) issue(st2, k + 2);
                            load_frags(DBUF ? (h ^ 1) : 0, smem + st1 * STAGE);
                        }
                        mfmas(DBUF ? h : 0);
                    }
                    st1 = st2;
                    st2 = (st1 + 1) % 3;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // in-flight LDS-DMA of the clamped refill must land before the workgroup ends
    }

    // epilogue: recombine limb weights, one round + overflow into C, store runs of 4 rows (column-major tile)
    const QStep st = g.to_c;
    char* C = (char*)g.C;
    const int64_t tile_base = ((int64_t)tile_m * tiles_n + tile_n) * TM * TN;
    using S = std::conditional_t<(NW == 1), int32_t, int64_t>;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        S s[4 * TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (KARA) {   // sum a'b' = P0 + 64 (P01 - P0 - P1) + 4096 P1; the biases go out with the row sums (k_mfma)
                    const int64_t cj = g.corr - g.biasA * g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 16 + fr];
                    const int64_t ra = g.rsA[(int64_t)tile_m * TM + (wm * TI + i) * 16 + 4 * fq + e];
                    const int64_t p0 = acc[0][i][j][e], p01 = acc[1][i][j][e], p1 = acc[2][i][j][e];
                    s[j * 4 + e] = (S)(p0 + 64 * (p01 - p0 - p1) + 4096 * p1 - g.biasB * ra + cj);
                    continue;
                }
                S x = (S)acc[NW - 1][i][j][e];
#pragma unroll
                for (int w = NW - 2; w >= 0; --w) x = x * 256 + (S)acc[w][i][j][e];
                if constexpr (NW > 1) {
                    if (g.rsA)   // centred operands (QPackedGeom::offs; k_mfma): the centres go out with the row sums (wrapping arithmetic)
                        x = (S)((uint64_t)x + (uint64_t)g.corr - (uint64_t)g.biasA * (uint64_t)g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 16 + fr] -
                                (uint64_t)g.biasB * (uint64_t)g.rsA[(int64_t)tile_m * TM + (wm * TI + i) * 16 + 4 * fq + e]);
                }
                s[j * 4 + e] = x;
            }
        bool converted = false;
        if constexpr (NW == 1 && !KARA) {
            if (g.rsA) {   // a centred single-limb pair (k_mfma)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[j * 4 + e] = (S)qg_step<int64_t>((int64_t)((uint64_t)(int64_t)s[j * 4 + e] + (uint64_t)g.corr -
                                                                     (uint64_t)g.biasA * (uint64_t)g.rsB[(int64_t)tile_n * TN + (wn * TJ + j) * 16 + fr] -
                                                                     (uint64_t)g.biasB * (uint64_t)g.rsA[(int64_t)tile_m * TM + (wm * TI + i) * 16 + 4 * fq + e]), st);
                converted = true;
            }
        }
        if (!converted) qg_step_all<S, 4 * TJ>(s, st);
        const int row0 = (wm * TI + i) * 16 + 4 * fq;
        if constexpr (EP) {
            const int64_t base0 = tile_base + (int64_t)(wn * TJ * 16 + fr) * TM + row0;   // run j starts 16 columns further
            int32_t v[4 * TJ];
#pragma unroll
            for (int e = 0; e < 4 * TJ; ++e) v[e] = (int32_t)s[e];
            qg_ep_apply_runs<int32_t, TJ>(v, g.ep, g.epa, base0, (int64_t)16 * TM);
#pragma unroll
            for (int j = 0; j < TJ; ++j) qg_ep_store_run<int32_t>(C, base0 + (int64_t)j * 16 * TM, g.ep.dbytes, v + 4 * j);
        } else
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = (wn * TJ + j) * 16 + fr;
            const int64_t base = tile_base + (int64_t)col * TM + row0;
            const S* q = s + j * 4;
            switch (g.cbytes) {
            case 1:
                *(uint32_t*)(C + base) = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
                break;
            case 2:
                *(uint2*)(C + base * 2) = make_uint2((uint32_t)(q[0] & 0xffff) | ((uint32_t)(q[1] & 0xffff) << 16), (uint32_t)(q[2] & 0xffff) | ((uint32_t)(q[3] & 0xffff) << 16));
                break;
            case 4:
                *(int4*)(C + base * 4) = make_int4((int)q[0], (int)q[1], (int)q[2], (int)q[3]);
                break;
            default: {
                int64_t* p = (int64_t*)(C + base * 8);
                *(longlong2*)p = make_longlong2((int64_t)q[0], (int64_t)q[1]);
                *(longlong2*)(p + 2) = make_longlong2((int64_t)q[2], (int64_t)q[3]);
                break;
            }
            }
        }
    }
}

// The partner of a 3 x 3 launch for operands whose third limb planes are empty: 2 x 2 limbs read from the 3-plane storage
// (each kernel of the pair returns at once unless the plane masks select it).
template <int BK, int WGM, int WGN, int TI, int TJ, int NSTAGE, bool EP>
void launch_plane_partner(const QMfmaArgs& a, hipStream_t st, int64_t blocks)
{
    static const bool no_partner = QG_DIAG_ENV("QG_NO_PLANE_MASK");   // A/B switch (tools/measure_masked.py); full-range data only!
    if (!(a.maskA || a.maskB) || no_partner) return;
    constexpr int TM = WGM * TI * 32, TN = WGN * TJ * 32;
    constexpr int lds2 = NSTAGE * (2 * TM + 2 * TN) * BK;
    static std::atomic<uint64_t> attr_done2{0};
    if (qg_lds_attr((const void*)k_mfma<2, 2, BK, WGM, WGN, TI, TJ, NSTAGE, 0, EP, 3, 3>, lds2, attr_done2) != hipSuccess) return;
    hipLaunchKernelGGL((k_mfma<2, 2, BK, WGM, WGN, TI, TJ, NSTAGE, 0, EP, 3, 3>), dim3((unsigned)blocks), dim3(64 * WGM * WGN), lds2, st, a);
}

template <int LA, int LB, int WGM, int WGN, int TI, int TJ, bool DBUF, bool EP = false, bool HINT = true, int PV = 1, int SA = LA, int SB = LB, bool KARA = false>
hipError_t launch16(const QMfmaArgs& a, hipStream_t st)
{
    if constexpr (HINT && !EP) {
        static const bool no_hint = QG_DIAG_ENV("QG_NO_HINT16");
        if (no_hint && !a.has_ep) return launch16<LA, LB, WGM, WGN, TI, TJ, DBUF, false, false>(a, st);
    }
    if constexpr (!EP && (LA * LB == 1 || LA * LB == 9)) {
        // (single limb: the fused variant spills; with the issue-order hint the spill code lands in the main loop, 2.9 ms instead
        // of 0.49 ms at 8192^2 x 4096 — it keeps the compiler's own order.  It is not the default placement anyway.)
        if (a.has_ep) return launch16<LA, LB, WGM, WGN, TI, TJ, DBUF, true, LA * LB != 1>(a, st);
    }
    if (a.has_ep && (!EP || !a.ep.bits32)) return hipErrorInvalidValue;
    constexpr int TM = WGM * TI * 16, TN = WGN * TJ * 16;
    const int lds = 3 * (LA * TM + LB * TN) * 64;
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (qg_lds_attr)
    if (hipError_t e = qg_lds_attr((const void*)k_mfma16<LA, LB, WGM, WGN, TI, TJ, DBUF, EP, HINT, PV, SA, SB, KARA>, lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (a.Mp / TM) * (a.Np / TN);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll || a.Kp % 64 || a.Mp % TM || a.Np % TN) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_mfma16<LA, LB, WGM, WGN, TI, TJ, DBUF, EP, HINT, PV, SA, SB, KARA>), dim3((unsigned)blocks), dim3(64 * WGM * WGN), lds, st, a);
    if constexpr (LA == 3 && LB == 3 && SA == 3) {
        // the partner for operands whose third limb planes are empty: 2 x 2 limbs read from the 3-plane storage (each kernel
        // of the pair returns at once unless the plane masks select it)
        static const bool no_partner = QG_DIAG_ENV("QG_NO_PLANE_MASK");   // A/B switch (tools/measure_masked.py); full-range data only!
        if ((a.maskA || a.maskB) && !no_partner) {
            constexpr int lds2 = 3 * (2 * TM + 2 * TN) * 64;
            static std::atomic<uint64_t> attr_done2{0};
            if (hipError_t e = qg_lds_attr((const void*)k_mfma16<2, 2, WGM, WGN, TI, TJ, DBUF, EP, HINT, PV, 3, 3>, lds2, attr_done2); e != hipSuccess) return e;
            hipLaunchKernelGGL((k_mfma16<2, 2, WGM, WGN, TI, TJ, DBUF, EP, HINT, PV, 3, 3>), dim3((unsigned)blocks), dim3(64 * WGM * WGN), lds2, st, a);
        }
    }
    return hipGetLastError();
}

template <int LA, int LB, int BK, int WGM, int WGN, int TI, int TJ, int NSTAGE, int ABL = 0, bool EP = false>
hipError_t launch(const QMfmaArgs& a, hipStream_t st)
{
    if constexpr (!EP && ABL == 0 && LA == LB && LA != 2) {   // fused variants exist for the 1x1 and 3x3 kernels only
        if (a.has_ep) return launch<LA, LB, BK, WGM, WGN, TI, TJ, NSTAGE, ABL, true>(a, st);
    }
    if (a.has_ep && (!EP || !a.ep.bits32)) return hipErrorInvalidValue;
    constexpr int TM = WGM * TI * 32, TN = WGN * TJ * 32;
    constexpr int STAGE = (LA * TM + LB * TN) * BK;
    const int lds = NSTAGE * STAGE;
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (qg_lds_attr)
    if (hipError_t e = qg_lds_attr((const void*)k_mfma<LA, LB, BK, WGM, WGN, TI, TJ, NSTAGE, ABL, EP>, lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (a.Mp / TM) * (a.Np / TN);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll || a.Kp % BK || a.Mp % TM || a.Np % TN) return hipErrorInvalidValue;
    if constexpr (LA == 2 && LB == 2 && ABL == 0 && !EP) {
        if (a.kara) {
            static std::atomic<uint64_t> attr_done_k{0};
            if (hipError_t e = qg_lds_attr((const void*)k_mfma<2, 2, BK, WGM, WGN, TI, TJ, NSTAGE, 0, false, 2, 2, true>, lds, attr_done_k); e != hipSuccess) return e;
            hipLaunchKernelGGL((k_mfma<2, 2, BK, WGM, WGN, TI, TJ, NSTAGE, 0, false, 2, 2, true>), dim3((unsigned)blocks), dim3(64 * WGM * WGN), lds, st, a);
            return hipGetLastError();
        }
    }
    if (a.kara) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_mfma<LA, LB, BK, WGM, WGN, TI, TJ, NSTAGE, ABL, EP>), dim3((unsigned)blocks), dim3(64 * WGM * WGN), lds, st, a);
    if constexpr (LA == 3 && LB == 3 && ABL == 0) launch_plane_partner<BK, WGM, WGN, TI, TJ, NSTAGE, EP>(a, st, blocks);
    return hipGetLastError();
}

// single limb, KS k groups per workgroup (see k_mfma)
template <int BK, int WGM, int WGN, int TI, int TJ, int NSTAGE, int KS>
hipError_t launch_ksplit(const QMfmaArgs& a, hipStream_t st)
{
    constexpr int TM = WGM * TI * 32, TN = WGN * TJ * 32;
    constexpr int STAGE = (TM + TN) * BK;
    constexpr int RED = (KS - 1) * WGM * WGN * TI * TJ * 16 * 64 * 4;
    const int lds = KS * NSTAGE * STAGE > RED ? KS * NSTAGE * STAGE : RED;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = qg_lds_attr((const void*)k_mfma<1, 1, BK, WGM, WGN, TI, TJ, NSTAGE, 0, false, 1, 1, false, KS>, lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (a.Mp / TM) * (a.Np / TN);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll || a.Kp % (BK * KS) || a.Mp % TM || a.Np % TN || a.has_ep || a.kara) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_mfma<1, 1, BK, WGM, WGN, TI, TJ, NSTAGE, 0, false, 1, 1, false, KS>), dim3((unsigned)blocks), dim3(64 * WGM * WGN * KS), lds, st, a);
    return hipGetLastError();
}

} // namespace

QMfmaCfg qg_mfma_pick(int LA, int LB, int64_t M, int64_t N, uint32_t opt_flags)
{
    QMfmaCfg c = {0, 0, 0, 0};
    if (LA < 1 || LB < 1 || LA > 3 || LB > 3) return c;
    if (LA == 1 && LB == 1) {
        // 256x256 tiles halve the L2->LDS traffic per MAC; use them once they fill the 256 CUs
        const int64_t big = ((M + 255) / 256) * ((N + 255) / 256);
        // two wave groups alternating on the matrix cores, 128-byte k-tiles (qg_mfma_pp.hip); QG_OPT_LOCKSTEP_TILES keeps k_mfma16
        if (big >= 256) return (opt_flags & QG_OPT_LOCKSTEP_TILES) ? QMfmaCfg{2, 256, 256, 64} : QMfmaCfg{9, 256, 256, 128};
        // small problems: 64x64 tiles once 128x128 ones would leave more than half of the 256 CUs without a workgroup
        // (1024^2: 64 -> 256 workgroups)
        const int64_t mid = ((M + 127) / 128) * ((N + 127) / 128);
        static const bool no_small = QG_DIAG_ENV("QG_NO_SMALL_TILES");   // A/B switch for tools/measure_small.py
        if (!no_small && mid <= 128 && ((M + 63) / 64) * ((N + 63) / 64) > mid) {
            // 128-byte k-tiles: these launches are bound by the per-k-tile barrier, DMA issue and exposed fragment reads of a
            // one-wave-per-SIMD workgroup (~0.2 us per 64-byte k-tile whatever the ring depth), so half as many k-tiles:
            // 1024^3 5.97 -> 5.35 us, 512^2 x 4096 13.3 -> 10.2 us (profiles/r03n_small_ring.jsonl).  QG_BK64 for A/B.
            static const bool bk64 = QG_DIAG_ENV("QG_BK64");
            return bk64 ? QMfmaCfg{5, 64, 64, 64} : QMfmaCfg{7, 64, 64, 128};
        }
        {   // at most one workgroup per CU: 128-byte k-tiles here as well (2048^2 x 8192 55.3 -> 44.8 us, 1792^2 x 4096 30.8 -> 24.4 us,
            // 2048^3 15.3 -> 15.2 us); with more workgroups the 96 KB LDS image would cost co-residency.  QG_BK64 for A/B.
            static const bool bk64 = QG_DIAG_ENV("QG_BK64");
            if (!bk64 && mid <= 256) return QMfmaCfg{8, 128, 128, 128};
        }
        return QMfmaCfg{1, 128, 128, 64};
    }
    {   // limb kernels: the same small-problem rule (1024^2 outputs: 64 -> 256 workgroups)
        static const bool no_small = QG_DIAG_ENV("QG_NO_SMALL_TILES");
        const int64_t mid = ((M + 127) / 128) * ((N + 127) / 128);
        // 3 x 3 limbs with at least a tile per CU: the two-group kernel (qg_mfma_ppl.hip; same packed layout as variant 3)
        // (2 x 2: unless the operands are Karatsuba-eligible, which plan_geometry decides and then returns to variant 3)
        if (((LA == 3 && LB == 3) || (LA == 2 && LB == 2)) && mid >= 256 && !(opt_flags & QG_OPT_LOCKSTEP_TILES)) return QMfmaCfg{10, 128, 128, 64};
        if (!no_small && mid <= 128 && ((M + 63) / 64) * ((N + 63) / 64) > mid) return QMfmaCfg{6, 64, 64, 64};
    }
    return QMfmaCfg{3, 128, 128, 64};
}

// QG_ABLATE=1..5 selects diagnostic variants of the two benchmarked kernels (results are WRONG by construction; used only
// by tools/ablate.py to price the phases of the loop).  They exist in the diagnostic build only (libqugemm_diag.so, -DQG_DIAG):
// the product library neither reads the variable nor contains the variants.
#ifdef QG_DIAG
static int ablation()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("QG_ABLATE"); v = e ? atoi(e) : 0; }
    return v;
}
#else
static constexpr int ablation() { return 0; }
#endif

hipError_t qg_launch_mfma(int LA, int LB, const QMfmaArgs& a_in, hipStream_t st)
{
    if (a_in.variant == 10 && qg_mfma_ppl_applies(LA, LB, a_in)) return qg_launch_mfma_ppl(LA, a_in, st);
    QMfmaArgs a_v3;
    if (a_in.variant == 10) { a_v3 = a_in; a_v3.variant = 3; }   // (fused chain, narrow C: the lock-step kernel on the same packed layout)
    const QMfmaArgs& a = a_in.variant == 10 ? a_v3 : a_in;
#ifdef QG_DIAG
    if (const int abl = ablation(); abl > 0 && abl < 16) {
        if (LA == 3 && LB == 3 && a.variant == 3) {   // (the diagnostic variants exist for the 128x128-tile geometry only)
            switch (abl) {
            case 1: return launch<3, 3, 64, 2, 4, 2, 1, 3, 1>(a, st);
            case 2: return launch<3, 3, 64, 2, 4, 2, 1, 3, 2>(a, st);
            case 3: return launch<3, 3, 64, 2, 4, 2, 1, 3, 3>(a, st);
            case 4: return launch<3, 3, 64, 2, 4, 2, 1, 3, 4>(a, st);
            case 6: return launch<3, 3, 64, 2, 4, 2, 1, 3, 6>(a, st);
            case 15: return launch<3, 3, 64, 2, 4, 2, 1, 3, 15>(a, st);   // the compiler's own issue order (correct results): A/B for interleave_hint
            default: return launch<3, 3, 64, 2, 4, 2, 1, 3, 5>(a, st);
            }
        }
        if (LA == 1 && LB == 1 && a.variant == 2) {
            switch (abl) {
            case 1: return launch<1, 1, 64, 2, 4, 4, 2, 3, 1>(a, st);
            case 2: return launch<1, 1, 64, 2, 4, 4, 2, 3, 2>(a, st);
            case 3: return launch<1, 1, 64, 2, 4, 4, 2, 3, 3>(a, st);
            case 4: return launch<1, 1, 64, 2, 4, 4, 2, 3, 4>(a, st);
            case 6: return launch<1, 1, 64, 2, 4, 4, 2, 3, 6>(a, st);
            default: return launch<1, 1, 64, 2, 4, 4, 2, 3, 5>(a, st);
            }
        }
    }
#endif
    if (LA == 1 && LB == 1) {
        // single limb: v_mfma_i32_16x16x64_i8 measured 8 % faster than 32x32x32 at the same tiles
        // (0.283 vs 0.308 ms at 8192x8192x4096, profiles/r01n_ablation_mfma_shape.log); QG_ABLATE=32 keeps the other
        if (a.variant == 9) return a.has_ep ? hipErrorInvalidValue : qg_launch_mfma_pp(a, st);
        if (a.variant == 5) return launch<1, 1, 64, 2, 2, 1, 1, 3>(a, st);   // 64x64 tiles, one 32x32 MFMA tile per wave
        if (a.variant == 7) {   // the same on 128-byte k-tiles
            // ... with two k groups per workgroup for long-k problems of at most one workgroup per CU (diag: QG_NO_KSPLIT for A/B).
            // Measured (us, split / not split): 512^2 x 4096 9.30 / 10.20, 256^2 x 4096 9.16 / 10.03 — but 1024^3 (8 k-tiles)
            // 5.50 / 5.30, and with more than 256 workgroups the 8-wave, 96 KB workgroups no longer share a CU: 1536 x 1024^2
            // 9.34 / 6.01, 2048 x 1024 x 512 8.17 / 5.02 (profiles/r04_small_ksplit.jsonl).
            static const bool no_ks = QG_DIAG_ENV("QG_NO_KSPLIT");
            if (!no_ks && !a.has_ep && (a.Kp / 128) % 2 == 0 && a.Kp / 128 >= 16 && (a.Mp / 64) * (a.Np / 64) <= 256)
                return launch_ksplit<128, 2, 2, 1, 1, 3, 2>(a, st);
            return launch<1, 1, 128, 2, 2, 1, 1, 3>(a, st);
        }
        if (a.variant == 8) return launch<1, 1, 128, 2, 2, 2, 2, 3>(a, st);   // 128x128 tiles on 128-byte k-tiles
#ifdef QG_DIAG
        if (ablation() == 32) return a.variant == 2 ? launch<1, 1, 64, 2, 4, 4, 2, 3>(a, st) : launch<1, 1, 64, 2, 2, 2, 2, 3>(a, st);
#endif
        if (a.variant == 2) {
            static const bool shallow = QG_DIAG_ENV("QG_NO_DEEP");   // A/B: one k-tile in flight instead of two
            if (shallow && !a.has_ep) return launch16<1, 1, 2, 4, 8, 4, true, false, true, 0>(a, st);
            return launch16<1, 1, 2, 4, 8, 4, true>(a, st);
        }
        return launch<1, 1, 64, 2, 2, 2, 2, 3>(a, st);  // 128x128 tiles (small problems): the 32x32x32 kernel is the faster one there
    }
    if (a.variant == 6 && !a.has_ep && !a.kara && (a.Mp / 64) * (a.Np / 64) <= 256) {
        // At most one workgroup per CU anyway: a 5-stage LDS ring (4 k-tiles in flight; a k-tile of these kernels is a few hundred
        // cycles, less than one trip to L2).  512^2 x 4096 int<8,8>: 39.8 -> 27.9 us; with more workgroups than CUs the ring's LDS
        // would cost co-residency (1536 x 1024 x 1024: 21.2 -> 24.4 us), and the single-limb 64x64 kernel gained nothing from it
        // (its k-tile is bound by the dependent MFMA pair and the barrier).  QG_STAGES3 keeps three stages (A/B).
        static const bool three = QG_DIAG_ENV("QG_STAGES3");
        if (!three) switch (LA * 10 + LB) {
            case 22: return launch<2, 2, 64, 2, 2, 1, 1, 5>(a, st);
            case 23: return launch<2, 3, 64, 2, 2, 1, 1, 5>(a, st);
            case 32: return launch<3, 2, 64, 2, 2, 1, 1, 5>(a, st);
            case 33: return launch<3, 3, 64, 2, 2, 1, 1, 5>(a, st);
            default: break;
            }
    }
    if (a.variant == 6) {   // 64x64 tiles, 4 waves, one 32x32 MFMA tile per wave
        switch (LA * 10 + LB) {
        case 12: return launch<1, 2, 64, 2, 2, 1, 1, 3>(a, st);
        case 21: return launch<2, 1, 64, 2, 2, 1, 1, 3>(a, st);
        case 22: return launch<2, 2, 64, 2, 2, 1, 1, 3>(a, st);
        case 13: return launch<1, 3, 64, 2, 2, 1, 1, 3>(a, st);
        case 31: return launch<3, 1, 64, 2, 2, 1, 1, 3>(a, st);
        case 23: return launch<2, 3, 64, 2, 2, 1, 1, 3>(a, st);
        case 32: return launch<3, 2, 64, 2, 2, 1, 1, 3>(a, st);
        case 33: return launch<3, 3, 64, 2, 2, 1, 1, 3>(a, st);
        default: return hipErrorInvalidValue;
        }
    }
    if (a.variant == 3 && a.kara && LA == 2 && LB == 2 && !a.has_ep) {
        static const bool kara32 = QG_DIAG_ENV("QG_KARA32");   // A/B: the Karatsuba kernel on 32x32x32
        if (!kara32) return launch16<2, 2, 2, 4, 4, 2, false, false, true, 1, 2, 2, true>(a, st);
    }
    if (a.variant == 3 && !a.kara) {
        // 128x128 limb tiles: v_mfma_i32_16x16x64_i8 with the row-step fragment pipeline (k_mfma16, DBUF = false) against
        // 32x32x32 (k_mfma) at 4096^3, same box, ms: 3x3 0.388 / 0.427, 2x3 0.290 / 0.329, 3x2 0.295 / 0.332, 2x2 0.221 / 0.232,
        // 1x3 0.194 / 0.199, 3x1 0.200 / 0.206, 1x2 0.128 / 0.176, 2x1 0.130 / 0.178 (the chip holds a higher clock on the
        // small shape; profiles/r03j_limb_shapes.log, r03k_pipeline.log).  QG_LIMB32 keeps 32x32x32 everywhere (A/B).
        static const bool limb32 = QG_DIAG_ENV("QG_LIMB32");
        if (!limb32) switch (LA * 10 + LB) {
            case 33: {
                static const bool spread = QG_DIAG_ENV("QG_DMA_SPREAD");   // A/B: LDS-DMA issues over all four row steps
                if (spread && !a.has_ep) return launch16<3, 3, 2, 4, 4, 2, false, false, true, 0>(a, st);
                return launch16<3, 3, 2, 4, 4, 2, false>(a, st);
            }
            case 23: if (!a.has_ep) return launch16<2, 3, 2, 4, 4, 2, false>(a, st); break;
            case 32: if (!a.has_ep) return launch16<3, 2, 2, 4, 4, 2, false>(a, st); break;
            case 12: if (!a.has_ep) return launch16<1, 2, 2, 4, 4, 2, false>(a, st); break;
            case 21: if (!a.has_ep) return launch16<2, 1, 2, 4, 4, 2, false>(a, st); break;
            case 13: if (!a.has_ep) return launch16<1, 3, 2, 4, 4, 2, false>(a, st); break;
            case 31: if (!a.has_ep) return launch16<3, 1, 2, 4, 4, 2, false>(a, st); break;
            case 22: if (!a.has_ep) return launch16<2, 2, 2, 4, 4, 2, false>(a, st); break;
            default: break;
            }
    }
    switch (LA * 10 + LB) {
    case 12: return launch<1, 2, 64, 2, 4, 2, 1, 3>(a, st);
    case 21: return launch<2, 1, 64, 2, 4, 2, 1, 3>(a, st);
    case 22: return launch<2, 2, 64, 2, 4, 2, 1, 3>(a, st);
    case 13: return launch<1, 3, 64, 2, 4, 2, 1, 3>(a, st);
    case 31: return launch<3, 1, 64, 2, 4, 2, 1, 3>(a, st);
    case 23: return launch<2, 3, 64, 2, 4, 2, 1, 3>(a, st);
    case 32: return launch<3, 2, 64, 2, 4, 2, 1, 3>(a, st);
    case 33: return launch<3, 3, 64, 2, 4, 2, 1, 3>(a, st);
    default: return hipErrorInvalidValue;
    }
}
