// qg_pack.hip — layout kernels (gfx950): reference-layout tensors <-> device-private packed
// operands, synthetic operand generation, packed C -> reference layout.
//
// Reference layout (what Qu<dim<…>,T>::data.data() holds): column-major array of int32/int64
// raw values, complex = struct {real; imag;}  (/root/reference/include/QuBLAS.h:2680-2692,
// :353, :2512-2513).  All of these are HBM-bound byte movers; they are not on the timed path
// (operands are packed once and stay resident), so they are written for coalescing, not more.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "qg_kernels.h"

namespace {

__device__ __forceinline__ int64_t load_host_part(const char* src, int64_t e, int elem_bytes, int off, int sb)
{
    const char* p = src + e * elem_bytes + off;
    return sb == 4 ? (int64_t)(*(const int32_t*)p) : *(const int64_t*)p;
}

__device__ __forceinline__ void store_container(char* dst, int64_t idx, int cbytes, int64_t v)
{
    switch (cbytes) {
    case 1: ((int8_t*)dst)[idx] = (int8_t)v; break;
    case 2: ((int16_t*)dst)[idx] = (int16_t)v; break;
    case 4: ((int32_t*)dst)[idx] = (int32_t)v; break;
    default: ((int64_t*)dst)[idx] = v; break;
    }
}

__device__ __forceinline__ int64_t load_container(const char* src, int64_t idx, int cbytes)
{
    switch (cbytes) {
    case 1: return ((const int8_t*)src)[idx];
    case 2: return ((const int16_t*)src)[idx];
    case 4: return ((const int32_t*)src)[idx];
    default: return ((const int64_t*)src)[idx];
    }
}

// Qu_s(double) on the device: v = m * 2^e exactly (53-bit m), raw = m * 2^(e+F): a left shift or ONE rounding of m
// with the format's QuMode, then its OfMode.  Zero, NaN and infinity give 0 (loadFromDouble, QuBLAS.h:670-674).
__device__ __forceinline__ int64_t quantize_f64(double v, int W, int S, int F, int Q, int O)
{
    if (v == 0.0 || v != v || (v - v) != 0.0) return 0;
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const bool neg = (bits >> 63) != 0;
    int ex = (int)((bits >> 52) & 0x7ff);
    int64_t m = (int64_t)(bits & 0xfffffffffffffull);
    if (ex == 0) ex = 1; else m |= (int64_t)1 << 52;
    const int e = ex - 1075 + F;
    if (neg) m = -m;
    const int64_t hi = ((int64_t)1 << W) - 1, lo = S ? -((int64_t)1 << W) : 0;
    if (e >= 0) {
        if (e <= 9) return qg_overflow<int64_t>(m << e, O, W, S, lo, hi);
        // |value| >= 2^62 exceeds every supported format: saturate by sign, or wrap the (exact) low bits
        switch (O) {
        case QG_SAT_TCPL: return neg ? lo : hi;
        case QG_SAT_ZERO: return 0;
        case QG_SAT_SMGN: return neg ? (S ? -hi : 0) : hi;
        default: return qg_overflow<int64_t>(e < 64 ? (int64_t)((uint64_t)m << e) : 0, O, W, S, lo, hi);
        }
    }
    int d = -e;
    if (d > 62) { m = neg ? -1 : 1; d = 8; }  // every mantissa bit lies below the rounding position
    return qg_overflow<int64_t>(qg_round<int64_t>(m, d, Q), O, W, S, lo, hi);
}

// write one logical value (r,k,part) into the packed operand
__device__ __forceinline__ unsigned put_packed(const QPackedGeom& p, char* dst, int part, int64_t r, int64_t k, int64_t v)  // r, k: indices inside the part
{
    unsigned mask = 0;   // bit l: limb l of this value is non-zero
    if (p.limbs == 0) {
        store_container(dst, ((int64_t)part * p.rows_p + r) * p.K_p + k, p.cbytes, v);
    } else {
        // pre-tiled, pre-swizzled limb planes (see QPackedGeom)
        // complex operands stack their parts along the row axis: part q occupies rows [q*rows_p, (q+1)*rows_p)
        r += (int64_t)part * p.rows_p;
        const int64_t nk = p.K_p / p.bk;
        const int rl = (int)(r % p.tr), kl = (int)(k % p.bk);
        const int cpr = p.bk / 16, rpb = 256 / p.bk;
        int sw = (rl / rpb) % cpr;
        if (p.bk == 64) sw = (0x78 >> (2 * sw)) & 3;  // the kernels' swz<64>(): {0,2,3,1}
        const int slot = (kl / 16) ^ sw;
        const int64_t blk = ((r / p.tr) * nk + k / p.bk) * p.limbs;
        if (p.digit6) {   // Karatsuba layout: unsigned base-64 digits of the (already biased, non-negative) value
            for (int l = 0; l < p.limbs; ++l) {
                ((int8_t*)dst)[((blk + l) * p.tr + rl) * p.bk + slot * 16 + (kl & 15)] = (int8_t)(v & 63);
                v >>= 6;
            }
            return 0;
        }
        for (int l = 0; l < p.limb0; ++l) v = (v - (int64_t)(int8_t)(v & 0xff)) >> 8;   // a limb group: the lower digits live elsewhere
        for (int l = 0; l < p.limbs; ++l) {
            int64_t d = (int64_t)(int8_t)(v & 0xff); // balanced digit in [-128,127]
            ((int8_t*)dst)[((blk + l) * p.tr + rl) * p.bk + slot * 16 + (kl & 15)] = (int8_t)d;
            mask |= d ? (1u << l) : 0u;
            v = (v - d) >> 8;
        }
    }
    return mask;
}

// Fast path of k_pack for the common operand: real, 32-bit host elements, balanced int8 limb planes with 64- or 128-byte k-tiles.
// One 64-row x 64-k block of the limb layout per workgroup; a thread owns 16 consecutive k of one row, i.e. one 16-byte
// chunk per limb plane (one 16-byte store each).  R_FAST: rows are the contiguous host axis (column-major A) -> the 64
// lanes of a wave take 64 consecutive rows (256-byte coalesced loads per k); otherwise k is contiguous (B, transposed A)
// -> 4 lanes cover the 64 k-bytes of a row and a wave stores 1 KiB contiguously per plane.  Same bytes as k_pack
// (tests/test_gpu_resources.py::test_fast_pack_paths_write_the_bytes_of_the_generic_kernels; QG_NO_FAST_PACK=1 disables).
// VEC (k contiguous, source and leading dimension 16-byte aligned): a thread's 16 consecutive k are four 16-byte loads instead
// of sixteen 4-byte ones (the B operand of 4096^2 int<8,8>: 0.042 -> see profiles/).
// F64 (quantise-on-load, qgemul_pack_f64): the source is a tensor of doubles; every value goes through quantize_f64 (the element
// type's own QuMode, then OfMode: Qu_s(double), QuBLAS.h:2387-2393) and then takes the same limb split and the same stores.
template <bool R_FAST, bool VEC = false, bool F64 = false>
__global__ __launch_bounds__(256) void k_pack_limb32(QOperandGeom g, QPackedGeom p, const int32_t* __restrict__ src, int8_t* __restrict__ dst,
                                                     int check, int* flag)
{
    // grid-stride over the 64 x 64 blocks: a workgroup ORs what its blocks saw into ONE plane-mask atomic at the end (an atomic
    // per wave and block — 16 384 for a 4096^2 operand, all on the trailer's memory channel — cost 0.06-0.17 ms on top of a
    // 0.03 ms pack)
    const int kt = (int)(p.K_p / 64);
    const int64_t nblk = (int64_t)kt * (p.rows_p / 64);
    const int t = threadIdx.x;
    const int row_l = R_FAST ? (t & 63) : (t >> 2), kc = R_FAST ? (t >> 6) : (t & 3);
    const int W = g.W[0];
    const int64_t lo = g.S[0] ? -((int64_t)1 << W) : 0, hi = ((int64_t)1 << W) - 1;
    unsigned mask = 0;
    bool bad = false;
    __shared__ unsigned long long rs_sh[64];
    for (int64_t blk_id = blockIdx.x; blk_id < nblk; blk_id += gridDim.x) {
    const int tk = (int)(blk_id % kt), tr64 = (int)(blk_id / kt);
    const int64_t r = (int64_t)tr64 * 64 + row_l, k0 = (int64_t)tk * 64 + kc * 16;
    int32_t v[16];
    const bool row_in = r < g.rows;
    const int32_t* q = src + r * g.rs + (k0 + g.k0) * g.ks;
    if constexpr (F64) {
        const double* qd = (const double*)src + r * g.rs + (k0 + g.k0) * g.ks;
        if (VEC && row_in && k0 + 16 <= g.K) {      // k contiguous: eight 16-byte loads of two doubles
            const double2* q2 = (const double2*)qd;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double2 x = q2[j];
                v[2 * j] = (int32_t)quantize_f64(x.x, W, g.S[0], g.F[0], g.Q[0], g.O[0]);
                v[2 * j + 1] = (int32_t)quantize_f64(x.y, W, g.S[0], g.F[0], g.Q[0], g.O[0]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = (row_in && k0 + j < g.K) ? (int32_t)quantize_f64(qd[(int64_t)j * g.ks], W, g.S[0], g.F[0], g.Q[0], g.O[0]) : 0;
        }
    } else
    if (VEC && row_in && k0 + 16 <= g.K) {
        const int4* q4 = (const int4*)q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int4 x = q4[j];
            v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) bad |= (v[j] < lo) | (v[j] > hi);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v[j] = (row_in && k0 + j < g.K) ? q[(int64_t)j * g.ks] : 0;
            bad |= (v[j] < lo) | (v[j] > hi);
        }
    }
    if (p.offs) {   // centred operand (QPackedGeom::offs): x - centre, padding stays 0; row sums of the stored values through LDS
        const int32_t bias = (int32_t)p.bias;   // (|bias| < 2^24 and W <= 24 on this path: qg_launch_pack)
        long long part = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (row_in && k0 + j < g.K) v[j] += bias;
            part += v[j];
        }
        if (p.offs != 3) {
            __syncthreads();                       // (the previous block's sums have left rs_sh)
            if (t < 64) rs_sh[t] = 0;
            __syncthreads();
            if (part) atomicAdd(&rs_sh[row_l], (unsigned long long)part);
            __syncthreads();
            if (t < 64 && rs_sh[t] && (int64_t)tr64 * 64 + t < p.rows_p)
                atomicAdd((unsigned long long*)((char*)dst + p.rowsum_off) + (int64_t)tr64 * 64 + t, rs_sh[t]);
        }
    }
    const int rl = (int)(r % p.tr);
    // the kernels' swz<BK>(): 64-byte rows {0,2,3,1}[(row / 4) % 4], 128-byte rows (row / 2) % 8
    const int sw = p.bk == 64 ? (0x78 >> (2 * ((rl >> 2) & 3))) & 3 : (rl >> 1) & 7;
    const int c = p.bk == 64 ? kc : (tk & 1) * 4 + kc;   // 16-byte chunk of this thread inside its k-tile
    const int64_t blk = ((r / p.tr) * (p.K_p / p.bk) + (p.bk == 64 ? tk : tk >> 1)) * p.limbs;
    int8_t* out = dst + (blk * p.tr + rl) * p.bk + ((c ^ sw) * 16);
    for (int l = 0; l < p.limb0; ++l)   // a limb group: the lower digits live elsewhere
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (int32_t)(((int64_t)v[j] - (int32_t)(int8_t)(v[j] & 0xff)) >> 8);
    for (int l = 0; l < p.limbs; ++l) {
        uint32_t w[4] = {0, 0, 0, 0};
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int32_t d = (int32_t)(int8_t)(v[j] & 0xff);   // balanced digit in [-128,127]
            w[j >> 2] |= (uint32_t)(d & 0xff) << (8 * (j & 3));
            any |= (uint32_t)d;
            v[j] = (int32_t)(((int64_t)v[j] - d) >> 8);
        }
        *(uint4*)(out + (int64_t)l * p.tr * p.bk) = make_uint4(w[0], w[1], w[2], w[3]);
        mask |= any ? (1u << l) : 0u;
    }
    }
    if (p.trailer) {   // plane mask of the operand: one atomic per workgroup that saw a non-zero limb
        __shared__ unsigned wmask[4];
#pragma unroll
        for (int o = 32; o; o >>= 1) mask |= __shfl_xor(mask, o);
        if ((t & 63) == 0) wmask[t >> 6] = mask;
        __syncthreads();
        if (t == 0) {
            const unsigned m = wmask[0] | wmask[1] | wmask[2] | wmask[3];
            if (m) atomicOr((unsigned*)(dst + p.trailer) + (blockIdx.x & (QG_MASK_WORDS - 1)), m);
        }
    }
    if (check && bad) atomicOr(flag, 1);
}

// Fast path of k_unpack_c: real, 32-bit containers and 32-bit host elements, tiled packed C.  A column of a tile is tm
// contiguous rows in both layouts, so unpacking is a copy of 16-byte pieces; thread = 4 consecutive rows of one column.
__global__ __launch_bounds__(256) void k_unpack_c32(QCGeom c, const int32_t* __restrict__ packed, int32_t* __restrict__ dst, int vec)
{
    const int64_t m4 = c.M / 4 + (c.M % 4 != 0);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m4 * c.N) return;
    const int64_t n = idx / m4, m = (idx % m4) * 4;
    const int64_t pi = (((m / c.tm) * (c.Np / c.tn) + n / c.tn) * c.tn + n % c.tn) * c.tm + m % c.tm;   // tm % 4 == 0: the 4 rows share a tile
    int32_t* q = dst + m + n * c.ldc;
    if (vec && m + 4 <= c.M) *(int4*)q = *(const int4*)(packed + pi);
    else
        for (int e = 0; e < 4 && m + e < c.M; ++e) q[e] = packed[pi + e];
}

// tile = 64 (k) x 64 (r); 256 threads.  The fast host axis is r when g.rs == 1 (non-transposed A)
// and k otherwise; global reads follow the fast host axis, packed writes always follow k.
__global__ __launch_bounds__(256) void k_pack(QOperandGeom g, QPackedGeom p, const char* __restrict__ src, char* __restrict__ dst,
                                              int check, int* flag, int fill, uint64_t seed, int dist)
{
    __shared__ int64_t tile[64][65];
    const int64_t kt = (p.K_p + 63) / 64, rt = (p.rows_p + 63) / 64;
    int64_t b = blockIdx.x;
    const int64_t tk = b % kt; b /= kt;
    const int64_t tr = b % rt; b /= rt;
    const int part = (int)b;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
    const bool r_fast = (g.rs == 1 && g.ks != 1);
    const int W = g.W[part], S = g.S[part];
    const int64_t lo = S ? -((int64_t)1 << W) : 0, hi = ((int64_t)1 << W) - 1;
    bool bad = false;
    for (int i = ty; i < 64; i += 4) {
        // (a,b) = (fast index, slow index) inside the tile
        int64_t r = tr * 64 + (r_fast ? tx : i);
        int64_t k = tk * 64 + (r_fast ? i : tx);
        int64_t v = 0;
        if (r < g.rows && k < g.K) {
            const int64_t kh = k + g.k0;   // reduction index in the host tensor (k: inside this packed operand)
            if (fill == 2) {
                const double* ds = (const double*)src + (r * g.rs + kh * g.ks) * g.parts + part;
                v = quantize_f64(*ds, W, S, g.F[part], g.Q[part], g.O[part]);
            } else if (fill) {
                // tight host linear index of the element, as a host-side fill of the tensor would see it
                v = qg_synth(W, S, seed, dist, (uint64_t)(r * g.rs + kh * g.ks), part);
            } else {
                v = load_host_part(src, r * g.rs + kh * g.ks, g.elem_bytes, g.off[part], g.sb[part]);
                if (check && (v < lo || v > hi)) bad = true;
            }
            if (p.digit6) v = (v + p.bias) & ((((int64_t)1) << (6 * p.limbs)) - 1);   // padding stays 0
            else if (p.offs) v = (int64_t)((uint64_t)v + (uint64_t)p.bias);          // centred operand: x - centre (padding stays 0)
        }
        if (r_fast) tile[i][tx] = v;  // tile[k_local][r_local]
        else tile[tx][i] = v;         // tile[k_local][r_local] with tx = k_local
    }
    __syncthreads();
    unsigned mask = 0;
    for (int i = ty; i < 64; i += 4) {
        int64_t r = tr * 64 + i, k = tk * 64 + tx;
        if (r < p.rows_p && k < p.K_p) mask |= put_packed(p, dst, part, r, k, tile[tx][i]);
    }
    if (p.digit6) {
        // row sums of the biased values: in the loop above the 64 lanes of a wave held 64 k's of ONE row per step; redo the
        // walk for the sums (values < 2^12, 64 of them: int32 is plenty)
        for (int i = ty; i < 64; i += 4) {
            const int64_t r = tr * 64 + i, k = tk * 64 + tx;
            int sum = (r < p.rows_p && k < p.K_p) ? (int)tile[tx][i] : 0;
#pragma unroll
            for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
            if (tx == 0 && r < p.rows_p && sum)
                atomicAdd((unsigned long long*)(dst + p.rowsum_off) + (int64_t)part * p.rows_p + r, (unsigned long long)sum);
        }
    }
    if (p.offs == 1 || p.offs == 2) {
        // row sums of the centred values (whole values, before a limb group is peeled off): the 64 lanes of a wave hold 64 k's of
        // one row per step
        for (int i = ty; i < 64; i += 4) {
            const int64_t r = tr * 64 + i, k = tk * 64 + tx;
            long long sum = (r < p.rows_p && k < p.K_p) ? (long long)tile[tx][i] : 0;
#pragma unroll
            for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
            if (tx == 0 && r < p.rows_p && sum)
                atomicAdd((unsigned long long*)(dst + p.rowsum_off) + (int64_t)part * p.rows_p + r, (unsigned long long)sum);
        }
    }
    if (p.trailer) {   // plane mask of the operand: one atomic per wave that saw a non-zero limb
#pragma unroll
        for (int o = 32; o; o >>= 1) mask |= __shfl_xor(mask, o);
        if (tx == 0 && mask) atomicOr((unsigned*)(dst + p.trailer) + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (QG_MASK_WORDS - 1)), mask);
    }
    if (bad) atomicOr(flag, 1);
}

// packed C [part][Mp][Np] (n contiguous) -> host column-major (i contiguous)
__global__ __launch_bounds__(256) void k_unpack_c(QCGeom c, const char* __restrict__ packed, char* __restrict__ dst)
{
    __shared__ int64_t tile[64][65];
    const int64_t nt = (c.N + 63) / 64, mt = (c.M + 63) / 64;
    int64_t b = blockIdx.x;
    const int64_t tn = b % nt; b /= nt;
    const int64_t tm = b % mt; b /= mt;
    const int part = (int)b;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int64_t m = tm * 64 + i, n = tn * 64 + tx;
        int64_t v = 0;
        if (m < c.M && n < c.N) {
            int64_t idx;
            if (c.tm == 0) idx = ((int64_t)part * c.Mp + m) * c.Np + n;
            else idx = ((((int64_t)part * (c.Mp / c.tm) + m / c.tm) * (c.Np / c.tn) + n / c.tn) * c.tn + n % c.tn) * c.tm + m % c.tm;
            v = load_container(packed, idx, c.cbytes);
        }
        tile[i][tx] = v; // tile[m_local][n_local]
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int64_t m = tm * 64 + tx, n = tn * 64 + i;
        if (m < c.M && n < c.N) {
            char* q = dst + (m + n * c.ldc) * c.elem_bytes + c.off[part];
            int64_t v = tile[tx][i];
            if (c.sb[part] == 4) *(int32_t*)q = (int32_t)v;
            else *(int64_t*)q = v;
        }
    }
}

// complex linear class: D = [A_re; A_im] x [B_re | B_im] raw dot products (tiled int64, as the MFMA kernel stores a
// cbytes = 8 C), combined into re = (P1 << s0) - (P2 << s1), im = (P3 << s2) + (P4 << s3), one round + overflow per part
__global__ __launch_bounds__(256) void k_cplx_combine(QCplxCombine g)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int64_t m = idx / g.N, n = idx % g.N;
    auto D = [&](int64_t r, int64_t c) {
        const int64_t tn = g.Np / g.tn;
        return g.D[(((r / g.tm) * tn + c / g.tn) * g.tn + c % g.tn) * g.tm + r % g.tm];
    };
    const int64_t P1 = D(m, n), P2 = D(g.Mh + m, g.Nh + n), P3 = D(m, g.Nh + n), P4 = D(g.Mh + m, n);
    const int64_t re = qg_shl<int64_t>(P1, g.sh[0]) - qg_shl<int64_t>(P2, g.sh[1]);
    const int64_t im = qg_shl<int64_t>(P3, g.sh[2]) + qg_shl<int64_t>(P4, g.sh[3]);
    store_container(g.C, m * g.N + n, g.cbytes, qg_step<int64_t>(re, g.to_c[0]));
    store_container(g.C, g.M * g.N + m * g.N + n, g.cbytes, qg_step<int64_t>(im, g.to_c[1]));
}

// BitStream export (QuBLAS.h:4811-4827): thread = one output element position.  pos -> source element (tensor-level
// chunk reversal) -> its raw value -> `width` characters, MSB first, element-level chunk reversal -> out[pos*width ...].
// ASCII: consecutive lanes write consecutive width-byte runs.  Packed: the stream is M*N*width bits; a thread ORs its
// bits into the bytes they fall into (atomicOr on 32-bit words, output zeroed by the launcher).
__global__ __launch_bounds__(256) void k_bitstream(QBitsArgs g)
{
    const int64_t n = g.c.M * g.c.N;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    int64_t src = pos;
    if (g.tensor_chunk > 0) {
        const int64_t nch = n / g.tensor_chunk, ch = pos / g.tensor_chunk;
        src = (nch - 1 - ch) * g.tensor_chunk + pos % g.tensor_chunk;
    }
    const uint64_t v = (uint64_t)load_container(g.packed, qg_c_index(g.c, 0, src % g.c.M, src / g.c.M), g.c.cbytes);
    const int w = g.width;
    // bits of the element string after the element-level reversal, first character in bit w-1 of `e`
    uint64_t e = 0;
    if (g.elem_chunk > 0) {
        const int nch = w / g.elem_chunk;
        const uint64_t mask = g.elem_chunk >= 64 ? ~0ull : ((1ull << g.elem_chunk) - 1);
        for (int q = 0; q < nch; ++q)   // output chunk q (from the left) = source chunk nch-1-q (from the left)
            e |= ((v >> (q * g.elem_chunk)) & mask) << ((nch - 1 - q) * g.elem_chunk);
    } else {
        e = w >= 64 ? v : (v & ((1ull << w) - 1));
    }
    if (!g.packed_bits) {
        char* o = g.out + pos * w;
        for (int j = 0; j < w; ++j) o[j] = ((e >> (w - 1 - j)) & 1) ? '1' : '0';
        return;
    }
    uint32_t* o32 = (uint32_t*)g.out;
    const int64_t b0 = pos * w;
    for (int j = 0; j < w; ++j) {
        if (!((e >> (w - 1 - j)) & 1)) continue;
        const int64_t b = b0 + j;                    // stream bit index: byte b/8, bit 7 - b%8 (little-endian words)
        atomicOr(o32 + (b >> 5), 1u << (((b >> 3) & 3) * 8 + (7 - (b & 7))));
    }
}

// The same for a complex tensor (QuBLAS.h:2553-2556: "(" + re + ", " + im + ")"): the host has folded the element-level
// chunk reversal into g.tab.  ASCII: all width characters; packed: only the binary characters, in stream order.
__global__ __launch_bounds__(256) void k_bitstream_cplx(QBitsCplxArgs g)
{
    const int64_t n = g.c.M * g.c.N;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    int64_t src = pos;
    if (g.tensor_chunk > 0) {
        const int64_t nch = n / g.tensor_chunk, ch = pos / g.tensor_chunk;
        src = (nch - 1 - ch) * g.tensor_chunk + pos % g.tensor_chunk;
    }
    const int64_t m = src % g.c.M, col = src / g.c.M;
    const uint64_t re = (uint64_t)load_container(g.packed, qg_c_index(g.c, 0, m, col), g.c.cbytes);
    const uint64_t im = (uint64_t)load_container(g.packed, qg_c_index(g.c, 1, m, col), g.c.cbytes);
    if (!g.packed_bits) {
        char* o = g.out + pos * g.width;
        for (int j = 0; j < g.width; ++j) {
            const int t = g.tab[j];
            char ch;
            if (t < 64) ch = ((re >> t) & 1) ? '1' : '0';
            else if (t < 128) ch = ((im >> (t - 64)) & 1) ? '1' : '0';
            else ch = t == QG_BITS_LIT_OPEN ? '(' : t == QG_BITS_LIT_COMMA ? ',' : t == QG_BITS_LIT_SPACE ? ' ' : ')';
            o[j] = ch;
        }
        return;
    }
    uint32_t* o32 = (uint32_t*)g.out;
    int64_t b = pos * g.nbits;
    for (int j = 0; j < g.width; ++j) {
        const int t = g.tab[j];
        if (t >= 128) continue;
        const bool one = t < 64 ? ((re >> t) & 1) : ((im >> (t - 64)) & 1);
        if (one) atomicOr(o32 + (b >> 5), 1u << (((b >> 3) & 3) * 8 + (7 - (b & 7))));
        ++b;
    }
}

// composite linear plans: see QLinCombine (qg_kernels.h).  One thread per element, slabs and C share ONE layout: a linear pass.
template <class SlabT, class AccT>
__global__ __launch_bounds__(256) void k_lin_combine(QLinCombine g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n) return;
    using UAcc = std::conditional_t<sizeof(AccT) == 16, qg_u128, uint64_t>;   // (wrapping sums: with centred operands sum a'b' may pass the accumulator where sum a b does not)
    AccT s = g.acc_in ? ((const AccT*)g.acc_in)[i] : (AccT)0;
#pragma unroll
    for (int j = 0; j < QG_MAX_SLABS; ++j)
        if (j < g.n_slabs) s = (AccT)((UAcc)s + ((UAcc)(AccT)(((const SlabT*)g.slab[j])[i]) << g.sh[j]));
    if (g.acc_out) { ((AccT*)g.acc_out)[i] = s; return; }
    if (g.rsA) {   // centred operands: sum a b = sum a'b' - biasB rsA[row] - biasA rsB[col] + K biasA biasB (wrapping: the sum itself fits)
        // (tile sizes are powers of two and there are fewer than 2^32 tiles: shifts, masks and ONE 32-bit division — with three 64-bit
        //  divisions this pass was as long as the uint8 GEMM in front of it)
        const int ltm = __ffs(g.tm) - 1, ltn = __ffs(g.tn) - 1;
        const uint32_t t = (uint32_t)(i >> (ltm + ltn)), tmq = t / (uint32_t)g.tiles_n, tnq = t - tmq * (uint32_t)g.tiles_n;
        const int64_t row = (int64_t)tmq * g.tm + (i & (g.tm - 1)), col = (int64_t)tnq * g.tn + ((i >> ltm) & (g.tn - 1));
        using U = std::conditional_t<sizeof(AccT) == 16, qg_u128, uint64_t>;
        s = (AccT)((U)s + (U)(AccT)g.corr * (U)(AccT)g.biasA * (U)(AccT)g.biasB - (U)(AccT)g.biasB * (U)(AccT)g.rsA[row] - (U)(AccT)g.biasA * (U)(AccT)g.rsB[col]);   // (corr: K)
    }
    if constexpr (sizeof(AccT) == 16) {
        // wide plans: the reference's multi-word conversion (qg_step_w), containers of up to 16 bytes
        const qg_i128 r = qg_step_w(s, g.to_c);
        if (g.cbytes == 16) {
            ((uint64_t*)g.out)[2 * i] = (uint64_t)(qg_u128)r;
            ((uint64_t*)g.out)[2 * i + 1] = (uint64_t)((qg_u128)r >> 64);
        } else {
            store_container((char*)g.out, i, g.cbytes, (int64_t)r);
        }
    } else {
        store_container((char*)g.out, i, g.cbytes, (int64_t)qg_step<AccT>(s, g.to_c));
    }
}

// packed C of a wide plan -> host layout: 16-byte containers are two little-endian words (ArbiInt<65..128>, QuBLAS.h:572-573);
// one thread per element, rows fastest (the host's order)
__global__ __launch_bounds__(256) void k_unpack_c_w(QCGeom c, const char* __restrict__ packed, char* __restrict__ dst)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= c.M * c.N * c.parts) return;
    const int part = (int)(idx / (c.M * c.N));
    const int64_t e = idx % (c.M * c.N), m = e % c.M, n = e / c.M;
    const int64_t pi = qg_c_index(c, part, m, n);
    char* q = dst + (m + n * c.ldc) * c.elem_bytes + c.off[part];
    uint64_t lo, hi;
    if (c.cbytes == 16) {
        lo = ((const uint64_t*)packed)[2 * pi];
        hi = ((const uint64_t*)packed)[2 * pi + 1];
    } else {
        const int64_t v = load_container(packed, pi, c.cbytes);
        lo = (uint64_t)v;
        hi = (uint64_t)(v >> 63);
    }
    if (c.sb[part] == 4) *(int32_t*)q = (int32_t)lo;
    else if (c.sb[part] == 8) *(int64_t*)q = (int64_t)lo;
    else { ((uint64_t*)q)[0] = lo; ((uint64_t*)q)[1] = hi; }
}

} // namespace

hipError_t qg_launch_lin_combine(const QLinCombine& g, hipStream_t st)
{
    if (g.n <= 0) return hipSuccess;
    const int64_t blocks = (g.n + 255) / 256;
    if (blocks > 0x7fffffffll || g.n_slabs < 1 || g.n_slabs > QG_MAX_SLABS) return hipErrorInvalidValue;
    if (g.rsA && (g.tm <= 0 || g.tn <= 0 || (g.tm & (g.tm - 1)) || (g.tn & (g.tn - 1)) || g.tiles_n <= 0 || g.tiles_n > 0x7fffffffll)) return hipErrorInvalidValue;
    if (g.wide) {
        if (g.slab_bytes == 4) hipLaunchKernelGGL((k_lin_combine<int32_t, qg_i128>), dim3((unsigned)blocks), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_lin_combine<int64_t, qg_i128>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    } else if (g.slab_bytes == 4) hipLaunchKernelGGL((k_lin_combine<int32_t, int64_t>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_lin_combine<int64_t, int64_t>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError();
}

hipError_t qg_launch_bitstream_cplx(const QBitsCplxArgs& a, hipStream_t st)
{
    const int64_t n = a.c.M * a.c.N;
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (a.packed_bits) {
        const int64_t bytes = ((n * a.nbits + 7) / 8 + 3) / 4 * 4;
        if (hipError_t e = hipMemsetAsync(a.out, 0, (size_t)bytes, st); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bitstream_cplx, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t qg_launch_bitstream(const QBitsArgs& a, hipStream_t st)
{
    const int64_t n = a.c.M * a.c.N;
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (a.packed_bits) {
        const int64_t bytes = ((n * a.width + 7) / 8 + 3) / 4 * 4;
        if (hipError_t e = hipMemsetAsync(a.out, 0, (size_t)bytes, st); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bitstream, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t qg_launch_cplx_combine(const QCplxCombine& g, hipStream_t st)
{
    const int64_t n = g.M * g.N;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cplx_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    return hipGetLastError();
}

static hipError_t zero_trailer(const QPackedGeom& p, void* dst, hipStream_t st)
{
    if (p.offs == 1)   // centred operand: its row sums (anywhere behind the planes; sub-operands of composite plans: the caller's)
        if (hipError_t e = hipMemsetAsync((char*)dst + p.rowsum_off, 0, (size_t)p.rows_p * 8, st); e != hipSuccess) return e;
    if (!p.trailer) return hipSuccess;
    const size_t bytes = QG_TRAILER_BYTES + (p.digit6 ? (size_t)p.rows_p * 8 : 0);   // (row sums sit right behind the trailer)
    return hipMemsetAsync((char*)dst + p.trailer, 0, bytes, st);
}

hipError_t qg_launch_pack(const QOperandGeom& g, const QPackedGeom& p, const void* src, void* dst, int check_range,
                          int* range_flag, hipStream_t st, int generic)
{
    int64_t blocks = ((p.K_p + 63) / 64) * ((p.rows_p + 63) / 64) * g.parts;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (hipError_t e = zero_trailer(p, dst, st); e != hipSuccess) return e;
    const bool no_fast = generic != 0;   // QG_OPT_GENERIC_LAYOUT: the any-format kernel (byte-identical; the equivalence test)
    const bool centred_fast = !p.offs || (g.W[0] <= 24 && p.bias > -(1ll << 24) && p.bias < (1ll << 24));   // (x - centre within int32)
    if (!no_fast && g.parts == 1 && g.elem_bytes == 4 && g.sb[0] == 4 && g.off[0] == 0 && p.limbs >= 1 && p.limbs <= 3 && !p.digit6 && centred_fast && (p.bk == 64 || p.bk == 128) &&
        p.tr % 64 == 0 && p.rows_p % p.tr == 0 && p.K_p % p.bk == 0 && g.W[0] <= 30 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        const int64_t nblk = (p.K_p / 64) * (p.rows_p / 64);
        const unsigned nb = (unsigned)(nblk < 2048 ? nblk : 2048);   // grid-stride beyond 8 workgroups per CU
        if (g.rs == 1 && g.ks != 1) hipLaunchKernelGGL(k_pack_limb32<true>, dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, check_range, range_flag);
        else if (g.ks == 1 && g.rs % 4 == 0 && g.k0 % 4 == 0 && ((uintptr_t)src & 15) == 0)
            hipLaunchKernelGGL((k_pack_limb32<false, true>), dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, check_range, range_flag);
        else hipLaunchKernelGGL(k_pack_limb32<false>, dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, check_range, range_flag);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, st, g, p, (const char*)src, (char*)dst, check_range,
                       range_flag, 0, 0ull, 0);
    return hipGetLastError();
}

hipError_t qg_launch_pack_f64(const QOperandGeom& g, const QPackedGeom& p, const void* src, void* dst, hipStream_t st, int generic)
{
    int64_t blocks = ((p.K_p + 63) / 64) * ((p.rows_p + 63) / 64) * g.parts;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (hipError_t e = zero_trailer(p, dst, st); e != hipSuccess) return e;
    const bool centred_fast = !p.offs || (g.W[0] <= 24 && p.bias > -(1ll << 24) && p.bias < (1ll << 24));
    if (!generic && g.parts == 1 && p.limbs >= 1 && p.limbs <= 3 && !p.digit6 && centred_fast && (p.bk == 64 || p.bk == 128) && p.tr % 64 == 0 && p.rows_p % p.tr == 0 &&
        p.K_p % p.bk == 0 && g.W[0] <= 30 && ((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 15) == 0) {
        const int64_t nblk = (p.K_p / 64) * (p.rows_p / 64);
        const unsigned nb = (unsigned)(nblk < 2048 ? nblk : 2048);
        if (g.rs == 1 && g.ks != 1) hipLaunchKernelGGL((k_pack_limb32<true, false, true>), dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, 0, (int*)nullptr);
        else if (g.ks == 1 && g.rs % 2 == 0 && g.k0 % 2 == 0 && ((uintptr_t)src & 15) == 0)
            hipLaunchKernelGGL((k_pack_limb32<false, true, true>), dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, 0, (int*)nullptr);
        else hipLaunchKernelGGL((k_pack_limb32<false, false, true>), dim3(nb), dim3(256), 0, st, g, p, (const int32_t*)src, (int8_t*)dst, 0, (int*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, st, g, p, (const char*)src, (char*)dst, 0, (int*)nullptr, 2, 0ull, 0);
    return hipGetLastError();
}

hipError_t qg_launch_fill(const QOperandGeom& g, const QPackedGeom& p, uint64_t seed, int dist, void* dst, hipStream_t st)
{
    int64_t blocks = ((p.K_p + 63) / 64) * ((p.rows_p + 63) / 64) * g.parts;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (hipError_t e = zero_trailer(p, dst, st); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, st, g, p, (const char*)nullptr, (char*)dst, 0,
                       (int*)nullptr, 1, seed, dist);
    return hipGetLastError();
}

hipError_t qg_launch_unpack_c(const QCGeom& c, const void* packed, void* dst, hipStream_t st, int generic)
{
    int64_t blocks = ((c.N + 63) / 64) * ((c.M + 63) / 64) * c.parts;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (c.cbytes == 16 || c.sb[0] == 16 || c.sb[1] == 16) {   // wide plans
        const int64_t n = c.M * c.N * c.parts;
        if ((n + 255) / 256 > 0x7fffffffll) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_unpack_c_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, (const char*)packed, (char*)dst);
        return hipGetLastError();
    }
    const bool no_fast = generic != 0;
    if (!no_fast && c.parts == 1 && c.cbytes == 4 && c.elem_bytes == 4 && c.sb[0] == 4 && c.off[0] == 0 && c.tm > 0 && c.tm % 4 == 0 &&
        ((uintptr_t)packed & 15) == 0 && ((uintptr_t)dst & 3) == 0) {
        const int vec = (c.ldc % 4 == 0) && (((uintptr_t)dst & 15) == 0);
        const int64_t nthr = (c.M / 4 + (c.M % 4 != 0)) * c.N;
        if ((nthr + 255) / 256 > 0x7fffffffll) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_unpack_c32, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, c, (const int32_t*)packed, (int32_t*)dst, vec);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_unpack_c, dim3((unsigned)blocks), dim3(256), 0, st, c, (const char*)packed, (char*)dst);
    return hipGetLastError();
}
