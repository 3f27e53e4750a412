// qg_tree.hip — exact tree-order evaluation on the vector ALUs ("class T").
//
// Every product is quantised (Qmul, /root/reference/include/QuBLAS.h:3152-3170) and every node of the pairwise
// reduction tree is quantised (Qadd in Reducer::reduce_impl, QuBLAS.h:4960-4984) in exactly the
// reference's order.  The tree over K leaves is evaluated as a stream: each lane owns output
// elements and keeps one pending partial per level (a binary counter over k), so the k-loop is
// sequential, needs O(log K) state and reproduces the vector overload's shape for any K
// (odd leftovers are flushed bottom-up with the converting copy, QuBLAS.h:4977-4980).
//
// k_tree_generic is the any-descriptor kernel: real or complex, any per-level formats and modes,
// 64-bit arithmetic — or 128-bit arithmetic for "wide" plans, whose intermediates / level formats / C exceed 62 bits (the
// reference's multi-word ArbiInt<N > 64>, QuBLAS.h:566-912) —, node parameters read from the plan's QTreeTable.  It is the correctness
// backstop; shape-specialised kernels for the hot configurations live beside it.
#include <hip/hip_runtime.h>

#include "qg_kernels.h"

namespace {

__device__ __forceinline__ int64_t ld_c(const char* p, int64_t idx, int cbytes)
{
    return cbytes == 4 ? (int64_t)((const int32_t*)p)[idx] : ((const int64_t*)p)[idx];
}

template <class T>
__device__ __forceinline__ void st_c(char* dst, int64_t idx, int cbytes, T v)
{
    switch (cbytes) {
    case 1: ((int8_t*)dst)[idx] = (int8_t)v; break;
    case 2: ((int16_t*)dst)[idx] = (int16_t)v; break;
    case 4: ((int32_t*)dst)[idx] = (int32_t)v; break;
    case 8: ((int64_t*)dst)[idx] = (int64_t)v; break;
    default: {   // 16: two little-endian words, the upper one carries the sign (the reference's ArbiInt<65..128>, QuBLAS.h:572-573)
        const qg_i128 w = (qg_i128)v;
        ((uint64_t*)dst)[2 * idx] = (uint64_t)(qg_u128)w;
        ((uint64_t*)dst)[2 * idx + 1] = (uint64_t)((qg_u128)w >> 64);
        break;
    }
    }
}

// the arithmetic of one value type: int64_t (every intermediate within 62 bits; unrounded products 64) or the 128-bit type of
// wide plans (qg_ops.h: qg_step_w reproduces the reference's multi-word conversions)
template <class T> struct Ar;
template <> struct Ar<int64_t> {
    static __device__ __forceinline__ int64_t mul(int64_t a, int64_t b, const QNode& n) { return qg_mul<int64_t>(a, b, n); }
    static __device__ __forceinline__ int64_t add(int64_t a, int64_t b, const QNode& n) { return qg_add<int64_t>(a, b, n); }
    static __device__ __forceinline__ int64_t sub(int64_t a, int64_t b, const QNode& n) { return qg_sub<int64_t>(a, b, n); }
    static __device__ __forceinline__ int64_t step(int64_t a, const QStep& s) { return qg_step<int64_t>(a, s); }
};
template <> struct Ar<qg_i128> {
    static __device__ __forceinline__ qg_i128 mul(qg_i128 a, qg_i128 b, const QNode& n) { return qg_mul_w(a, b, n); }
    static __device__ __forceinline__ qg_i128 add(qg_i128 a, qg_i128 b, const QNode& n) { return qg_add_w(a, b, n); }
    static __device__ __forceinline__ qg_i128 sub(qg_i128 a, qg_i128 b, const QNode& n) { return qg_sub_w(a, b, n); }
    static __device__ __forceinline__ qg_i128 step(qg_i128 a, const QStep& s) { return qg_step_w(a, s); }
};

// one (possibly complex) product through the descriptor's sub-operations
template <class T>
__device__ __forceinline__ void product(const QTreeTable* __restrict__ t, const T x[2], const T y[2], T out[2])
{
    using A = Ar<T>;
    if (!t->is_complex) {
        out[0] = A::mul(x[0], y[0], t->mul[QG_MUL_REAL]);
        out[1] = 0;
        return;
    }
    const T a = x[0], b = x[1], c = y[0], d = y[1];
    if (t->cmul == QG_CMUL_TF) {
        T ab = A::add(a, b, t->mul[QG_T_AB]);
        T cd = A::add(c, d, t->mul[QG_T_CD]);
        T ba = A::sub(b, a, t->mul[QG_T_BA]);
        T PA = A::mul(ab, c, t->mul[QG_T_A]);
        T PB = A::mul(cd, b, t->mul[QG_T_B]);
        T PC = A::mul(ba, d, t->mul[QG_T_C]);
        out[0] = A::sub(PA, PB, t->mul[QG_T_RE]);
        out[1] = A::sub(PB, PC, t->mul[QG_T_IM]);
    } else {
        T ac = A::mul(a, c, t->mul[QG_B_AC]);
        T bd = A::mul(b, d, t->mul[QG_B_BD]);
        T ad = A::mul(a, d, t->mul[QG_B_AD]);
        T bc = A::mul(b, c, t->mul[QG_B_BC]);
        out[0] = A::sub(ac, bd, t->mul[QG_B_RE]);
        out[1] = A::add(ad, bc, t->mul[QG_B_IM]);
    }
}

constexpr int TG_T = 16;  // 16x16 outputs per block, one per thread
constexpr int TG_KC = 32; // k-chunk staged in LDS

template <class T>
__global__ __launch_bounds__(256) void k_tree_generic(const QTreeTable* __restrict__ tab, const char* __restrict__ A,
                                                      const char* __restrict__ B, char* __restrict__ C, int64_t M, int64_t N,
                                                      int64_t K, QPackedGeom pa, QPackedGeom pb, QCGeom pc)
{
    using Arith = Ar<T>;
    __shared__ int64_t sA[2][TG_T][TG_KC + 1];   // (operand elements are one-word values in both instantiations)
    __shared__ int64_t sB[2][TG_T][TG_KC + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t tiles_n = (N + TG_T - 1) / TG_T;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * TG_T, n0 = (int64_t)(blockIdx.x % tiles_n) * TG_T;
    const int parts = tab->parts;
    const int nl = tab->n_levels;

    T val[2][QG_MAX_LEVELS + 1];
    uint64_t has = 0; // bit l: a pending element of list l (the same for both parts)

    for (int64_t k0 = 0; k0 < K; k0 += TG_KC) {
        __syncthreads();
        // stage: 16 rows x 32 k for A and B, each part
        for (int idx = threadIdx.x; idx < TG_T * TG_KC; idx += 256) {
            int r = idx / TG_KC, kk = idx % TG_KC;
            int64_t k = k0 + kk;
            for (int p = 0; p < parts; ++p) {
                int64_t va = 0, vb = 0;
                if (k < K) {
                    if (m0 + r < M) va = ld_c(A, ((int64_t)p * pa.rows_p + m0 + r) * pa.K_p + k, pa.cbytes);
                    if (n0 + r < N) vb = ld_c(B, ((int64_t)p * pb.rows_p + n0 + r) * pb.K_p + k, pb.cbytes);
                }
                sA[p][r][kk] = va;
                sB[p][r][kk] = vb;
            }
        }
        __syncthreads();
        const int kend = (K - k0) < TG_KC ? (int)(K - k0) : TG_KC;
        for (int kk = 0; kk < kend; ++kk) {
            T x[2] = {(T)sA[0][ty][kk], parts > 1 ? (T)sA[1][ty][kk] : (T)0};
            T y[2] = {(T)sB[0][tx][kk], parts > 1 ? (T)sB[1][tx][kk] : (T)0};
            T v[2];
            product<T>(tab, x, y, v);
            // push into the binary counter: element of list l meets a pending one -> node of level l
            int l = 0;
            while ((has >> l) & 1) {
                for (int p = 0; p < parts; ++p) {
                    T s = Arith::add(val[p][l], v[p], tab->level_add[p][l]);
                    v[p] = Arith::step(s, tab->level_cvt[p][l]);
                }
                has &= ~(1ull << l);
                ++l;
            }
            for (int p = 0; p < parts; ++p) val[p][l] = v[p];
            has |= 1ull << l;
        }
    }
    // flush odd leftovers bottom-up (only when K is not a power of two)
    for (int l = 0; l < nl; ++l) {
        if (!((has >> l) & 1)) continue;
        T v[2];
        for (int p = 0; p < parts; ++p) v[p] = Arith::step(val[p][l], tab->leftover[p][l]);
        has &= ~(1ull << l);
        int u = l + 1;
        while ((has >> u) & 1) {
            for (int p = 0; p < parts; ++p) {
                T s = Arith::add(val[p][u], v[p], tab->level_add[p][u]);
                v[p] = Arith::step(s, tab->level_cvt[p][u]);
            }
            has &= ~(1ull << u);
            ++u;
        }
        for (int p = 0; p < parts; ++p) val[p][u] = v[p];
        has |= 1ull << u;
    }
    const int64_t m = m0 + ty, n = n0 + tx;
    if (m < M && n < N) {
        for (int p = 0; p < parts; ++p) {
            T r = Arith::step(val[p][nl], tab->c_cvt[p]);
            st_c<T>(C, ((int64_t)p * pc.Mp + m) * pc.Np + n, pc.cbytes, r);
        }
    }
}

} // namespace

hipError_t qg_launch_tree_generic(const QTreeTable* dev_table, int parts, const void* A, const void* B, void* C, int64_t M,
                                  int64_t N, int64_t K, const QPackedGeom& pa, const QPackedGeom& pb, const QCGeom& pc,
                                  hipStream_t st, int wide)
{
    (void)parts;
    int64_t blocks = ((M + TG_T - 1) / TG_T) * ((N + TG_T - 1) / TG_T);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (wide)
        hipLaunchKernelGGL(k_tree_generic<qg_i128>, dim3((unsigned)blocks), dim3(256), 0, st, dev_table, (const char*)A, (const char*)B,
                           (char*)C, M, N, K, pa, pb, pc);
    else
        hipLaunchKernelGGL(k_tree_generic<int64_t>, dim3((unsigned)blocks), dim3(256), 0, st, dev_table, (const char*)A, (const char*)B,
                           (char*)C, M, N, K, pa, pb, pc);
    return hipGetLastError();
}
