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
// 64-bit arithmetic, node parameters read from the plan's QTreeTable.  It is the correctness
// backstop; shape-specialised kernels for the hot configurations live beside it.
#include <hip/hip_runtime.h>

#include "qg_kernels.h"

namespace {

__device__ __forceinline__ int64_t ld_c(const char* p, int64_t idx, int cbytes)
{
    return cbytes == 4 ? (int64_t)((const int32_t*)p)[idx] : ((const int64_t*)p)[idx];
}

__device__ __forceinline__ void st_c(char* dst, int64_t idx, int cbytes, int64_t v)
{
    switch (cbytes) {
    case 1: ((int8_t*)dst)[idx] = (int8_t)v; break;
    case 2: ((int16_t*)dst)[idx] = (int16_t)v; break;
    case 4: ((int32_t*)dst)[idx] = (int32_t)v; break;
    default: ((int64_t*)dst)[idx] = v; break;
    }
}

// one (possibly complex) product through the descriptor's sub-operations
__device__ __forceinline__ void product(const QTreeTable* __restrict__ t, const int64_t x[2], const int64_t y[2], int64_t out[2])
{
    if (!t->is_complex) {
        out[0] = qg_mul<int64_t>(x[0], y[0], t->mul[QG_MUL_REAL]);
        out[1] = 0;
        return;
    }
    const int64_t a = x[0], b = x[1], c = y[0], d = y[1];
    if (t->cmul == QG_CMUL_TF) {
        int64_t ab = qg_add<int64_t>(a, b, t->mul[QG_T_AB]);
        int64_t cd = qg_add<int64_t>(c, d, t->mul[QG_T_CD]);
        int64_t ba = qg_sub<int64_t>(b, a, t->mul[QG_T_BA]);
        int64_t A = qg_mul<int64_t>(ab, c, t->mul[QG_T_A]);
        int64_t B = qg_mul<int64_t>(cd, b, t->mul[QG_T_B]);
        int64_t C = qg_mul<int64_t>(ba, d, t->mul[QG_T_C]);
        out[0] = qg_sub<int64_t>(A, B, t->mul[QG_T_RE]);
        out[1] = qg_sub<int64_t>(B, C, t->mul[QG_T_IM]);
    } else {
        int64_t ac = qg_mul<int64_t>(a, c, t->mul[QG_B_AC]);
        int64_t bd = qg_mul<int64_t>(b, d, t->mul[QG_B_BD]);
        int64_t ad = qg_mul<int64_t>(a, d, t->mul[QG_B_AD]);
        int64_t bc = qg_mul<int64_t>(b, c, t->mul[QG_B_BC]);
        out[0] = qg_sub<int64_t>(ac, bd, t->mul[QG_B_RE]);
        out[1] = qg_add<int64_t>(ad, bc, t->mul[QG_B_IM]);
    }
}

constexpr int TG_T = 16;  // 16x16 outputs per block, one per thread
constexpr int TG_KC = 32; // k-chunk staged in LDS

__global__ __launch_bounds__(256) void k_tree_generic(const QTreeTable* __restrict__ tab, const char* __restrict__ A,
                                                      const char* __restrict__ B, char* __restrict__ C, int64_t M, int64_t N,
                                                      int64_t K, QPackedGeom pa, QPackedGeom pb, QCGeom pc)
{
    __shared__ int64_t sA[2][TG_T][TG_KC + 1];
    __shared__ int64_t sB[2][TG_T][TG_KC + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t tiles_n = (N + TG_T - 1) / TG_T;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * TG_T, n0 = (int64_t)(blockIdx.x % tiles_n) * TG_T;
    const int parts = tab->parts;
    const int nl = tab->n_levels;

    int64_t val[2][QG_MAX_LEVELS + 1];
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
            int64_t x[2] = {sA[0][ty][kk], parts > 1 ? sA[1][ty][kk] : 0};
            int64_t y[2] = {sB[0][tx][kk], parts > 1 ? sB[1][tx][kk] : 0};
            int64_t v[2];
            product(tab, x, y, v);
            // push into the binary counter: element of list l meets a pending one -> node of level l
            int l = 0;
            while ((has >> l) & 1) {
                for (int p = 0; p < parts; ++p) {
                    int64_t s = qg_add<int64_t>(val[p][l], v[p], tab->level_add[p][l]);
                    v[p] = qg_step<int64_t>(s, tab->level_cvt[p][l]);
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
        int64_t v[2];
        for (int p = 0; p < parts; ++p) v[p] = qg_step<int64_t>(val[p][l], tab->leftover[p][l]);
        has &= ~(1ull << l);
        int u = l + 1;
        while ((has >> u) & 1) {
            for (int p = 0; p < parts; ++p) {
                int64_t s = qg_add<int64_t>(val[p][u], v[p], tab->level_add[p][u]);
                v[p] = qg_step<int64_t>(s, tab->level_cvt[p][u]);
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
            int64_t r = qg_step<int64_t>(val[p][nl], tab->c_cvt[p]);
            st_c(C, ((int64_t)p * pc.Mp + m) * pc.Np + n, pc.cbytes, r);
        }
    }
}

} // namespace

hipError_t qg_launch_tree_generic(const QTreeTable* dev_table, int parts, const void* A, const void* B, void* C, int64_t M,
                                  int64_t N, int64_t K, const QPackedGeom& pa, const QPackedGeom& pb, const QCGeom& pc,
                                  hipStream_t st)
{
    (void)parts;
    int64_t blocks = ((M + TG_T - 1) / TG_T) * ((N + TG_T - 1) / TG_T);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tree_generic, dim3((unsigned)blocks), dim3(256), 0, st, dev_table, (const char*)A, (const char*)B,
                       (char*)C, M, N, K, pa, pb, pc);
    return hipGetLastError();
}
