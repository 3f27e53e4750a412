// qg_tree64.hip — exact tree-order evaluation for REAL descriptors whose values need more than 32 bits (wide level
// or product formats in the tree class, e.g. QgemulAddArgs<Qu<intBits<29>,fracBits<16>>> behind a rounding product).
//
// Same semantics as k_tree_generic (qg_tree.hip) — every product quantised (Qmul, /root/reference/include/QuBLAS.h:3152-3170),
// every node of the pairwise tree quantised (Qadd inside Reducer::reduce_impl, QuBLAS.h:4960-4984), reference order — with
// the mapping of the 32-bit kernels instead of one output per lane and a scratch-resident counter: each lane owns a 2x2
// block of outputs, k streams through LDS in chunks of 32, the tree is a binary counter whose lower four levels (16
// leaves) are unrolled over compile-time leaf indices and whose upper levels are a statically indexed register array, and
// the wave-uniform mode switches are hoisted around 4-value arrays (qg_step_all.h).  64-bit arithmetic throughout (the
// planner bounds every intermediate by 62 bits).  K is any value with 5..16 tree levels: the packed operands are
// zero-padded to 2^n_levels leaves (a zero right child is the reference's converting copy).
#include <hip/hip_runtime.h>

#include "qg_kernels.h"
#include "qg_step_all.h"

namespace {

constexpr int KC = 32;
constexpr int TMB = 32, TNB = 32;   // 16 x 16 threads x (2 x 2)
constexpr int PITCH = KC + 2;

struct QTree64Args {
    const QTreeTable* tab;
    const char* A;   // [M][K] containers of abytes, K contiguous
    const char* B;   // [N][K]
    char* C;         // [M][N] containers of cbytes
    int64_t M, N, K;
    int32_t abytes, bbytes, cbytes, pad_;
};

__device__ __forceinline__ int64_t ld64(const char* p, int64_t idx, int bytes)
{
    return bytes == 4 ? (int64_t)((const int32_t*)p)[idx] : ((const int64_t*)p)[idx];
}

__device__ __forceinline__ void node4(int64_t (&v)[4], const int64_t (&x)[4], const QTreeTable* __restrict__ t, int l)
{
#pragma unroll
    for (int o = 0; o < 4; ++o) v[o] = x[o] + v[o];   // x = the parked LEFT child, v = the arriving right child
    qg_step_all<int64_t, 4>(v, t->level_add[0][l].q);
}

template <int MAXL>
__global__ __launch_bounds__(256) void k_tree64(QTree64Args g)
{
    __shared__ __attribute__((aligned(16))) int64_t sA[TMB][PITCH];
    __shared__ __attribute__((aligned(16))) int64_t sB[TNB][PITCH];
    const QTreeTable* __restrict__ tab = g.tab;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t tiles_n = (g.N + TNB - 1) / TNB;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * TMB, n0 = (int64_t)(blockIdx.x % tiles_n) * TNB;
    const int nl = tab->n_levels_k;   // (a tree shorter than 5 levels is continued with identity levels: qg_plan.h)

    int64_t low[4][4], up[MAXL - 4][4], v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) v[o] = 0;

    for (int64_t k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();
        // stage 32 rows x 32 k of A and of B: 1024 values each, 4 per thread
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = tid + 256 * c, r = idx >> 5, kk = idx & 31;
            sA[r][kk] = (m0 + r < g.M) ? ld64(g.A, (m0 + r) * g.K + k0 + kk, g.abytes) : 0;
            sB[r][kk] = (n0 + r < g.N) ? ld64(g.B, (n0 + r) * g.K + k0 + kk, g.bbytes) : 0;
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; ++kb) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int64_t a0 = sA[ty * 2][kb * 16 + kk], a1 = sA[ty * 2 + 1][kb * 16 + kk];
                const int64_t b0 = sB[tx * 2][kb * 16 + kk], b1 = sB[tx * 2 + 1][kb * 16 + kk];
                v[0] = a0 * b0; v[1] = a0 * b1; v[2] = a1 * b0; v[3] = a1 * b1;
                qg_step_all<int64_t, 4>(v, tab->mul[0].q);
                // lower four levels: binary counter on the compile-time leaf index
                if ((kk & 1) == 0) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) low[0][o] = v[o];
                } else {
                    node4(v, low[0], tab, 0);
                    if ((kk & 2) == 0) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) low[1][o] = v[o];
                    } else {
                        node4(v, low[1], tab, 1);
                        if ((kk & 4) == 0) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) low[2][o] = v[o];
                        } else {
                            node4(v, low[2], tab, 2);
                            if ((kk & 8) == 0) {
#pragma unroll
                                for (int o = 0; o < 4; ++o) low[3][o] = v[o];
                            } else {
                                node4(v, low[3], tab, 3);
                            }
                        }
                    }
                }
            }
            // v = the partial result of one 16-leaf block (an element of list 4): carry it up
            const unsigned idx = (unsigned)((k0 >> 4) + kb);
            bool parked = false;   // wave-uniform
#pragma unroll
            for (int u = 0; u < MAXL - 4; ++u) {
                if (!parked && 4 + u < nl) {
                    if (((idx >> u) & 1u) == 0) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) up[u][o] = v[o];
                        parked = true;
                    } else {
                        node4(v, up[u], tab, 4 + u);
                    }
                }
            }
        }
    }
    qg_step_all<int64_t, 4>(v, tab->c_cvt[0]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + ty * 2 + i, n = n0 + tx * 2 + j;
            if (m < g.M && n < g.N) {
                const int64_t idx = m * g.N + n, r = v[i * 2 + j];
                switch (g.cbytes) {
                case 1: ((int8_t*)g.C)[idx] = (int8_t)r; break;
                case 2: ((int16_t*)g.C)[idx] = (int16_t)r; break;
                case 4: ((int32_t*)g.C)[idx] = (int32_t)r; break;
                default: ((int64_t*)g.C)[idx] = r; break;
                }
            }
        }
}

} // namespace

hipError_t qg_launch_tree64(const QTreeTable* dev_table, int n_levels, const void* A, const void* B, void* C, int64_t M, int64_t N,
                            int64_t K, int abytes, int bbytes, int cbytes, hipStream_t st)
{
    if (K % KC != 0 || (K & (K - 1)) || n_levels < 5 || n_levels > 16) return hipErrorInvalidValue;
    QTree64Args g{dev_table, (const char*)A, (const char*)B, (char*)C, M, N, K, abytes, bbytes, cbytes, 0};
    const int64_t blocks = ((M + TMB - 1) / TMB) * ((N + TNB - 1) / TNB);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    if (n_levels <= 12) hipLaunchKernelGGL((k_tree64<12>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_tree64<16>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError();
}
