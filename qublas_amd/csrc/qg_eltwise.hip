// qg_eltwise.hip — element-wise epilogue outside the MFMA kernels (gfx950).
//
//  * k_eltwise: the qgemul_epilogue chain as a pass over packed C -> packed D, for the kernels that do not fuse it
//    (the exact tree kernels).  C, the tensor operands and D share ONE index space (the plan's packed-C layout),
//    so the pass is linear in memory: every lane handles 16 consecutive elements.  HBM-bound:
//    (cbytes + sum ebytes + dbytes) bytes per element.
//  * k_pack_e: a tensor operand in reference layout (column-major M x N, QuBLAS.h:2680-2692) -> that index space
//    (one launch per part of a complex operand).
#include <hip/hip_runtime.h>

#include "qg_eltwise.h"
#include "qg_kernels.h"

namespace {

__device__ __forceinline__ void store_one(char* dst, int64_t idx, int bytes, int64_t v)
{
    switch (bytes) {
    case 1: ((int8_t*)dst)[idx] = (int8_t)v; break;
    case 2: ((int16_t*)dst)[idx] = (int16_t)v; break;
    case 4: ((int32_t*)dst)[idx] = (int32_t)v; break;
    default: ((int64_t*)dst)[idx] = v; break;
    }
}

__global__ __launch_bounds__(256) void k_eltwise(QEltwiseArgs g)
{
    // 16 consecutive elements per lane (4 runs of 4): enough independent loads in flight per lane for 1-byte containers too
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= g.n) return;
    if (i0 + 16 <= g.n) {
        if (g.t.bits32) {
            int32_t v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) qg_ep_load_run<4, int32_t>(g.C, i0 + 4 * q, g.cbytes, v + 4 * q);
            qg_ep_apply_runs<int32_t, 4>(v, g.t, g.a, i0, 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) qg_ep_store_run<int32_t>(g.D, i0 + 4 * q, g.t.dbytes, v + 4 * q);
        } else {
            int64_t v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) qg_ep_load_run<4, int64_t>(g.C, i0 + 4 * q, g.cbytes, v + 4 * q);
            qg_ep_apply_runs<int64_t, 4>(v, g.t, g.a, i0, 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) qg_ep_store_run<int64_t>(g.D, i0 + 4 * q, g.t.dbytes, v + 4 * q);
        }
        return;
    }
    for (int64_t i = i0; i < g.n; ++i) {   // tail of a packed C whose size is not a multiple of 16
        int64_t v[1] = {qg_ep_load_one(g.C, i, g.cbytes)};
        for (int k = 0; k < g.t.n; ++k) {
            int64_t e[1] = {g.t.st[k].scalar ? g.a.scalar[k] : qg_ep_load_one(g.a.e[k], i, g.t.st[k].ebytes)};
            qg_ep_stage<int64_t, 1>(v, e, g.t.st[k]);
        }
        qg_step_all<int64_t, 1>(v, g.t.to_d);
        store_one(g.D, i, g.t.dbytes, v[0]);
    }
}

// host column-major (m fastest) -> packed-C index space; 64 x 64 tile through LDS so that both sides are coalesced.
// A host element is `stride` bytes; this part's raw value sits `off` bytes into it (complex operands: {re, im} structs).
__global__ __launch_bounds__(256) void k_pack_e(QCGeom c, int part, const char* __restrict__ src, int64_t ld, int stride, int off, int src_bytes,
                                                char* __restrict__ dst, int ebytes)
{
    __shared__ int64_t tile[64][65];
    const int64_t nt = (c.N + 63) / 64;
    const int64_t tn = blockIdx.x % nt, tm = blockIdx.x / nt;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t m = tm * 64 + tx, n = tn * 64 + i;
        int64_t v = 0;
        if (m < c.M && n < c.N) {
            const char* q = src + (m + n * ld) * stride + off;
            v = src_bytes == 4 ? (int64_t) * (const int32_t*)q : *(const int64_t*)q;
        }
        tile[tx][i] = v;   // tile[m_local][n_local]
    }
    __syncthreads();
    const bool m_fast = c.tm != 0;   // tiled layout: rows contiguous inside a tile column; row-major otherwise
    for (int i = ty; i < 64; i += 4) {
        const int ml = m_fast ? tx : i, nl = m_fast ? i : tx;
        const int64_t m = tm * 64 + ml, n = tn * 64 + nl;
        if (m < c.M && n < c.N) store_one(dst, qg_c_index(c, part, m, n), ebytes, tile[ml][nl]);
    }
}

} // namespace

hipError_t qg_launch_eltwise(const QEltwiseArgs& g, hipStream_t st)
{
    if (g.n <= 0) return hipSuccess;
    const int64_t blocks = (g.n + 4095) / 4096;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_eltwise, dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError();
}

hipError_t qg_launch_pack_e(const QCGeom& c, int part, const void* src, int64_t ld, int stride, int off, int src_bytes, void* dst, int ebytes,
                            hipStream_t st)
{
    const int64_t blocks = ((c.N + 63) / 64) * ((c.M + 63) / 64);
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    // padding = 0 (this part's region of the packed tensor)
    if (hipError_t e = hipMemsetAsync((char*)dst + (size_t)(part * c.Mp * c.Np) * ebytes, 0, (size_t)(c.Mp * c.Np) * ebytes, st); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pack_e, dim3((unsigned)blocks), dim3(256), 0, st, c, part, (const char*)src, ld, stride, off, src_bytes, (char*)dst, ebytes);
    return hipGetLastError();
}
