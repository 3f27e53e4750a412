// qg_eltwise.h — device side of the fused element-wise epilogue (include/qgemul.h, qgemul_epilogue).
//
// The reference evaluates  D[i] = cvt_DT( Qop_n<..>( ... Qop_1<..>(C[i], e_1[i]) ..., e_n[i]) )  one element at a
// time through its lazy tensor expressions (/root/reference/include/QuBLAS.h:3780-3877, :4079-4100, tensor
// construction :2732-2746).  Here the chain runs on a small register array of values the GEMM kernel has just
// converted into C's element type; the (wave-uniform) op and mode switches are hoisted out of the per-element loops
// exactly as in qg_step_all.h.  All arithmetic is 64-bit: the planner has bounded every intermediate by 62 bits.
#pragma once
#include "qg_eltwise_args.h"
#include "qg_step_all.h"

// RUN consecutive packed elements starting at element index idx
template <int RUN, class T>
__device__ __forceinline__ void qg_ep_load_run(const char* p, int64_t idx, int ebytes, T* out)
{
    static_assert(RUN == 4, "the MFMA C/D layouts give every lane runs of 4 consecutive rows");
    switch (ebytes) {
    case 1: {
        const uint32_t w = *(const uint32_t*)(p + idx);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = (T)(int8_t)(w >> (8 * r));
        break;
    }
    case 2: {
        const uint2 w = *(const uint2*)(p + idx * 2);
        out[0] = (T)(int16_t)(w.x & 0xffff);
        out[1] = (T)(int16_t)(w.x >> 16);
        out[2] = (T)(int16_t)(w.y & 0xffff);
        out[3] = (T)(int16_t)(w.y >> 16);
        break;
    }
    case 4: {
        const int4 w = *(const int4*)(p + idx * 4);
        out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w;
        break;
    }
    default: {
        const longlong2 a = *(const longlong2*)(p + idx * 8), b = *(const longlong2*)(p + idx * 8 + 16);
        out[0] = (T)a.x; out[1] = (T)a.y; out[2] = (T)b.x; out[3] = (T)b.y;
        break;
    }
    }
}

__device__ __forceinline__ int64_t qg_ep_load_one(const char* p, int64_t idx, int ebytes)
{
    switch (ebytes) {
    case 1: return ((const int8_t*)p)[idx];
    case 2: return ((const int16_t*)p)[idx];
    case 4: return ((const int32_t*)p)[idx];
    default: return ((const int64_t*)p)[idx];
    }
}

// one stage on N values: v = Qop(v, e) or Qop(e, v), rounded / overflowed into the stage's format
template <class T, int N>
__device__ __forceinline__ void qg_ep_stage(T (&v)[N], const T (&e)[N], const QEpStage& s)
{
    if (s.op == QG_EW_PASS) {
        qg_step_all<T, N>(v, s.cvt);
        return;
    }
    if (s.op == QG_EW_MUL) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] *= e[o];
    } else {
        const int sx = s.x_first ? s.node.sa : s.node.sb, se = s.x_first ? s.node.sb : s.node.sa;
        if (s.op == QG_EW_ADD) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = qg_shl<T>(v[o], sx) + qg_shl<T>(e[o], se);
        } else if (s.x_first) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = qg_shl<T>(v[o], sx) - qg_shl<T>(e[o], se);
        } else {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = qg_shl<T>(e[o], se) - qg_shl<T>(v[o], sx);
        }
    }
    qg_step_all<T, N>(v, s.node.q);
    qg_step_all<T, N>(v, s.cvt);
}

// the whole chain on NR runs of 4 consecutive packed elements; run q starts at element index idx0 + q * stride
// T = int32_t when the planner has bounded the whole chain by 32 bits (QEpTable::bits32), else int64_t
template <class T, int NR>
__device__ __forceinline__ void qg_ep_apply_runs(T (&v)[4 * NR], const QEpTable& t, const QEpArgs& a, int64_t idx0, int64_t stride)
{
    for (int k = 0; k < t.n; ++k) {
        T e[4 * NR];
        if (t.st[k].scalar) {
#pragma unroll
            for (int o = 0; o < 4 * NR; ++o) e[o] = (T)a.scalar[k];
        } else {
#pragma unroll
            for (int q = 0; q < NR; ++q) qg_ep_load_run<4, T>(a.e[k], idx0 + q * stride, t.st[k].ebytes, e + 4 * q);
        }
        qg_ep_stage<T, 4 * NR>(v, e, t.st[k]);
    }
    qg_step_all<T, 4 * NR>(v, t.to_d);
}

// RUN consecutive packed elements of `bytes` each, starting at element index idx
template <class T>
__device__ __forceinline__ void qg_ep_store_run(char* D, int64_t idx, int bytes, const T* v)
{
    switch (bytes) {
    case 1:
        *(uint32_t*)(D + idx) = (uint32_t)(v[0] & 0xff) | ((uint32_t)(v[1] & 0xff) << 8) | ((uint32_t)(v[2] & 0xff) << 16) | ((uint32_t)(v[3] & 0xff) << 24);
        break;
    case 2:
        *(uint2*)(D + idx * 2) = make_uint2((uint32_t)(v[0] & 0xffff) | ((uint32_t)(v[1] & 0xffff) << 16), (uint32_t)(v[2] & 0xffff) | ((uint32_t)(v[3] & 0xffff) << 16));
        break;
    case 4:
        *(int4*)(D + idx * 4) = make_int4((int)v[0], (int)v[1], (int)v[2], (int)v[3]);
        break;
    default: {
        int64_t* p = (int64_t*)(D + idx * 8);
        *(longlong2*)p = make_longlong2((int64_t)v[0], (int64_t)v[1]);
        *(longlong2*)(p + 2) = make_longlong2((int64_t)v[2], (int64_t)v[3]);
        break;
    }
    }
}
