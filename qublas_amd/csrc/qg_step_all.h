// qg_step_all.h — round + overflow applied to a small register array with the (wave-uniform)
// mode switches hoisted OUTSIDE the per-element loops, so a block of N values costs N x (2..6)
// VALU instructions plus a handful of scalar branches, instead of a branch ladder per element.
// Arithmetic identical to qg_round / qg_overflow in qg_ops.h (fracConvert / intConvert,
// /root/reference/include/QuBLAS.h:2002-2204, :2227-2334).
#pragma once
#include "qg_ops.h"

template <class T, int N>
__device__ __forceinline__ void qg_round_all(T (&v)[N], int d, int Q)
{
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
    if (d == 0) return;
    if (d < 0) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (T)((U)v[o] << (-d));
        return;
    }
    const T one = 1;
    const T mask = (T)((one << d) - 1), t = (T)(one << (d - 1));
    switch (Q) {
    case QG_TRN_TCPL:
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] >>= d;
        break;
    case QG_TRN_SMGN:
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (v[o] >> d) + (T)((v[o] < 0) & ((v[o] & mask) != 0));
        break;
    case QG_RND_POS_INF:
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (v[o] >> d) + (T)((v[o] & mask) >= t);
        break;
    case QG_RND_NEG_INF:
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (v[o] >> d) + (T)((v[o] & mask) > t);
        break;
    case QG_RND_ZERO:
#pragma unroll
        for (int o = 0; o < N; ++o) { const T l = v[o] & mask; v[o] = (v[o] >> d) + (T)((l > t) | ((l == t) & (v[o] < 0))); }
        break;
    case QG_RND_INF:
#pragma unroll
        for (int o = 0; o < N; ++o) { const T l = v[o] & mask; v[o] = (v[o] >> d) + (T)((l > t) | ((l == t) & (v[o] > 0))); }
        break;
    default: // RND::CONV
#pragma unroll
        for (int o = 0; o < N; ++o) { const T l = v[o] & mask, h = v[o] >> d; v[o] = h + (T)((l > t) | ((l == t) & ((h & 1) != 0))); }
        break;
    }
}

template <class T, int N>
__device__ __forceinline__ void qg_overflow_all(T (&v)[N], const QStep& s)
{
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
    const T lo = (T)s.lo, hi = (T)s.hi;
    switch (s.O) {
    case QG_SAT_TCPL:   // clamp (lo <= hi always): v_med3_i32 for 32-bit values, max-then-min otherwise; no VCC round trip
#pragma unroll
        for (int o = 0; o < N; ++o) {
            if constexpr (sizeof(T) == 4) v[o] = (T)qg_clamp_i32((int)v[o], (int)lo, (int)hi);
            else { const T a = v[o] < lo ? lo : v[o]; v[o] = a > hi ? hi : a; }
        }
        break;
    case QG_SAT_ZERO: {
        const U span = (U)hi - (U)lo;
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = ((U)v[o] - (U)lo > span) ? (T)0 : v[o];
        break;
    }
    case QG_SAT_SMGN: {
        const T l2 = s.S ? (T)(-hi) : (T)0;
#pragma unroll
        for (int o = 0; o < N; ++o) {
            if constexpr (sizeof(T) == 4) v[o] = (T)qg_clamp_i32((int)v[o], (int)l2, (int)hi);
            else { const T a = v[o] < l2 ? l2 : v[o]; v[o] = a > hi ? hi : a; }
        }
        break;
    }
    default: // WRP::TCPL
        if (s.S) {
            const int sh = (int)sizeof(T) * 8 - (s.W + 1);
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (T)((U)v[o] << sh) >> sh;
        } else {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] &= hi;
        }
        break;
    }
}

template <class T, int N>
__device__ __forceinline__ void qg_step_all(T (&v)[N], const QStep& s)
{
    if (s.identity) return;
    qg_round_all<T, N>(v, s.d, s.Q);
    qg_overflow_all<T, N>(v, s);
}
