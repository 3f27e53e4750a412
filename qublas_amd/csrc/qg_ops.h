// qg_ops.h — quantisation primitives shared by host planning code and gfx950 device code.
//
// The arithmetic follows the reference's converters (fracConvert /root/reference/include/QuBLAS.h:2002-2204,
// intConvert :2227-2334); the engine evaluates them on 32- or 64-bit two's-complement integers,
// which the planner has proven wide enough (qg_plan.cpp: every intermediate < 2^62).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "../../include/qgemul.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QG_HD __host__ __device__ __forceinline__
#else
#define QG_HD inline
#endif

// One quantising step, pre-resolved on the host: value v (frac `from`) -> round by d -> overflow.
// identity != 0: the converting constructor's same-type shortcut (QuBLAS.h:2401-2404).
struct QStep {
    int32_t d;        // from.F - to.F  (<= 0: exact left shift by -d)
    int32_t Q;        // QuMode code of the target
    int32_t O;        // OfMode code of the target
    int32_t W;        // I + F of the target
    int32_t S;        // isSigned of the target
    int32_t identity; // skip entirely
    int64_t lo, hi;   // representable range [S ? -2^W : 0, 2^W - 1] (formats of at most 62 value bits; wide steps derive it from W, S)
    // Wide plans only (qg_step_w below).  refcmp: the value that reaches the overflow handling is a MULTI-WORD ArbiInt in the
    // reference (type width after rounding > 64 bits) while the target's storage fits one word: the reference's comparison
    // with the bounds (operator<=>, QuBLAS.h:1781-1793) then reads the low word as a signed number and the narrowing keeps the
    // low word (:436-441) — reproduced, see qg_overflow_w.
    int32_t refcmp, pad_;
};

// A two-input node: operands are first aligned (left shifts sa, sb), combined, then `q` applies.
struct QNode {
    int32_t sa, sb;   // add/sub: alignment shifts; mul: unused (0)
    QStep q;
};

template <class T>
QG_HD T qg_shl(T x, int s)
{
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
    return (T)((U)x << s);
}

// fracConvert with d > 0 handled by mode; d <= 0 exact shift (QuBLAS.h:2011-2014 et al.)
template <class T>
QG_HD T qg_round(T x, int d, int mode)
{
    if (d <= 0) return qg_shl(x, -d);
    const int bits = (int)sizeof(T) * 8;
    if (d >= bits - 1) { // every kept bit is sign: h = 0 / -1; the engine's planner never emits this for RND
        T h = x < 0 ? (T)-1 : (T)0;
        if (mode == QG_TRN_SMGN) return (T)0;
        return h;
    }
    T h = x >> d;
    T one = 1;
    T l = x & (T)((one << d) - 1);
    T t = one << (d - 1);
    switch (mode) {
    case QG_RND_POS_INF: return h + (T)(l >= t);
    case QG_RND_NEG_INF: return h + (T)(l > t);
    case QG_RND_ZERO: return h + (T)((l > t) | ((l == t) & (x < 0)));
    case QG_RND_INF: return h + (T)((l > t) | ((l == t) & (x > 0)));
    case QG_RND_CONV: return h + (T)((l > t) | ((l == t) & ((h & 1) != 0)));
    case QG_TRN_SMGN: return x < 0 ? (T)(-((T)(-x) >> d)) : h;
    default: return h; // TRN::TCPL
    }
}

// intConvert (QuBLAS.h:2227-2334)
template <class T>
QG_HD T qg_overflow(T x, int O, int W, int S, T lo, T hi)
{
    switch (O) {
    case QG_SAT_TCPL: return x > hi ? hi : (x < lo ? lo : x);
    case QG_SAT_ZERO: return (x > hi || x < lo) ? (T)0 : x;
    case QG_SAT_SMGN: {
        T l2 = S ? (T)(-hi) : (T)0;
        return x > hi ? hi : (x < l2 ? l2 : x);
    }
    case QG_WRP_TCPL: {
        if (S) {
            const int bits = (int)sizeof(T) * 8;
            int sh = bits - (W + 1);
            return (T)(qg_shl(x, sh) >> sh); // keep W+1 low bits, sign-extend
        }
        return x & hi;
    }
    default: return x;
    }
}

template <class T>
QG_HD T qg_step(T x, const QStep& s)
{
    if (s.identity) return x;
    return qg_overflow<T>(qg_round<T>(x, s.d, s.Q), s.O, s.W, s.S, (T)s.lo, (T)s.hi);
}

// ---- 128-bit values (wide plans: intermediates beyond 62 bits; the reference's ArbiInt<N > 64>, QuBLAS.h:566-912) ----
typedef __int128 qg_i128;
typedef unsigned __int128 qg_u128;

QG_HD qg_i128 qg_round_w(qg_i128 x, int d, int mode)   // fracConvert (QuBLAS.h:2002-2204); RND::CONV on a wide value is rejected by the planner
{
    if (d <= 0) return (qg_i128)((qg_u128)x << -d);
    const qg_i128 h = x >> d;
    const qg_i128 one = 1;
    const qg_i128 l = x & ((one << d) - 1);
    const qg_i128 t = one << (d - 1);
    switch (mode) {
    case QG_RND_POS_INF: return h + (qg_i128)(l >= t);
    case QG_RND_NEG_INF: return h + (qg_i128)(l > t);
    case QG_RND_ZERO: return h + (qg_i128)((l > t) | ((l == t) & (x < 0)));
    case QG_RND_INF: return h + (qg_i128)((l > t) | ((l == t) & (x > 0)));
    case QG_RND_CONV: return h + (qg_i128)((l > t) | ((l == t) & ((h & 1) != 0)));
    case QG_TRN_SMGN: return x < 0 ? -((-x) >> d) : h;
    default: return h;
    }
}

// sign of (v <=> bound) as the reference computes it for a multi-word v and a one-word bound (QuBLAS.h:1781-1793)
QG_HD int qg_ref_cmp(qg_i128 v, int64_t bound)
{
    const int64_t hi = (int64_t)(v >> 64), lo = (int64_t)(uint64_t)(qg_u128)v, ext = bound < 0 ? -1 : 0;
    if (hi != ext) return hi < ext ? -1 : 1;
    return lo < bound ? -1 : (lo > bound ? 1 : 0);
}

QG_HD qg_i128 qg_overflow_w(qg_i128 x, const QStep& s)   // intConvert (QuBLAS.h:2227-2334) into (W, S, O)
{
    const qg_i128 one = 1;
    const qg_i128 hi = (one << s.W) - 1, lo = s.S ? -(one << s.W) : (qg_i128)0;
    if (s.refcmp && s.O <= QG_SAT_SMGN) {   // (1 + W <= 64 here: the bounds are one-word values)
        const int64_t h64 = (int64_t)hi, l64 = s.O == QG_SAT_SMGN ? (s.S ? -h64 : 0) : (int64_t)lo;
        if (qg_ref_cmp(x, h64) > 0) return s.O == QG_SAT_ZERO ? (qg_i128)0 : (qg_i128)h64;
        if (qg_ref_cmp(x, l64) < 0) return s.O == QG_SAT_ZERO ? (qg_i128)0 : (qg_i128)l64;
        return 1 + s.W <= 32 ? (qg_i128)(int32_t)(uint32_t)(qg_u128)x : (qg_i128)(int64_t)(uint64_t)(qg_u128)x;   // ArbiInt<M>(val): the low word
    }
    switch (s.O) {
    case QG_SAT_TCPL: return x > hi ? hi : (x < lo ? lo : x);
    case QG_SAT_ZERO: return (x > hi || x < lo) ? (qg_i128)0 : x;
    case QG_SAT_SMGN: {
        const qg_i128 l2 = s.S ? -hi : (qg_i128)0;
        return x > hi ? hi : (x < l2 ? l2 : x);
    }
    case QG_WRP_TCPL: {
        if (s.S) {
            const int sh = 128 - (s.W + 1);
            return (qg_i128)((qg_u128)x << sh) >> sh;
        }
        return x & hi;
    }
    default: return x;
    }
}

QG_HD qg_i128 qg_step_w(qg_i128 x, const QStep& s)
{
    if (s.identity) return x;
    return qg_overflow_w(qg_round_w(x, s.d, s.Q), s);
}
QG_HD qg_i128 qg_mul_w(qg_i128 a, qg_i128 b, const QNode& n) { return qg_step_w(a * b, n.q); }
QG_HD qg_i128 qg_add_w(qg_i128 a, qg_i128 b, const QNode& n) { return qg_step_w((qg_i128)((qg_u128)a << n.sa) + (qg_i128)((qg_u128)b << n.sb), n.q); }
QG_HD qg_i128 qg_sub_w(qg_i128 a, qg_i128 b, const QNode& n) { return qg_step_w((qg_i128)((qg_u128)a << n.sa) - (qg_i128)((qg_u128)b << n.sb), n.q); }

// clamp to [lo, hi] (lo <= hi) in ONE VALU instruction.  hipcc forms v_med3_i32 only for compile-time bounds; with run-time
// bounds it emits v_max_i32 + v_min_i32, and clamps are half of the tree kernels' instruction stream.
QG_HD int qg_clamp_i32(int x, int lo, int hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(lo), "v"(hi));   // in place: no copy when the value lives across a branch
    return x;
#else
    return x < lo ? lo : (x > hi ? hi : x);
#endif
}

template <class T>
QG_HD T qg_mul(T a, T b, const QNode& n)
{
    return qg_step<T>(a * b, n.q);
}
template <class T>
QG_HD T qg_add(T a, T b, const QNode& n)
{
    return qg_step<T>(qg_shl(a, n.sa) + qg_shl(b, n.sb), n.q);
}
template <class T>
QG_HD T qg_sub(T a, T b, const QNode& n)
{
    return qg_step<T>(qg_shl(a, n.sa) - qg_shl(b, n.sb), n.q);
}

// counter-based synthetic generator shared with oracle/qoracle.c (restated there independently)
QG_HD uint64_t qg_rand(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
QG_HD int64_t qg_synth(int W, int S, uint64_t seed, int dist, uint64_t elem, int part)
{
    int b = dist == 1 ? W / 2 : W;
    int bits = b + (S ? 1 : 0);
    if (bits <= 0) return 0;
    uint64_t r = qg_rand(seed, elem * 2 + (uint64_t)part);
    uint64_t v = bits >= 64 ? r : (r >> (64 - bits));
    uint64_t lo = S ? (uint64_t)0 - ((uint64_t)1 << b) : 0;   // unsigned arithmetic: well defined for b = 63 too
    return (int64_t)(lo + v);
}
