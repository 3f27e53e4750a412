// qg_fix.h — the compact fixed-mode steps of qg_plan.h (QFix) on the device: shared by the complex and the real 32-bit tree kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "qg_plan.h"

// One scalar load of 8 dwords per step, no branch on the step's mode, alignment / exact left shifts folded into 24-bit
// multiply-adds, the rounding addend into the same instruction, saturation as one v_med3_i32 (lower bound from an SGPR,
// upper bound moved to a VGPR once per step): 2-4 vector instructions per value and step where the table-driven run-time
// forms spend 5-7 and a scalar branch ladder (52 -> 31 instructions per complex MAC of configuration 5; the real kernel's
// run-time-mode form is 4.3x slower than this one at 2048^3).
// v_mad_i32_i24 written out (hipcc keeps a __mul24 and its addend apart): value * scalar factor + value, value * value + scalar
__device__ __forceinline__ int mad24_vsv(int a, int k, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
}
__device__ __forceinline__ int mad24_vvs(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

// LEFT-JUSTIFIED values (a signed SAT::TCPL format of W + 1 bits held as x * 2^s, s = 32 - (W + 1)): the format's range is the
// int32 range, so saturation is the VOP3 clamp bit of the add / subtract / multiply-add itself — tools/ubench/sat_semantics.hip
// checks on the hardware that the clamp acts on the exact result (the product at full width) — and costs no instruction.  A
// positive saturation leaves 2^31 - 1, whose low s bits are ones where hi * 2^s has zeros: read as floor(v / 2^s) the value is
// right, and stays right through ONE further saturating add (ones + zeros never carry; two saturated values saturate again);
// a second add could carry, and a subtrahend must be clean, so the low bits are cleared (v_and) after every second add and
// after every product (whose low bits are the fraction the rounding drops: flooring them IS the rounding's shift).
__device__ __forceinline__ int sat_add(int a, int b)
{
    int r;
    asm("v_add_i32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int sat_sub(int a, int b)
{
    int r;
    asm("v_sub_i32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int sat_mad24_vsv(int a, int b, int c)   // (b wave-uniform)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int sat_mad24_vvs(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

// UNSIGNED formats (all operands unsigned: nothing ever goes below 0): x * 2^(32 - bits) fills the uint32 range, and the clamp bit of
// v_mad_u32_u24 / v_add_u32 (v_pk_mad_u16 / v_pk_add_u16) saturates at 2^32 - 1 (2^16 - 1) — the same ones-below-the-unit rule
__device__ __forceinline__ int usat_add(int a, int b)
{
    int r;
    asm("v_add_u32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int usat_mad24_vvs(int a, int b, int c)
{
    int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
// ... and in packed 16-bit halves (x * 2^(16 - width), two values per register).  The multiply-add's first operand is one half
// (HALF) of its register for BOTH results: an A element against a pair of B columns.
template <int HALF>
__device__ __forceinline__ int pk_mad_sat(int a2, int b2, int t2)
{
    int r;
    if (HALF == 0) asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(r) : "v"(a2), "v"(b2), "s"(t2));
    else asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "=v"(r) : "v"(a2), "v"(b2), "s"(t2));
    return r;
}
__device__ __forceinline__ int pk_add_sat(int a, int b)
{
    int r;
    asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int HALF>
__device__ __forceinline__ int pk_mad_usat(int a2, int b2, int t2)
{
    int r;
    if (HALF == 0) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(r) : "v"(a2), "v"(b2), "s"(t2));
    else asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "=v"(r) : "v"(a2), "v"(b2), "s"(t2));
    return r;
}
__device__ __forceinline__ int pk_add_usat(int a, int b)
{
    int r;
    asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk_sub_sat(int a, int b)
{
    int r;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk2(int lo, int hi) { return (int)(((unsigned)lo & 0xffffu) | ((unsigned)hi << 16)); }


template <int N>
__device__ __forceinline__ void fx_finish(int (&v)[N], const QFix& f)
{
    if (f.d) {   // (wave-uniform; sums of equally aligned values have d == 0)
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] >>= f.d;
    }
    const int hi = f.hi;
#pragma unroll
    for (int o = 0; o < N; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(v[o]) : "s"(f.lo), "v"(hi));
}
// BIASED records (the real kernel's fast_mode 4, qg_plan.cpp): the value is u = v - lo >= 0 of its own format; after the shift
// the overflow kind of the record (QFix::kb) acts on u with span = hi - lo (QFix::hi) and the biased zero B = -lo (QFix::lo):
// 0 clamp = med3(u, 0, span); 1 SAT::ZERO = one unsigned compare + select; 2 WRP::TCPL = u & span; 4 none.  Wave-uniform.
template <int N>
__device__ __forceinline__ void fx_finish_biased(int (&v)[N], const QFix& f)
{
    if (f.d) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] >>= f.d;
    }
    if (f.kb == 1) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = ((unsigned)v[o] > (unsigned)f.hi) ? f.lo : v[o];
    } else if (f.kb == 0) {
#pragma unroll
        for (int o = 0; o < N; ++o) asm("v_med3_i32 %0, %0, 0, %1" : "+v"(v[o]) : "s"(f.hi));
    } else if (f.kb == 2) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] &= f.hi;
    }
}
// UNBIASED records with an overflow kind (the real kernel's fast_mode 5: formats too wide for the biased form), QFix::kb: 0 one clamp (SAT::TCPL, SAT::SMGN), 1 out of range -> 0 (SAT::ZERO), 2 wrap, signed (WRP::TCPL: keep the low
// W + 1 bits, sign-extended; hi = 2^W - 1), 3 wrap, unsigned (v & hi).  Wave-uniform.
// ... and a rounding kind (QFix::skip of such a record) for the modes whose addend depends on the value: 0 the constant
// addend is already in (TRN::TCPL, RND::POS_INF, RND::NEG_INF); else the record's t is 0 and, with half = 2^(d-1) and c = QFix::ka,
//   1 RND::ZERO   (v + (half - 1) + [v < 0]) >> d          c = half - 1
//   2 RND::INF    (v + half - [v < 0]) >> d                c = half
//   3 RND::CONV   (v + (half - 1) + bit d of v) >> d       c = half - 1      (bit d of v = the parity of the truncated result)
//   4 TRN::SMGN   (v + ([v < 0] ? 2^d - 1 : 0)) >> d       c = 2^d - 1
// (each equals the reference's fracConvert for that mode: qg_round in qg_ops.h; checked against it by the fuzz suites).
template <int N>
__device__ __forceinline__ void fx_finish_any(int (&v)[N], const QFix& f)
{
    if (f.d) {
        if (f.skip == 0) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] >>= f.d;
        } else if (f.skip == 1) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + f.ka + (int)((unsigned)v[o] >> 31)) >> f.d;
        } else if (f.skip == 2) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + f.ka + (v[o] >> 31)) >> f.d;
        } else if (f.skip == 3) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + f.ka + ((v[o] >> f.d) & 1)) >> f.d;
        } else {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + ((v[o] >> 31) & f.ka)) >> f.d;
        }
    }
    if (f.kb == 0) {
        const int hi = f.hi;
#pragma unroll
        for (int o = 0; o < N; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(v[o]) : "s"(f.lo), "v"(hi));
    } else if (f.kb == 1) {
        const unsigned span = (unsigned)f.hi - (unsigned)f.lo;
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = ((unsigned)v[o] - (unsigned)f.lo > span) ? 0 : v[o];
    } else if (f.kb == 2) {
        const int sh = __builtin_clz((unsigned)f.hi) - 1;   // 31 - (W + 1) for hi = 2^W - 1
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (int)((unsigned)v[o] << sh) >> sh;
    } else {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] &= f.hi;
    }
}
// COMPLEX records carry both factors, so their kinds are packed into QFix::skip: bit 0 identity, bits 8..15 the rounding kind
// (as above), bits 16.. the overflow kind (0 clamp, 1 SAT::ZERO, 2 / 3 WRP::TCPL signed / unsigned); the rounding kind's
// constant follows from d.  Unbiased values.
template <int N>
__device__ __forceinline__ void fx_finish_packed(int (&v)[N], const QFix& f)
{
    const int rk = (f.skip >> 8) & 0xff, ok = f.skip >> 16;
    if (f.d) {
        const int half = 1 << (f.d - 1);
        if (rk == 0) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] >>= f.d;
        } else if (rk == 1) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + (half - 1) + (int)((unsigned)v[o] >> 31)) >> f.d;
        } else if (rk == 2) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + half + (v[o] >> 31)) >> f.d;
        } else if (rk == 3) {
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + (half - 1) + ((v[o] >> f.d) & 1)) >> f.d;
        } else {
            const int mask = (half << 1) - 1;
#pragma unroll
            for (int o = 0; o < N; ++o) v[o] = (v[o] + ((v[o] >> 31) & mask)) >> f.d;
        }
    }
    if (ok == 0) {
        const int hi = f.hi;
#pragma unroll
        for (int o = 0; o < N; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(v[o]) : "s"(f.lo), "v"(hi));
    } else if (ok == 1) {
        const unsigned span = (unsigned)f.hi - (unsigned)f.lo;
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = ((unsigned)v[o] - (unsigned)f.lo > span) ? 0 : v[o];
    } else if (ok == 2) {
        const int sh = __builtin_clz((unsigned)f.hi) - 1;
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] = (int)((unsigned)v[o] << sh) >> sh;
    } else {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] &= f.hi;
    }
}
// ... and the same WITHOUT a branch on the kind, for descriptors whose kinds are all among FEAT's (the scalar branch ladder of
// fx_finish_packed costs more than the vector work it selects): bit 0 (R) roundings that add one bit of the value —
// RND::ZERO (+ the sign bit), RND::CONV (+ bit d), TRN::SMGN (+ sign * (2^d - 1)) — as v_bfe_u32 + v_mad_i32_i24 with the
// bit's offset in skip[12:8] and the factor k (0: none) handed in; the bit is read AFTER the step's constant t = 2^(d-1) - 1
// was added, which moves it only where the result does not depend on it.  RND::INF adds the INVERTED sign bit: its constant
// carries 2^31 as well, and the shifted value is repaired by W's sign-extraction of the low 31 - d bits.  Bit 1 (Z) SAT::ZERO next to clamps: w = med3(v),
// v = (w == v) ? v : (w & cm), cm = -1 clamp / 0 zero in skip[24].  Bit 2 (W) WRP::TCPL of signed formats: v_bfe_i32 of the low
// skip[21:16] bits (31 = no wrap: every value of the 32-bit kernels fits 31 bits).
template <int FEAT, int N>
__device__ __forceinline__ void fx_finish_feat(int (&v)[N], const QFix& f, int k)
{
    if (FEAT & 1) {
        const int off = (f.skip >> 8) & 31;
#pragma unroll
        for (int o = 0; o < N; ++o) {
            int c;
            asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(c) : "v"(v[o]), "s"(off));
            v[o] = mad24_vsv(c, k, v[o]);
        }
    }
    if (f.d) {
#pragma unroll
        for (int o = 0; o < N; ++o) v[o] >>= f.d;
    }
    if (FEAT & 4) {
        const int wd = (f.skip >> 16) & 63;
#pragma unroll
        for (int o = 0; o < N; ++o) asm("v_bfe_i32 %0, %0, 0, %1" : "+v"(v[o]) : "s"(wd));
    }
    const int hi = f.hi;
    if (FEAT & 2) {
        const int cm = -((f.skip >> 24) & 1);
#pragma unroll
        for (int o = 0; o < N; ++o) {
            int w = v[o];
            asm("v_med3_i32 %0, %0, %1, %2" : "+v"(w) : "s"(f.lo), "v"(hi));
            v[o] = (w == v[o]) ? w : (w & cm);
        }
    } else {
#pragma unroll
        for (int o = 0; o < N; ++o) asm("v_med3_i32 %0, %0, %1, %2" : "+v"(v[o]) : "s"(f.lo), "v"(hi));
    }
}
// the rounding addend in a VGPR, so that it can ride in a multiply-add next to a scalar factor (one scalar operand per instruction)
__device__ __forceinline__ int fx_vgpr(int s)
{
    int v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
// The steps' parameters are loop-invariant; left alone, the compiler hoists all of their scalar loads out of the k loop
// (16 steps x 8 dwords), runs out of SGPRs and spills them through VGPR lanes.  The load is therefore written out: one
// s_load_dwordx8 from the scalar cache right where the step runs.
typedef int fx_v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ QFix fx_at(const QTreeTable* t, unsigned byte_off)
{
    fx_v8i r;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(t), "s"(byte_off));
    QFix f;
    f.ka = r[0]; f.kb = r[1]; f.t = r[2]; f.d = r[3]; f.lo = r[4]; f.hi = r[5]; f.skip = r[6]; f.ls = r[7];
    return f;
}
// two / three records whose steps follow each other: the loads go out back to back and share ONE wait (a step's first vector
// instruction needs its record, so every separate load exposes a scalar-cache round trip that only other waves can cover)
__device__ __forceinline__ QFix fx_of(const fx_v8i& r)
{
    QFix f;
    f.ka = r[0]; f.kb = r[1]; f.t = r[2]; f.d = r[3]; f.lo = r[4]; f.hi = r[5]; f.skip = r[6]; f.ls = r[7];
    return f;
}
__device__ __forceinline__ void fx_at2(const QTreeTable* t, unsigned o0, unsigned o1, QFix& f0, QFix& f1)
{
    fx_v8i r0, r1;
    asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(t), "s"(o0), "s"(o1));
    f0 = fx_of(r0);
    f1 = fx_of(r1);
}
__device__ __forceinline__ void fx_at3(const QTreeTable* t, unsigned o0, unsigned o1, unsigned o2, QFix& f0, QFix& f1, QFix& f2)
{
    fx_v8i r0, r1, r2;
    asm volatile("s_load_dwordx8 %0, %3, %4\n\ts_load_dwordx8 %1, %3, %5\n\ts_load_dwordx8 %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2)
                 : "s"(t), "s"(o0), "s"(o1), "s"(o2));
    f0 = fx_of(r0);
    f1 = fx_of(r1);
    f2 = fx_of(r2);
}
__device__ __forceinline__ void fx_at4(const QTreeTable* t, unsigned o0, unsigned o1, unsigned o2, unsigned o3, QFix& f0, QFix& f1, QFix& f2, QFix& f3)
{
    fx_v8i r0, r1, r2, r3;
    asm volatile("s_load_dwordx8 %0, %4, %5\n\ts_load_dwordx8 %1, %4, %6\n\ts_load_dwordx8 %2, %4, %7\n\ts_load_dwordx8 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
                 : "s"(t), "s"(o0), "s"(o1), "s"(o2), "s"(o3));
    f0 = fx_of(r0);
    f1 = fx_of(r1);
    f2 = fx_of(r2);
    f3 = fx_of(r3);
}
__device__ __forceinline__ void fx_at5(const QTreeTable* t, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned o4, QFix& f0, QFix& f1, QFix& f2, QFix& f3,
                                       QFix& f4)
{
    fx_v8i r0, r1, r2, r3, r4;
    asm volatile("s_load_dwordx8 %0, %5, %6\n\ts_load_dwordx8 %1, %5, %7\n\ts_load_dwordx8 %2, %5, %8\n\ts_load_dwordx8 %3, %5, %9\n\ts_load_dwordx8 %4, %5, %10\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4)
                 : "s"(t), "s"(o0), "s"(o1), "s"(o2), "s"(o3), "s"(o4));
    f0 = fx_of(r0);
    f1 = fx_of(r1);
    f2 = fx_of(r2);
    f3 = fx_of(r3);
    f4 = fx_of(r4);
}
#define FX_OFF_MUL(slot) ((unsigned)(offsetof(QTreeTable, fmul) + (slot) * sizeof(QFix)))
#define FX_OFF_ADD(part, l) ((unsigned)(offsetof(QTreeTable, fadd) + ((part) * QG_MAX_LEVELS + (l)) * sizeof(QFix)))
#define FX_OFF_CVT(part, l) ((unsigned)(offsetof(QTreeTable, fcvt) + ((part) * QG_MAX_LEVELS + (l)) * sizeof(QFix)))

