// ref_driver.hpp — TEST INFRASTRUCTURE (build container only; never shipped, never linked
// into the engine).
//
// Own code that #includes the REAL reference header from where it lies
// (/root/reference/include/QuBLAS.h, passed with -I; nothing of it is copied here) and composes
// the GEMM the way SURVEY.md §8-a/§8-c defines it, because `Qgemul` itself is absent from the
// snapshot (readme.md:84-87 only):
//     p[k]   = Qmul<MulTags…>(A'[i,k], B[k,j])          QuBLAS.h:3980-3985
//     s      = Qreduce<Levels…>(p)                       QuBLAS.h:5014-5018 (vector overload)
//     C[i,j] = s                                          converting ctor QuBLAS.h:2398-2411
// For power-of-two K > 512 the tree is evaluated as 512-leaf subtrees followed by a tree over
// the partial sums with the level list shifted by 9 (a perfect binary tree over 2^p leaves is
// a tree of 2^(p-9) such subtrees); the vector overload cannot compile beyond 1000 leaves
// (QuBLAS.h:2694-2697, :2856-2862).
//
// Besides the results, each case prints the fully RESOLVED formats of every node as the
// reference's own types report them (T::intB, T::fracB, T::isS, T::QuM, T::OfM) — that pins the
// descriptor lowering of include/QuBLAS_amd.h / include/qgemul_reference_binding.hpp and qublas_amd/desc.py.
#pragma once
#include "QuBLAS.h"

#include <cstdint>
#include <cstdio>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace refdrv {
using namespace QuBLAS;

// ---- the synthetic generator of oracle/qoracle.c (own code, restated) ----
inline uint64_t qrand(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class T>
int64_t synth(uint64_t seed, int dist, uint64_t elem, int part)
{
    constexpr int W = T::intB + T::fracB;
    int b = dist == 1 ? W / 2 : W;
    int bits = b + (T::isS ? 1 : 0);
    if (bits <= 0) return 0;
    uint64_t r = qrand(seed, elem * 2 + uint64_t(part));
    uint64_t v = bits >= 64 ? r : (r >> (64 - bits));
    uint64_t lo = T::isS ? uint64_t(0) - (uint64_t(1) << b) : 0;   // unsigned arithmetic: well defined for b = 63 too
    return int64_t(lo + v);
}

// ---- format printing ----
template <class T>
std::string fmt_json()
{
    char buf[128];
    std::snprintf(buf, sizeof buf, "[%d,%d,%d,%d,%d]", T::intB, T::fracB, int(T::isS), T::QuM, T::OfM);
    return buf;
}

template <class T>
constexpr bool is_cplx = T::is_complex;

template <class T, bool = is_cplx<T>>
struct parts { using re = T; using im = T; };
template <class T>
struct parts<T, true> { using re = typename T::realType; using im = typename T::imagType; };

template <class T>
std::string fmt2_json()
{
    return "[" + fmt_json<typename parts<T>::re>() + "," + fmt_json<typename parts<T>::im>() + "]";
}

// raw access.  ArbiInt<N <= 64> keeps an int32_t / int64_t in .data; ArbiInt<N > 64> a little-endian std::array<uint64_t, n>
// whose top word carries the sign (QuBLAS.h:566-573, isNegative :635-638).  Values of up to 128 bits travel as __int128.
using raw_t = __int128;
template <class A>
void put_words(A& a, raw_t v)
{
    if constexpr (requires { a.data.size(); }) {
        static_assert(sizeof(a.data) <= 16, "the drivers handle values of at most 128 bits");
        for (size_t i = 0; i < a.data.size(); ++i) a.data[i] = uint64_t(v >> (64 * i));
    } else {
        a.data = decltype(a.data)(v);
    }
}
template <class A>
raw_t get_words(const A& a)
{
    if constexpr (requires { a.data.size(); }) {
        static_assert(sizeof(a.data) <= 16, "the drivers handle values of at most 128 bits");
        unsigned __int128 u = 0;
        for (size_t i = 0; i < a.data.size(); ++i) u |= (unsigned __int128)a.data[i] << (64 * i);
        if (a.data.size() == 1) return raw_t(int64_t(a.data[0]));
        return raw_t(u);
    } else {
        return raw_t(a.data);
    }
}
template <class T>
void set_raw(T& x, raw_t re, raw_t im)
{
    if constexpr (is_cplx<T>) { put_words(x.real.data, re); put_words(x.imag.data, im); }
    else { put_words(x.data, re); (void)im; }
}
template <class T>
void get_raw(const T& x, raw_t& re, raw_t& im)
{
    if constexpr (is_cplx<T>) { re = get_words(x.real.data); im = get_words(x.imag.data); }
    else { re = get_words(x.data); im = 0; }
}
template <class T>
void get_raw(const T& x, int64_t& re, int64_t& im)   // (drivers whose values fit 64 bits)
{
    raw_t r, i;
    get_raw(x, r, i);
    re = int64_t(r);
    im = int64_t(i);
}
// decimal, any magnitude below 2^127 (JSON readers with big integers take it as it is)
inline std::string dec(raw_t v)
{
    if (v == 0) return "0";
    const bool neg = v < 0;
    unsigned __int128 u = neg ? -(unsigned __int128)v : (unsigned __int128)v;
    std::string s;
    while (u) { s.insert(s.begin(), char('0' + int(u % 10))); u /= 10; }
    return neg ? "-" + s : s;
}

// ---- tag lists ----
template <class List> struct apply_mul;
template <class... Tags>
struct apply_mul<TypeList<Tags...>> {
    template <class X, class Y>
    static auto mul(const X& x, const Y& y) { return Qmul<Tags...>(x, y); }
};

template <size_t Q, class List> struct drop;
template <size_t Q> struct drop<Q, TypeList<>> { using type = TypeList<>; };
template <class H, class... T> struct drop<0, TypeList<H, T...>> { using type = TypeList<H, T...>; };
template <size_t Q, class H, class... T>
    requires(Q > 0)
struct drop<Q, TypeList<H, T...>> {
    // levels >= n-1 all use the last type (QuBLAS.h:4913): never drop the last entry
    using type = std::conditional_t<sizeof...(T) == 0, TypeList<H>, typename drop<Q - 1, TypeList<T...>>::type>;
};

template <class List> struct apply_reduce;
template <class... Ls>
struct apply_reduce<TypeList<Ls...>> {
    template <class V>
    static auto reduce(const V& v) { return Qreduce<Ls...>(v); }
};

// level type selection as Reducer::ReducerTypeSelector does (QuBLAS.h:4906-4921, :4966)
template <size_t L, class Prev, class List> struct level_type;
template <size_t L, class Prev> struct level_type<L, Prev, TypeList<>> {
    using tag = std::nullptr_t; // Qadd<nullptr_t>: no tag matches -> default merge
    using buf = Prev;           // buffer of the incoming element type
};
template <size_t L, class Prev, class... Ls> struct level_type<L, Prev, TypeList<Ls...>> {
    using tag = TypeAt<(L >= sizeof...(Ls) ? sizeof...(Ls) - 1 : L), TypeList<Ls...>>;
    using buf = tag;
};

template <size_t L, size_t NL, class Prev, class List>
void print_levels(std::string& add, std::string& lev)
{
    if constexpr (L < NL) {
        using sel = level_type<L, Prev, List>;
        using add_t = decltype(Qadd<typename sel::tag>(std::declval<Prev>(), std::declval<Prev>()));
        using buf_t = typename sel::buf;
        if (L) { add += ","; lev += ","; }
        add += fmt2_json<add_t>();
        lev += fmt2_json<buf_t>();
        print_levels<L + 1, NL, buf_t, List>(add, lev);
    }
}

constexpr size_t ceil_log2(size_t k)
{
    size_t n = 0;
    while (k > 1) { k = (k + 1) / 2; ++n; }
    return n;
}

// ---- resolved sub-op formats of the complex multipliers, obtained from the reference's own
//      member typedefs of Qmul_s<…, BasicComplexMul/TFComplexMul<…>> (QuBLAS.h:3429-3435, :3513-3520)
template <class CA, class CB, class MulList> struct mul_slots;

template <class CA, class CB, class... Tags>
    requires(!is_cplx<CA>)
struct mul_slots<CA, CB, TypeList<Tags...>> {
    static std::string json()
    {
        using p_t = decltype(Qmul<Tags...>(std::declval<CA>(), std::declval<CB>()));
        std::string z = fmt_json<p_t>();
        std::string s = "[" + z;
        for (int i = 1; i < 8; ++i) s += ",[0,0,0,0,0]";
        return s + "]";
    }
    static constexpr int cmul = 0;
};

template <class CA, class CB, class... Args>
    requires(is_cplx<CA>)
struct mul_slots<CA, CB, TypeList<BasicComplexMul<Args...>>> {
    using S = Qmul_s<CA, CB, BasicComplexMul<Args...>>;
    using a_t = typename CA::realType; using b_t = typename CA::imagType;
    using c_t = typename CB::realType; using d_t = typename CB::imagType;
    using ac_t = decltype(Qmul<typename S::mulACType>(std::declval<a_t>(), std::declval<c_t>()));
    using bd_t = decltype(Qmul<typename S::mulBDType>(std::declval<b_t>(), std::declval<d_t>()));
    using ad_t = decltype(Qmul<typename S::mulADType>(std::declval<a_t>(), std::declval<d_t>()));
    using bc_t = decltype(Qmul<typename S::mulBCType>(std::declval<b_t>(), std::declval<c_t>()));
    using re_t = decltype(Qsub<typename S::subACBDType>(std::declval<ac_t>(), std::declval<bd_t>()));
    using im_t = decltype(Qadd<typename S::addADBCType>(std::declval<ad_t>(), std::declval<bc_t>()));
    static std::string json()
    {
        return "[" + fmt_json<ac_t>() + "," + fmt_json<bd_t>() + "," + fmt_json<ad_t>() + "," + fmt_json<bc_t>() + "," +
               fmt_json<re_t>() + "," + fmt_json<im_t>() + ",[0,0,0,0,0],[0,0,0,0,0]]";
    }
    static constexpr int cmul = 1;
};
template <class CA, class CB>
    requires(is_cplx<CA>)
struct mul_slots<CA, CB, TypeList<>> : mul_slots<CA, CB, TypeList<BasicComplexMul<>>> {}; // QuBLAS.h:3422-3424

template <class CA, class CB, class... Args>
    requires(is_cplx<CA>)
struct mul_slots<CA, CB, TypeList<TFComplexMul<Args...>>> {
    using S = Qmul_s<CA, CB, TFComplexMul<Args...>>;
    using a_t = typename CA::realType; using b_t = typename CA::imagType;
    using c_t = typename CB::realType; using d_t = typename CB::imagType;
    using ab_t = decltype(Qadd<typename S::addabType>(std::declval<a_t>(), std::declval<b_t>()));
    using cd_t = decltype(Qadd<typename S::addcdType>(std::declval<c_t>(), std::declval<d_t>()));
    using ba_t = decltype(Qsub<typename S::subbaType>(std::declval<b_t>(), std::declval<a_t>()));
    using A_t = decltype(Qmul<typename S::mulabcType>(std::declval<ab_t>(), std::declval<c_t>()));
    using B_t = decltype(Qmul<typename S::mulbadType>(std::declval<cd_t>(), std::declval<b_t>()));
    using C_t = decltype(Qmul<typename S::mulcdbType>(std::declval<ba_t>(), std::declval<d_t>()));
    using re_t = decltype(Qsub<typename S::subABType>(std::declval<A_t>(), std::declval<B_t>()));
    using im_t = decltype(Qsub<typename S::subBCType>(std::declval<B_t>(), std::declval<C_t>()));
    static std::string json()
    {
        return "[" + fmt_json<ab_t>() + "," + fmt_json<cd_t>() + "," + fmt_json<ba_t>() + "," + fmt_json<A_t>() + "," +
               fmt_json<B_t>() + "," + fmt_json<C_t>() + "," + fmt_json<re_t>() + "," + fmt_json<im_t>() + "]";
    }
    static constexpr int cmul = 2;
};

// ---- explicit or synthetic inputs ----
struct Inputs {
    bool synthetic = true;
    uint64_t seedA = 1, seedB = 2;
    int dist = 0;
    std::vector<int64_t> A, B; // explicit raw values, host linear order, complex interleaved re,im
};

// one dot product through the reference primitives
template <class EA, class EB, class EC, class MulList, class AddList, size_t K>
EC ref_dot(const std::vector<EA>& arow, const std::vector<EB>& bcol)
{
    using prod_t = decltype(apply_mul<MulList>::mul(std::declval<EA>(), std::declval<EB>()));
    constexpr bool pow2 = (K & (K - 1)) == 0;
    if constexpr (K <= 512 || !pow2) {
        static_assert(K <= 1000, "the vector Qreduce cannot compile beyond 1000 leaves");
        Qu_s<dim<K>, prod_t> v;
        for (size_t k = 0; k < K; ++k) v[k] = apply_mul<MulList>::mul(arow[k], bcol[k]);
        EC c = apply_reduce<AddList>::reduce(v);
        return c;
    } else {
        constexpr size_t BK = 512, Q = 9, NB = K / BK;
        using part_t = decltype(apply_reduce<AddList>::reduce(std::declval<Qu_s<dim<BK>, prod_t>>()));
        Qu_s<dim<NB>, part_t> partial;
        Qu_s<dim<BK>, prod_t> v;
        for (size_t b = 0; b < NB; ++b) {
            for (size_t k = 0; k < BK; ++k) v[k] = apply_mul<MulList>::mul(arow[b * BK + k], bcol[b * BK + k]);
            partial[b] = apply_reduce<AddList>::reduce(v);
        }
        EC c = apply_reduce<typename drop<Q, AddList>::type>::reduce(partial);
        return c;
    }
}

template <class EA, class EB, class EC, class MulList, class AddList, bool TA, size_t M, size_t N, size_t K>
void run_case(const char* name, const Inputs& in, FILE* out)
{
    using prod_t = decltype(apply_mul<MulList>::mul(std::declval<EA>(), std::declval<EB>()));
    constexpr size_t NL = ceil_log2(K);
    constexpr bool cx = is_cplx<EA>;
    static_assert(is_cplx<EB> == cx && is_cplx<EC> == cx);

    // host tensors: A dim<M,K> (or dim<K,M> when transposed), B dim<K,N>, column-major
    std::vector<EA> A(M * K);
    std::vector<EB> B(K * N);
    for (size_t e = 0; e < M * K; ++e) {
        int64_t re, im = 0;
        if (in.synthetic) {
            re = synth<typename parts<EA>::re>(in.seedA, in.dist, e, 0);
            if (cx) im = synth<typename parts<EA>::im>(in.seedA, in.dist, e, 1);
        } else {
            re = cx ? in.A[2 * e] : in.A[e];
            if (cx) im = in.A[2 * e + 1];
        }
        set_raw(A[e], re, im);
    }
    for (size_t e = 0; e < K * N; ++e) {
        int64_t re, im = 0;
        if (in.synthetic) {
            re = synth<typename parts<EB>::re>(in.seedB, in.dist, e, 0);
            if (cx) im = synth<typename parts<EB>::im>(in.seedB, in.dist, e, 1);
        } else {
            re = cx ? in.B[2 * e] : in.B[e];
            if (cx) im = in.B[2 * e + 1];
        }
        set_raw(B[e], re, im);
    }

    std::vector<raw_t> C(M * N * (cx ? 2 : 1));
    std::vector<EA> arow(K);
    std::vector<EB> bcol(K);
    for (size_t j = 0; j < N; ++j) {
        for (size_t k = 0; k < K; ++k) bcol[k] = B[k + j * K];
        for (size_t i = 0; i < M; ++i) {
            for (size_t k = 0; k < K; ++k) arow[k] = TA ? A[k + i * K] : A[i + k * M];
            EC c = ref_dot<EA, EB, EC, MulList, AddList, K>(arow, bcol);
            raw_t re, im;
            get_raw(c, re, im);
            if (cx) { C[2 * (i + j * M)] = re; C[2 * (i + j * M) + 1] = im; }
            else C[i + j * M] = re;
        }
    }

    std::string la, lv;
    print_levels<0, NL, prod_t, AddList>(la, lv);
    std::fprintf(out, "{\"name\":\"%s\",\"M\":%zu,\"N\":%zu,\"K\":%zu,\"transA\":%d,\"is_complex\":%d,\"cmul\":%d,\n",
                 name, M, N, K, int(TA), int(cx), mul_slots<EA, EB, MulList>::cmul);
    std::fprintf(out, " \"a\":%s,\"b\":%s,\"c\":%s,\n", fmt2_json<EA>().c_str(), fmt2_json<EB>().c_str(), fmt2_json<EC>().c_str());
    std::fprintf(out, " \"mul\":%s,\"prod\":%s,\"n_levels\":%zu,\n", mul_slots<EA, EB, MulList>::json().c_str(), fmt2_json<prod_t>().c_str(), NL);
    std::fprintf(out, " \"level_add\":[%s],\"level\":[%s],\n", la.c_str(), lv.c_str());
    if (in.synthetic)
        std::fprintf(out, " \"inputs\":{\"seedA\":%llu,\"seedB\":%llu,\"dist\":%d},\n", (unsigned long long)in.seedA,
                     (unsigned long long)in.seedB, in.dist);
    else {
        std::fprintf(out, " \"inputs\":{\"A\":[");
        for (size_t e = 0; e < in.A.size(); ++e) std::fprintf(out, "%s%lld", e ? "," : "", (long long)in.A[e]);
        std::fprintf(out, "],\"B\":[");
        for (size_t e = 0; e < in.B.size(); ++e) std::fprintf(out, "%s%lld", e ? "," : "", (long long)in.B[e]);
        std::fprintf(out, "]},\n");
    }
    std::fprintf(out, " \"C\":[");
    for (size_t e = 0; e < C.size(); ++e) std::fprintf(out, "%s%s", e ? "," : "", dec(C[e]).c_str());
    std::fprintf(out, "]}\n");
}

} // namespace refdrv
