// QuBLAS_amd.h — standalone C++23 host header for the MI355X fixed-point GEMM engine.
//
// It keeps the spelling of the reference library's tag API for the Qgemul path
//   Qu<intBits<>, fracBits<>, isSigned<>, QuMode<RND::…|TRN::…>, OfMode<SAT::…|WRP::…>>,
//   Qu<dim<…>, T>, Qcomplex<R, I>, TypeList<…>, FullPrec,
//   BasicComplexMul<acT<>,bdT<>,adT<>,bcT<>,acbdT<>,adbcT<>>, TFComplexMul<abT<>,…,BCT<>>,
//   Qgemul<QgemulAddArgs<…>, QgemulMulArgs<…>, QgemulTransposedA<…>>(C, A, B)
// (reference: /root/reference/readme.md:22-87, tags /root/reference/include/QuBLAS.h:1986-1999, :2209-2225, :2346-2359)
// but is written from scratch around a different idea: formats are constexpr VALUES (struct Fmt),
// tag packs are parsed once into a constexpr TagSet, and the merger rules (QuBLAS.h:3107-3139)
// are ordinary constexpr functions.  K, M, N are runtime fields of the descriptor, so compile
// time does not grow with the reduction length (the reference's Reducer instantiates a template
// per level and per element, QuBLAS.h:4960-4984).
//
// The header does no fixed-point arithmetic beyond constructing values from doubles; Qgemul
// lowers to a qgemul_desc (include/qgemul.h) and calls libqugemm.so, which runs on gfx950 only.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "qgemul.h"

namespace QuBLAS_amd {

// ------------------------------------------------------------------ tags
template <typename... Ts> struct TypeList { static constexpr size_t size = sizeof...(Ts); };

struct RND {
    struct POS_INF { static constexpr int value = QG_RND_POS_INF; };
    struct NEG_INF { static constexpr int value = QG_RND_NEG_INF; };
    struct ZERO { static constexpr int value = QG_RND_ZERO; };
    struct INF { static constexpr int value = QG_RND_INF; };
    struct CONV { static constexpr int value = QG_RND_CONV; };
};
struct TRN {
    struct TCPL { static constexpr int value = QG_TRN_TCPL; };
    struct SMGN { static constexpr int value = QG_TRN_SMGN; };
};
struct SAT {
    struct TCPL { static constexpr int value = QG_SAT_TCPL; };
    struct ZERO { static constexpr int value = QG_SAT_ZERO; };
    struct SMGN { static constexpr int value = QG_SAT_SMGN; };
};
struct WRP {
    struct TCPL { static constexpr int value = QG_WRP_TCPL; };
    template <auto N> struct TCPL_SAT { static constexpr int value = QG_WRP_TCPL_SAT; };
};

template <int V> struct intBits {};
template <int V> struct fracBits {};
template <bool V> struct isSigned {};
template <typename M> struct QuMode {};
template <typename M> struct OfMode {};
struct FullPrec {};
template <size_t... D> struct dim {
    static constexpr size_t rank = sizeof...(D);
    static constexpr size_t elems = (D * ... * 1);
    static constexpr std::array<size_t, sizeof...(D)> extent = {D...};
};

// ------------------------------------------------------------------ formats as values
struct Fmt {
    int I = 8, F = 8;       // defaults of the reference, QuBLAS.h:2355-2359
    bool S = true;
    int Q = QG_TRN_TCPL, O = QG_SAT_TCPL;
    constexpr bool operator==(const Fmt&) const = default;
    constexpr qfmt c() const { return qfmt{int16_t(I), int16_t(F), uint8_t(S), uint8_t(Q), uint8_t(O), 0}; }
};

// a parsed tag pack: which of the five tags (and FullPrec) are present.  The FIRST occurrence of
// a tag wins, as with the reference's tagExtractor (QuBLAS.h:144-148, :187-190).
struct TagSet {
    bool hasI = false, hasF = false, hasS = false, hasQ = false, hasO = false, full = false;
    int I = 0, F = 0, Q = 0, O = 0;
    bool S = true;
};

template <typename... Args> class Qu_s;  // scalar, complex or tensor; specialisations below

namespace detail {

template <class T> struct tag_apply { static constexpr void go(TagSet&) {} };  // unknown tags are ignored
template <int V> struct tag_apply<intBits<V>> { static constexpr void go(TagSet& t) { if (!t.hasI) { t.hasI = true; t.I = V; } } };
template <int V> struct tag_apply<fracBits<V>> { static constexpr void go(TagSet& t) { if (!t.hasF) { t.hasF = true; t.F = V; } } };
template <bool V> struct tag_apply<isSigned<V>> { static constexpr void go(TagSet& t) { if (!t.hasS) { t.hasS = true; t.S = V; } } };
template <class M> struct tag_apply<QuMode<M>> { static constexpr void go(TagSet& t) { if (!t.hasQ) { t.hasQ = true; t.Q = M::value; } } };
template <class M> struct tag_apply<OfMode<M>> { static constexpr void go(TagSet& t) { if (!t.hasO) { t.hasO = true; t.O = M::value; } } };
template <> struct tag_apply<FullPrec> { static constexpr void go(TagSet& t) { t.full = true; } };

// parse<Tags…>: loose tags; a TypeList is opened; a full scalar Qu type counts as its five tags
// ONLY when it is the sole argument (MergerArgsWrapper_s single-argument specialisation,
// QuBLAS.h:3097-3099) — next to other arguments the reference leaves it wrapped and no tag
// extractor matches it, so it is ignored here too.
template <typename... Ts> struct parse {
    static constexpr TagSet value = [] { TagSet t; (tag_apply<Ts>::go(t), ...); return t; }();
};
template <typename... Ts> struct parse<TypeList<Ts...>> : parse<Ts...> {};
template <int I, int F, bool S, class Q, class O>
struct parse<Qu_s<intBits<I>, fracBits<F>, isSigned<S>, QuMode<Q>, OfMode<O>>> {
    static constexpr TagSet value = TagSet{true, true, true, true, true, false, I, F, Q::value, O::value, S};
};

constexpr int imax(int a, int b) { return a > b ? a : b; }

// MulMerger / AddMerger (QuBLAS.h:3107-3120, :3125-3139)
constexpr Fmt merge_mul(Fmt a, Fmt b, TagSet t)
{
    return Fmt{t.hasI ? t.I : (t.full ? a.I + b.I : imax(a.I, b.I)), t.hasF ? t.F : (t.full ? a.F + b.F : imax(a.F, b.F)),
               t.hasS ? t.S : (a.S || b.S), t.hasQ ? t.Q : (a.Q == b.Q ? a.Q : int(QG_TRN_TCPL)),
               t.hasO ? t.O : (a.O == b.O ? a.O : int(QG_SAT_TCPL))};
}
constexpr Fmt merge_add(Fmt a, Fmt b, TagSet t)
{
    return Fmt{t.hasI ? t.I : (t.full ? imax(a.I, b.I) + 1 : imax(a.I, b.I)), t.hasF ? t.F : imax(a.F, b.F),
               t.hasS ? t.S : (a.S || b.S), t.hasQ ? t.Q : (a.Q == b.Q ? a.Q : int(QG_TRN_TCPL)),
               t.hasO ? t.O : (a.O == b.O ? a.O : int(QG_SAT_TCPL))};
}

// double -> raw with the format's own rounding and overflow (Qu_s(double), QuBLAS.h:2387-2393).
// Exact: v = m * 2^e with a 53-bit integer m, so the raw value at F fraction bits is m * 2^(e+F) — a left
// shift, or ONE rounding of the integer m by d = -(e+F) bits with the type's QuMode; then its OfMode.
// (The reference loads the double into a 2400-bit buffer first; zero, NaN and infinity give 0, :670-674.)
inline int64_t from_double(double v, Fmt f)
{
    if (v == 0.0 || std::isnan(v) || std::isinf(v)) return 0;
    int ex = 0;
    const double fr = std::frexp(v, &ex);                 // v = fr * 2^ex, 0.5 <= |fr| < 1
    __int128 m = (__int128)std::ldexp(fr, 53);            // exact 53-bit signed integer
    int e = ex - 53 + f.F;                                // v * 2^F = m * 2^e
    __int128 r;
    if (e >= 0) {
        r = m * ((__int128)1 << (e > 60 ? 60 : e));       // far outside every supported format when clipped
    } else {
        int d = -e;
        if (d > 120) { m = m < 0 ? -1 : 1; d = 8; }       // all mantissa bits below the rounding position
        const __int128 h = m >> d, l = m & (((__int128)1 << d) - 1), t = (__int128)1 << (d - 1);
        r = h;
        switch (f.Q) {
        case QG_RND_POS_INF: r = h + (l >= t); break;
        case QG_RND_NEG_INF: r = h + (l > t); break;
        case QG_RND_ZERO: r = h + (l > t || (l == t && m < 0)); break;
        case QG_RND_INF: r = h + (l > t || (l == t && m > 0)); break;
        case QG_RND_CONV: r = h + (l > t || (l == t && (h & 1))); break;
        case QG_TRN_SMGN: r = m < 0 ? -((-m) >> d) : h; break;
        default: break;  // TRN::TCPL
        }
    }
    const int W = f.I + f.F;
    const __int128 hi = ((__int128)1 << W) - 1, lo = f.S ? -((__int128)1 << W) : 0;
    switch (f.O) {
    case QG_SAT_TCPL: r = r > hi ? hi : (r < lo ? lo : r); break;
    case QG_SAT_ZERO: r = (r > hi || r < lo) ? 0 : r; break;
    case QG_SAT_SMGN: { const __int128 l2 = f.S ? -hi : 0; r = r > hi ? hi : (r < l2 ? l2 : r); break; }
    default:  // WRP::TCPL
        if (f.S) { const __int128 mask = ((__int128)1 << (W + 1)) - 1, w = r & mask; r = (w >> W) ? (w | ~mask) : w; }
        else r &= hi;
        break;
    }
    return int64_t(r);
}

} // namespace detail

// ------------------------------------------------------------------ scalar
template <int I, int F, bool S, class QM, class OM>
class Qu_s<intBits<I>, fracBits<F>, isSigned<S>, QuMode<QM>, OfMode<OM>> {
public:
    static_assert(I + F >= 0, "The total number of bits must be non-negative.");
    static constexpr int intB = I, fracB = F;
    static constexpr bool isS = S;
    static constexpr int QuM = QM::value, OfM = OM::value;
    static constexpr bool is_complex = false;
    static constexpr Fmt fmt = Fmt{I, F, S, QM::value, OM::value};
    // storage: always a sign bit; one int32 up to 32 bits, int64 beyond (ArbiInt<N>, QuBLAS.h:353, :2384-2385)
    using raw_t = std::conditional_t<(1 + I + F <= 32), int32_t, int64_t>;
    raw_t data = 0;

    constexpr Qu_s() = default;
    // QuMode<RND::CONV>: the reference's construction from a double goes through a 2400-bit buffer whose CONV branch
    // returns an artefact (the format maximum for every negative input; tests/test_from_double.py).  This header does not
    // imitate it and does not silently differ either: it throws, unless QUBLAS_AMD_ARITHMETIC_CONV is defined, in which
    // case the value is rounded half-to-even (the definition the reference's own <= 64-bit conversions implement).
    Qu_s(double v) : data(raw_t(detail::from_double(v, fmt)))
    {
#ifndef QUBLAS_AMD_ARITHMETIC_CONV
        if constexpr (QM::value == QG_RND_CONV)
            throw std::runtime_error("Qu(double) with QuMode<RND::CONV>: reference result is a multi-word artefact; define QUBLAS_AMD_ARITHMETIC_CONV for round-half-even");
#endif
    }
    double toDouble() const { return std::ldexp(double(data), -F); }
    Qu_s& fill(int64_t raw) { data = raw_t(raw); return *this; }  // raw store, no range check (QuBLAS.h:2447-2452)
};

template <typename... Args> struct QuInput {
    static constexpr TagSet t = detail::parse<Args...>::value;
    static constexpr Fmt f = [] { Fmt d; return Fmt{t.hasI ? t.I : d.I, t.hasF ? t.F : d.F, t.hasS ? t.S : d.S, t.hasQ ? t.Q : d.Q, t.hasO ? t.O : d.O}; }();
};

namespace detail {
template <int code> struct qmode_t;
template <> struct qmode_t<QG_RND_POS_INF> { using type = RND::POS_INF; };
template <> struct qmode_t<QG_RND_NEG_INF> { using type = RND::NEG_INF; };
template <> struct qmode_t<QG_RND_ZERO> { using type = RND::ZERO; };
template <> struct qmode_t<QG_RND_INF> { using type = RND::INF; };
template <> struct qmode_t<QG_RND_CONV> { using type = RND::CONV; };
template <> struct qmode_t<QG_TRN_TCPL> { using type = TRN::TCPL; };
template <> struct qmode_t<QG_TRN_SMGN> { using type = TRN::SMGN; };
template <int code> struct omode_t;
template <> struct omode_t<QG_SAT_TCPL> { using type = SAT::TCPL; };
template <> struct omode_t<QG_SAT_ZERO> { using type = SAT::ZERO; };
template <> struct omode_t<QG_SAT_SMGN> { using type = SAT::SMGN; };
template <> struct omode_t<QG_WRP_TCPL> { using type = WRP::TCPL; };
template <> struct omode_t<QG_WRP_TCPL_SAT> { using type = WRP::TCPL_SAT<0>; };

template <Fmt f>
using scalar_of = Qu_s<intBits<f.I>, fracBits<f.F>, isSigned<f.S>, QuMode<typename qmode_t<f.Q>::type>, OfMode<typename omode_t<f.O>::type>>;

// Qu<…> front-end: order-free optional tags -> canonical scalar; dim<> first -> tensor; two scalar
// types -> complex (QuInputHelper, QuBLAS.h:2480-2498, :2607-2617)
template <typename... Args> struct qu_helper { using type = scalar_of<QuInput<Args...>::f>; };
template <int I, int F, bool S, class Q, class O>
struct qu_helper<Qu_s<intBits<I>, fracBits<F>, isSigned<S>, QuMode<Q>, OfMode<O>>> {
    using type = Qu_s<intBits<I>, fracBits<F>, isSigned<S>, QuMode<Q>, OfMode<O>>;
};
template <typename... R, typename... Im> struct qu_helper<Qu_s<R...>, Qu_s<Im...>> { using type = Qu_s<Qu_s<R...>, Qu_s<Im...>>; };
template <typename... R, typename... Im> struct qu_helper<Qu_s<Qu_s<R...>, Qu_s<Im...>>> { using type = Qu_s<Qu_s<R...>, Qu_s<Im...>>; };
template <size_t... D, typename... Rest> struct qu_helper<dim<D...>, Rest...> { using type = Qu_s<dim<D...>, typename qu_helper<Rest...>::type>; };
} // namespace detail

template <typename... Args> using Qu = typename detail::qu_helper<Args...>::type;

// ------------------------------------------------------------------ complex
template <typename... R, typename... Im>
class Qu_s<Qu_s<R...>, Qu_s<Im...>> {
public:
    using realType = Qu_s<R...>;
    using imagType = Qu_s<Im...>;
    static constexpr bool is_complex = true;
    realType real;
    imagType imag;
    constexpr Qu_s() = default;
    template <class T1, class T2> Qu_s(T1 a, T2 b) : real(a), imag(b) {}
    template <class T> Qu_s(T a) : real(a), imag(0) {}
};
template <class R, class Im> using Qcomplex = Qu_s<R, Im>;

// ------------------------------------------------------------------ tensor (column-major, QuBLAS.h:2680-2692)
template <size_t... D, typename Elem>
class Qu_s<dim<D...>, Elem> {
public:
    using size = dim<D...>;
    using elem_t = Elem;
    static constexpr size_t elemSize = size::elems;
    std::vector<Elem> data = std::vector<Elem>(elemSize);  // contiguous; data.data() is what the engine reads

    Qu_s() = default;
    template <class... V>
        requires(sizeof...(V) == elemSize && sizeof...(V) > 1)
    Qu_s(V... v) : data{Elem(v)...} {}

    Elem& operator[](size_t i) { return data[i]; }
    const Elem& operator[](size_t i) const { return data[i]; }
    template <class... Ix>
        requires(sizeof...(Ix) == sizeof...(D) && sizeof...(Ix) > 1)
    Elem& operator[](Ix... ix) { return data[linear(ix...)]; }
    template <class... Ix>
        requires(sizeof...(Ix) == sizeof...(D) && sizeof...(Ix) > 1)
    const Elem& operator[](Ix... ix) const { return data[linear(ix...)]; }

    // uniform raw values over the whole representable range, like Qu::fill() (QuBLAS.h:526-536)
    Qu_s& fill(uint64_t seed = 1)
    {
        std::mt19937_64 g(seed);
        auto one = [&](auto& s) {
            using S = std::remove_reference_t<decltype(s)>;
            const int W = S::intB + S::fracB;
            const int64_t lo = S::isS ? -(int64_t(1) << W) : 0, hi = (int64_t(1) << W) - 1;
            s.data = typename S::raw_t(std::uniform_int_distribution<int64_t>(lo, hi)(g));
        };
        for (auto& e : data) {
            if constexpr (Elem::is_complex) { one(e.real); one(e.imag); }
            else one(e);
        }
        return *this;
    }

private:
    template <class... Ix> static size_t linear(Ix... ix)
    {
        const size_t idx[] = {size_t(ix)...};
        size_t lin = 0, stride = 1;
        for (size_t k = 0; k < sizeof...(D); ++k) { lin += idx[k] * stride; stride *= size::extent[k]; }
        return lin;
    }
};

// ------------------------------------------------------------------ complex-multiply wrappers and Qgemul tags
template <typename... A> struct BasicComplexMul {};
template <typename... A> struct TFComplexMul {};
template <typename... A> struct acT {}; template <typename... A> struct bdT {}; template <typename... A> struct adT {};
template <typename... A> struct bcT {}; template <typename... A> struct acbdT {}; template <typename... A> struct adbcT {};
template <typename... A> struct abT {}; template <typename... A> struct cdT {}; template <typename... A> struct baT {};
template <typename... A> struct abcT {}; template <typename... A> struct cdbT {}; template <typename... A> struct badT {};
template <typename... A> struct ABT {}; template <typename... A> struct BCT {};
template <typename... A> struct realT {}; template <typename... A> struct imagT {};   // per-part tags of complex Qadd / Qsub / complex x real Qmul (QuBLAS.h:3537-3547)

template <typename... A> struct QgemulAddArgs {};
template <typename... A> struct QgemulMulArgs {};
template <bool V> struct QgemulTransposedA { static constexpr bool value = V; };

namespace detail {

// the tag set a sub-operation sees: its own wrapper's arguments if present (first match), else
// every argument of the enclosing multiplier (the default `xT<toArgs…>`, QuBLAS.h:3429, :164-170)
template <template <typename...> class W, typename... All> struct sub_tags {
    template <typename... Tail> struct search { static constexpr bool found = false; static constexpr TagSet value{}; };
    template <typename... Own, typename... Tail> struct search<W<Own...>, Tail...> {
        static constexpr bool found = true;
        static constexpr TagSet value = parse<Own...>::value;
    };
    template <typename T0, typename... Tail> struct search<T0, Tail...> : search<Tail...> {};
    static constexpr TagSet value = search<All...>::found ? search<All...>::value : parse<All...>::value;
};

// the tag set one PART of a complex element-wise operator sees: realT<…> / imagT<…> if present, else every argument
// (QuBLAS.h:3551-3552); two bare scalar types are <realT<first>, imagT<second>> (:3566-3568)
template <int P, typename... Tags> struct part_tags {
    static constexpr TagSet value = P == 0 ? sub_tags<realT, Tags...>::value : sub_tags<imagT, Tags...>::value;
};
template <int P, int I1, int F1, bool S1, class Q1, class O1, int I2, int F2, bool S2, class Q2, class O2>
struct part_tags<P, Qu_s<intBits<I1>, fracBits<F1>, isSigned<S1>, QuMode<Q1>, OfMode<O1>>, Qu_s<intBits<I2>, fracBits<F2>, isSigned<S2>, QuMode<Q2>, OfMode<O2>>> {
    static constexpr TagSet value = P == 0 ? parse<Qu_s<intBits<I1>, fracBits<F1>, isSigned<S1>, QuMode<Q1>, OfMode<O1>>>::value
                                           : parse<Qu_s<intBits<I2>, fracBits<F2>, isSigned<S2>, QuMode<Q2>, OfMode<O2>>>::value;
};

template <class E> constexpr Fmt re_fmt() { if constexpr (E::is_complex) return E::realType::fmt; else return E::fmt; }
template <class E> constexpr Fmt im_fmt() { if constexpr (E::is_complex) return E::imagType::fmt; else return E::fmt; }

struct Slots { int cmul = QG_CMUL_NONE; Fmt m[8]{}; Fmt prod[2]{}; };

template <class EA, class EB, class MulList> struct slots_of;
template <class EA, class EB, typename... Tags>
    requires(!EA::is_complex)
struct slots_of<EA, EB, TypeList<Tags...>> {
    static constexpr Slots value = [] { Slots s; s.m[0] = merge_mul(EA::fmt, EB::fmt, parse<Tags...>::value); s.prod[0] = s.prod[1] = s.m[0]; return s; }();
};
template <class EA, class EB, typename... Args>
    requires(EA::is_complex)
struct slots_of<EA, EB, TypeList<BasicComplexMul<Args...>>> {
    static constexpr Slots value = [] {
        constexpr Fmt a = re_fmt<EA>(), b = im_fmt<EA>(), c = re_fmt<EB>(), d = im_fmt<EB>();
        Slots s; s.cmul = QG_CMUL_BASIC;
        s.m[QG_B_AC] = merge_mul(a, c, sub_tags<acT, Args...>::value);
        s.m[QG_B_BD] = merge_mul(b, d, sub_tags<bdT, Args...>::value);
        s.m[QG_B_AD] = merge_mul(a, d, sub_tags<adT, Args...>::value);
        s.m[QG_B_BC] = merge_mul(b, c, sub_tags<bcT, Args...>::value);
        s.m[QG_B_RE] = merge_add(s.m[QG_B_AC], s.m[QG_B_BD], sub_tags<acbdT, Args...>::value);
        s.m[QG_B_IM] = merge_add(s.m[QG_B_AD], s.m[QG_B_BC], sub_tags<adbcT, Args...>::value);
        s.prod[0] = s.m[QG_B_RE]; s.prod[1] = s.m[QG_B_IM];
        return s;
    }();
};
template <class EA, class EB>
    requires(EA::is_complex)
struct slots_of<EA, EB, TypeList<>> : slots_of<EA, EB, TypeList<BasicComplexMul<>>> {};  // QuBLAS.h:3422-3424
template <class EA, class EB, typename... Args>
    requires(EA::is_complex)
struct slots_of<EA, EB, TypeList<TFComplexMul<Args...>>> {
    static constexpr Slots value = [] {
        constexpr Fmt a = re_fmt<EA>(), b = im_fmt<EA>(), c = re_fmt<EB>(), d = im_fmt<EB>();
        Slots s; s.cmul = QG_CMUL_TF;
        s.m[QG_T_AB] = merge_add(a, b, sub_tags<abT, Args...>::value);
        s.m[QG_T_CD] = merge_add(c, d, sub_tags<cdT, Args...>::value);
        s.m[QG_T_BA] = merge_add(b, a, TagSet{});                                    // baT is never honoured (QuBLAS.h:3515)
        s.m[QG_T_A] = merge_mul(s.m[QG_T_AB], c, sub_tags<abcT, Args...>::value);
        s.m[QG_T_B] = merge_mul(s.m[QG_T_CD], b, sub_tags<badT, Args...>::value);   // B uses badT (QuBLAS.h:3525)
        s.m[QG_T_C] = merge_mul(s.m[QG_T_BA], d, sub_tags<cdbT, Args...>::value);   // C uses cdbT (QuBLAS.h:3526)
        s.m[QG_T_RE] = merge_add(s.m[QG_T_A], s.m[QG_T_B], sub_tags<ABT, Args...>::value);
        s.m[QG_T_IM] = merge_add(s.m[QG_T_B], s.m[QG_T_C], sub_tags<BCT, Args...>::value);
        s.prod[0] = s.m[QG_T_RE]; s.prod[1] = s.m[QG_T_IM];
        return s;
    }();
};

// level list -> array of (re, im) formats
template <class List> struct levels_of;
template <typename... Ls> struct levels_of<TypeList<Ls...>> {
    static constexpr size_t n = sizeof...(Ls);
    static constexpr std::array<std::array<Fmt, 2>, (n ? n : 1)> value = [] {
        std::array<std::array<Fmt, 2>, (n ? n : 1)> v{};
        size_t i = 0;
        ((v[i][0] = re_fmt<Ls>(), v[i][1] = im_fmt<Ls>(), ++i), ...);
        return v;
    }();
    static constexpr bool all_complex = (Ls::is_complex && ... && true);
    static constexpr bool all_real = (!Ls::is_complex && ... && true);
};

template <class... Tags> struct pick_add { using type = TypeList<>; };
template <class... L, class... Rest> struct pick_add<QgemulAddArgs<L...>, Rest...> { using type = TypeList<L...>; };
template <class... L, class... Rest> struct pick_add<QgemulAddArgs<TypeList<L...>>, Rest...> { using type = TypeList<L...>; };
template <class T, class... Rest> struct pick_add<T, Rest...> : pick_add<Rest...> {};
template <class... Tags> struct pick_mul { using type = TypeList<>; };
template <class... L, class... Rest> struct pick_mul<QgemulMulArgs<L...>, Rest...> { using type = TypeList<L...>; };
template <class T, class... Rest> struct pick_mul<T, Rest...> : pick_mul<Rest...> {};
template <class... Tags> struct pick_ta { static constexpr bool value = false; };
template <bool V, class... Rest> struct pick_ta<QgemulTransposedA<V>, Rest...> { static constexpr bool value = V; };
template <class T, class... Rest> struct pick_ta<T, Rest...> : pick_ta<Rest...> {};

} // namespace detail

// lower a Qgemul call to the C-ABI descriptor (pure host computation, no device access)
template <typename... Tags, size_t CM, size_t CN, size_t AR, size_t AC, size_t BKd, size_t BN, class EC, class EA, class EB>
qgemul_desc Qgemul_lower_types(std::type_identity<Qu_s<dim<CM, CN>, EC>>, std::type_identity<Qu_s<dim<AR, AC>, EA>>,
                               std::type_identity<Qu_s<dim<BKd, BN>, EB>>);

template <typename... Tags, class TC, class TA, class TB>
qgemul_desc Qgemul_lower(const TC&, const TA&, const TB&)
{
    return Qgemul_lower_types<Tags...>(std::type_identity<TC>{}, std::type_identity<TA>{}, std::type_identity<TB>{});
}

// (only the operand TYPES enter the lowering)
template <typename... Tags, size_t CM, size_t CN, size_t AR, size_t AC, size_t BKd, size_t BN, class EC, class EA, class EB>
qgemul_desc Qgemul_lower_types(std::type_identity<Qu_s<dim<CM, CN>, EC>>, std::type_identity<Qu_s<dim<AR, AC>, EA>>,
                               std::type_identity<Qu_s<dim<BKd, BN>, EB>>)
{
    using namespace detail;
    constexpr bool ta = pick_ta<Tags...>::value;
    constexpr size_t M = ta ? AC : AR, K = ta ? AR : AC;
    static_assert(M == CM && BN == CN && BKd == K, "Qgemul: C is MxN, A is MxK (KxM when transposed), B is KxN");
    static_assert(EA::is_complex == EB::is_complex && EA::is_complex == EC::is_complex, "Qgemul: all real or all complex");
    using add_list = typename pick_add<Tags...>::type;
    using L = levels_of<add_list>;
    static_assert(EA::is_complex ? L::all_complex : L::all_real, "Qgemul: level types must be complex for complex operands, real otherwise");
    constexpr Slots s = slots_of<EA, EB, typename pick_mul<Tags...>::type>::value;
    qgemul_desc d{};
    d.abi = QGEMUL_ABI_VERSION;
    d.transA = ta;
    d.is_complex = EA::is_complex;
    d.cmul = uint8_t(s.cmul);
    d.M = int64_t(M); d.N = int64_t(CN); d.K = int64_t(K);
    d.a[0] = re_fmt<EA>().c(); d.a[1] = im_fmt<EA>().c();
    d.b[0] = re_fmt<EB>().c(); d.b[1] = im_fmt<EB>().c();
    d.c[0] = re_fmt<EC>().c(); d.c[1] = im_fmt<EC>().c();
    for (int i = 0; i < (s.cmul == QG_CMUL_NONE ? 1 : s.cmul == QG_CMUL_TF ? 8 : 6); ++i) d.mul[i] = s.m[i].c();
    for (size_t k = K; k > 1; k = (k + 1) / 2) ++d.n_levels;
    Fmt prev[2] = {s.prod[0], s.prod[1]};
    for (uint32_t l = 0; l < d.n_levels; ++l) {
        for (int p = 0; p < 2; ++p) {
            Fmt buf = prev[p], add = prev[p];
            if constexpr (L::n > 0) {
                buf = L::value[l < L::n ? l : L::n - 1][p];
                // real: Qadd<T_l> yields T_l; complex: the complex tag is ignored, the add is the default
                // merge of two equal formats (= that format) and the level buffer converts (QuBLAS.h:4966)
                add = EA::is_complex ? merge_add(prev[p], prev[p], TagSet{}) : buf;
            }
            d.level_add[p][l] = add.c();
            d.level[p][l] = buf.c();
            prev[p] = buf;
        }
    }
    return d;
}

// Options of the one-shot calls below, per translation unit (QG_OPT_* of qgemul.h).  Several GPUs in one process:
//     QgemulRunFlags() |= QG_OPT_ALL_DEVICES;     // every Qgemul<...>(C, A, B) is row-sharded over all visible gfx950 devices
// QgemulRelease() frees what the library caches for the calling thread (context, plan, device buffers); the library also does
// it when the thread exits.
inline uint32_t& QgemulRunFlags()
{
    static uint32_t flags = 0;
    return flags;
}
inline void QgemulRelease() { qgemul_run_release(); }
// Descriptor flags OR-ed into every lowered Qgemul of this translation unit.  QgemulDescFlags() |= QG_DESC_REFERENCE_ARTEFACTS asks
// for the reference's own result where that is an implementation artefact the engine otherwise refuses (qgemul.h: C of an unsigned
// WRP::TCPL format with exactly 32 value bits comes out unwrapped).
inline uint8_t& QgemulDescFlags()
{
    static uint8_t flags = 0;
    return flags;
}

// C = A' * B, quantised per product and per tree node exactly as the reference's primitives do
template <typename... Tags, class TC, class TA, class TB>
void Qgemul(TC& C, const TA& A, const TB& B)
{
    qgemul_desc d = Qgemul_lower<Tags...>(C, A, B);
    d.flags |= QgemulDescFlags();
    qgemul_opts opts{};
    opts.device = -1;
    opts.flags = QgemulRunFlags();
    const int st = qgemul_run(&d, C.data.data(), A.data.data(), B.data.data(), &opts);
    if (st != QG_OK) throw std::runtime_error(std::string("Qgemul: ") + qgemul_strerror(st));
}

// ------------------------------------------------------------------ element-wise operators after the GEMM (SURVEY.md §8-f "next" #2)
// The reference's lazy tensor operators Qmul / Qadd / Qsub<tags…>(tensor, tensor | scalar) (QuBLAS.h:3780-3877,
// front-ends :4079-4100) applied to a Qgemul result:
//     Qgemul<…>(C, A, B);   Qu<dim<M,N>, T1> t = Qmul<t1…>(C, s);   Qu<dim<M,N>, DT> D = Qadd<t2…>(t, Bias);
// written as ONE call that never materialises C or t:
//     Qgemul<…, QgemulResult<CT>>(D, A, B, ThenMul<T1, t1…>(s), ThenAdd<void, t2…>(Bias));
// QgemulResult<CT> names the element type the Qgemul result WOULD have (C's type above).  Then{Mul,Add,Sub}<Into, tags…>(e)
// is Qop<tags…>(x, e) with x the running value; ThenRsub is Qsub<tags…>(e, x).  Into = the element type of the tensor
// the operator's result is assigned to before the next operator (void: the operator's own result type); the last
// operator's result is assigned to D.  e is a tensor of D's shape or a scalar.  After a complex Qgemul: complex operands
// for ThenAdd / ThenSub / ThenRsub (realT<…> / imagT<…> tags), real operands for all four; no complex x complex ThenMul.
template <class CT> struct QgemulResult {};

namespace detail {
template <int OP, bool XFIRST, class Into, class Operand, typename... Tags>
struct EwStage {
    const Operand& e;
    static constexpr int op = OP;
    static constexpr bool x_first = XFIRST;
    static constexpr bool scalar = !requires { typename Operand::elem_t; };   // tensors have an element type
    template <class O, bool = !requires { typename O::elem_t; }> struct elem { using type = O; };
    template <class O> struct elem<O, false> { using type = typename O::elem_t; };
    using e_t = typename elem<Operand>::type;
    static constexpr bool e_complex = e_t::is_complex;
    static constexpr Fmt efmt = re_fmt<e_t>();
    static constexpr Fmt result(Fmt x)
    {
        const Fmt a = XFIRST ? x : efmt, b = XFIRST ? efmt : x;
        return OP == QG_EW_MUL ? merge_mul(a, b, parse<Tags...>::value) : merge_add(a, b, parse<Tags...>::value);
    }
    static constexpr Fmt into(Fmt r) { if constexpr (std::is_void_v<Into>) return r; else return Into::fmt; }
    // ---- on a complex running value: the stage of part P (include/qgemul.h's table) and its result format
    template <int P> static constexpr Fmt efmt_part() { return P == 0 ? re_fmt<e_t>() : im_fmt<e_t>(); }
    template <int P> static constexpr int op_part() { return (P == 1 && !e_complex && OP != QG_EW_MUL && (OP == QG_EW_ADD || XFIRST)) ? int(QG_EW_PASS) : OP; }
    template <int P> static constexpr bool scalar_part() { return scalar || (P == 1 && !e_complex && OP != QG_EW_MUL); }
    template <int P> static constexpr Fmt result_part(Fmt x)
    {
        if (op_part<P>() == QG_EW_PASS) return x;
        // complex (+|-) real hands ALL its tags to the one real Qadd / Qsub (QuBLAS.h:3654, :3670, :3686, :3701)
        const TagSet t = (e_complex || OP == QG_EW_MUL) ? part_tags<P, Tags...>::value : parse<Tags...>::value;
        const Fmt e = efmt_part<P>(), a = XFIRST ? x : e, b = XFIRST ? e : x;
        return OP == QG_EW_MUL ? merge_mul(a, b, t) : merge_add(a, b, t);
    }
    template <int P> static constexpr Fmt into_part(Fmt r)
    {
        if constexpr (std::is_void_v<Into>) return r; else return P == 0 ? re_fmt<Into>() : im_fmt<Into>();
    }
};
template <class... Tags> struct pick_result { using type = void; };
template <class CT, class... Rest> struct pick_result<QgemulResult<CT>, Rest...> { using type = CT; };
template <class T, class... Rest> struct pick_result<T, Rest...> : pick_result<Rest...> {};

template <class... Stages>
constexpr qgemul_epilogue lower_chain(Fmt c, Fmt d)
{
    static_assert(sizeof...(Stages) <= QG_MAX_EW, "at most QG_MAX_EW element-wise operators");
    qgemul_epilogue ep{};
    ep.n_stages = sizeof...(Stages);
    Fmt x = c;
    uint32_t k = 0;
    ([&] {
        const Fmt r = Stages::result(x);
        ep.stage[k].op = uint8_t(Stages::op);
        ep.stage[k].x_first = Stages::x_first;
        ep.stage[k].e_scalar = Stages::scalar;
        ep.stage[k].e = Stages::efmt.c();
        ep.stage[k].r = r.c();
        x = Stages::into(r);
        ep.stage[k].t = x.c();
        ++k;
    }(), ...);
    ep.d = d.c();
    return ep;
}

template <class... Stages>
constexpr qgemul_epilogue_cplx lower_chain_cplx(Fmt cre, Fmt cim, Fmt dre, Fmt dim_)
{
    static_assert(sizeof...(Stages) <= QG_MAX_EW, "at most QG_MAX_EW element-wise operators");
    static_assert((!(Stages::e_complex && Stages::op == QG_EW_MUL) && ...), "complex x complex multiplication mixes the parts: not an element-wise stage");
    qgemul_epilogue_cplx ep{};
    ep.part[0].n_stages = ep.part[1].n_stages = sizeof...(Stages);
    Fmt x[2] = {cre, cim};
    uint32_t k = 0;
    ([&] {
        ep.e_complex[k] = Stages::e_complex;
        auto one = [&]<int P>() {
            qgemul_ew_stage& s = ep.part[P].stage[k];
            const Fmt r = Stages::template result_part<P>(x[P]);
            s.op = uint8_t(Stages::template op_part<P>());
            s.x_first = Stages::x_first;
            s.e_scalar = Stages::template scalar_part<P>();
            s.e = Stages::template efmt_part<P>().c();
            s.r = r.c();
            x[P] = Stages::template into_part<P>(r);
            s.t = x[P].c();
        };
        one.template operator()<0>();
        one.template operator()<1>();
        ++k;
    }(), ...);
    ep.part[0].d = dre.c();
    ep.part[1].d = dim_.c();
    return ep;
}
} // namespace detail

template <class Into = void, typename... Tags, class Operand> auto ThenMul(const Operand& e) { return detail::EwStage<QG_EW_MUL, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenAdd(const Operand& e) { return detail::EwStage<QG_EW_ADD, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenSub(const Operand& e) { return detail::EwStage<QG_EW_SUB, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenRsub(const Operand& e) { return detail::EwStage<QG_EW_SUB, false, Into, Operand, Tags...>{e}; }

// the chain's C-ABI form (pure host computation)
template <typename... Tags, class TD, class... Stages>
constexpr qgemul_epilogue Qgemul_lower_epilogue(const TD&, const Stages&...)
{
    using CT = typename detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>: the element type of the Qgemul result");
    static_assert(!CT::is_complex && !TD::elem_t::is_complex, "a complex chain is lowered by Qgemul_lower_epilogue_cplx");
    return detail::lower_chain<Stages...>(CT::fmt, TD::elem_t::fmt);
}
// the same after a COMPLEX Qgemul (include/qgemul.h, qgemul_epilogue_cplx): complex operands for ThenAdd / ThenSub / ThenRsub
// with realT<…> / imagT<…> tags (QuBLAS.h:3549-3589), real operands for all four (:3604-3707)
template <typename... Tags, class TD, class... Stages>
constexpr qgemul_epilogue_cplx Qgemul_lower_epilogue_cplx(const TD&, const Stages&...)
{
    using CT = typename detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>: the element type of the Qgemul result");
    static_assert(CT::is_complex && TD::elem_t::is_complex, "a complex chain runs from a complex Qgemul result into a complex tensor");
    using DT = typename TD::elem_t;
    return detail::lower_chain_cplx<Stages...>(CT::realType::fmt, CT::imagType::fmt, DT::realType::fmt, DT::imagType::fmt);
}

// D = the element-wise chain applied to A' * B
template <typename... Tags, class TD, class TA, class TB, class S0, class... Stages>
void Qgemul(TD& D, const TA& A, const TB& B, const S0& s0, const Stages&... st)
{
    using CT = typename detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>");
    qgemul_desc d = Qgemul_lower_types<Tags...>(std::type_identity<Qu_s<typename TD::size, CT>>{}, std::type_identity<TA>{}, std::type_identity<TB>{});
    d.flags |= QgemulDescFlags();
    auto ptr = [](const auto& stage) -> const void* {
        if constexpr (std::remove_cvref_t<decltype(stage)>::scalar) return &stage.e;   // one element as the tensors store them
        else {
            static_assert(std::is_same_v<typename std::remove_cvref_t<decltype(stage.e)>::size, typename TD::size>, "a tensor operand has D's shape");
            return stage.e.data.data();
        }
    };
    const void* E[QG_MAX_EW] = {ptr(s0), ptr(st)...};
    int rc;
    if constexpr (CT::is_complex) {
        const qgemul_epilogue_cplx ep = Qgemul_lower_epilogue_cplx<Tags...>(D, s0, st...);
        rc = qgemul_run_epc(&d, &ep, D.data.data(), A.data.data(), B.data.data(), E, nullptr);
    } else {
        const qgemul_epilogue ep = Qgemul_lower_epilogue<Tags...>(D, s0, st...);
        rc = qgemul_run_ep(&d, &ep, D.data.data(), A.data.data(), B.data.data(), E, nullptr);
    }
    if (rc != QG_OK) throw std::runtime_error(std::string("Qgemul: ") + qgemul_strerror(rc));
}

// ------------------------------------------------------------------ Qreduce (SURVEY.md §8-f "next" #1)
// The reference's tree reduction of a tensor, Qreduce<L…>(v) (QuBLAS.h:4960-4990, :5014-5018), on the
// same engine path: C[1 x 1] = A[1 x len] * ones[len x 1] with the product format equal to the element
// format (Qmul(a, 1) into a's own format is the identity; for a signed SAT::SMGN element type, whose conversion would clamp
// the raw minimum -2^W that the reference's Qreduce adds as it is, the leaf format is the element's with SAT::TCPL and the
// default level type — the element type — is named explicitly) and the level list L.  The result type is
// the reducer's: the last level type used, or the element type when there are no levels / one element.
namespace detail {
template <class Elem, size_t Len, class List> struct reduce_result { using type = Elem; };
template <class Elem, size_t Len, class L0, class... Ls>
struct reduce_result<Elem, Len, TypeList<L0, Ls...>> {
    static constexpr size_t nl = [] { size_t n = 0; for (size_t k = Len; k > 1; k = (k + 1) / 2) ++n; return n; }();
    static constexpr size_t n = 1 + sizeof...(Ls);
    static constexpr Fmt f = levels_of<TypeList<L0, Ls...>>::value[nl == 0 ? 0 : (nl < n ? nl : n) - 1][0];
    using type = std::conditional_t<(Len <= 1), Elem, scalar_of<f>>;
};
template <class... Ls> struct as_list { using type = TypeList<Ls...>; };
template <class... Ls> struct as_list<TypeList<Ls...>> { using type = TypeList<Ls...>; };
} // namespace detail

template <typename... Levels, size_t... D, class Elem>
    requires(!Elem::is_complex)
auto Qreduce(const Qu_s<dim<D...>, Elem>& v)
{
    using namespace detail;
    constexpr size_t len = dim<D...>::elems;
    using list = typename as_list<Levels...>::type;
    using res_t = typename reduce_result<Elem, len, list>::type;
    using one_t = Qu<intBits<1>, fracBits<0>, isSigned<false>>;
    Qu_s<dim<len, 1>, Elem> a;       // dim<K, M> with TransposedA: the vector is row 0 of A'
    a.data = v.data;
    Qu_s<dim<len, 1>, one_t> ones;
    for (auto& o : ones.data) o.data = 1;
    constexpr bool smgn = Elem::isS && Elem::OfM == QG_SAT_SMGN;
    using leaf_t = std::conditional_t<smgn, scalar_of<Fmt{Elem::intB, Elem::fracB, Elem::isS, Elem::QuM, int(QG_SAT_TCPL)}>, Elem>;
    using lv_t = std::conditional_t<(smgn && std::is_same_v<list, TypeList<>>), TypeList<Elem>, list>;
    Qu_s<dim<1, 1>, std::conditional_t<(smgn && len <= 1), leaf_t, res_t>> c;
    [&]<class... Ls>(TypeList<Ls...>) {
        qgemul_desc d = Qgemul_lower<QgemulAddArgs<Ls...>, QgemulMulArgs<leaf_t>, QgemulTransposedA<true>>(c, a, ones);
        // level 0's type IS the element type: the reference copies an odd leftover into the level-0 buffer unconverted
        // (a same-type copy, QuBLAS.h:4977-4980), so the raw minimum survives it (include/qgemul.h, QG_DESC_LEFTOVER0_COPY)
        if constexpr (smgn && sizeof...(Ls) > 0)
            if (levels_of<TypeList<Ls...>>::value[0][0] == Elem::fmt) d.flags |= QG_DESC_LEFTOVER0_COPY;
        qgemul_opts opts{};
        opts.device = -1;
        opts.flags = QgemulRunFlags();
        const int st = qgemul_run(&d, c.data.data(), a.data.data(), ones.data.data(), &opts);
        if (st != QG_OK) throw std::runtime_error(std::string("Qreduce: ") + qgemul_strerror(st));
    }(lv_t{});
    res_t r;
    r.data = typename res_t::raw_t(c.data[0].data);
    return r;
}

// ---- the VARIADIC overload: Qreduce<L...>(q1, q2, ...) — any number of scalars of any (real) types (readme.md:62;
// QuBLAS.h:4924-4951).  Level l adds neighbours with Qadd<T_l> (T_l = L[min(l, n-1)]; no L: the default merge of the two
// operand types); an odd leftover is added AFTER the recursion over the pair sums, with the CURRENT level's type (the vector
// overload above copies it into the next level instead: the two differ for lengths that are not powers of two).  Every
// Qadd<T>(x, y) of two scalars is ONE two-term Qgemul on the engine: both operands are written exactly in a common
// super-format (the alignment shifts Qadd performs itself, QuBLAS.h:3190), times 1, and level 0 is the node's result type.
// "Please avoid using this version for efficiency" holds here as it does in the reference (a device call per node).
namespace detail {
template <class T> concept RealScalar = requires { T::fmt; typename T::raw_t; } && !T::is_complex && !requires { typename T::elem_t; };

template <TagSet tags, class X, class Y>
auto add_node(const X& x, const Y& y)
{
    constexpr Fmt fx = X::fmt, fy = Y::fmt, fr = merge_add(fx, fy, tags);
    constexpr int F = fx.F > fy.F ? fx.F : fy.F;
    constexpr Fmt sup{fx.I > fy.I ? fx.I : fy.I, F, fx.S || fy.S, int(QG_TRN_TCPL), int(QG_SAT_TCPL)};
    using sup_t = scalar_of<sup>;
    using res_t = scalar_of<fr>;
    using one_t = Qu<intBits<1>, fracBits<0>, isSigned<false>>;
    Qu_s<dim<2, 1>, sup_t> a;                    // dim<K, M> with TransposedA: the two operands are row 0 of A'
    a.data[0].data = typename sup_t::raw_t(int64_t(x.data) << (F - fx.F));
    a.data[1].data = typename sup_t::raw_t(int64_t(y.data) << (F - fy.F));
    Qu_s<dim<2, 1>, one_t> ones;
    for (auto& o : ones.data) o.data = 1;
    Qu_s<dim<1, 1>, res_t> c;
    Qgemul<QgemulAddArgs<res_t>, QgemulMulArgs<sup_t>, QgemulTransposedA<true>>(c, a, ones);
    return c.data[0];
}

template <size_t layer, class List, class... Ts>
auto reduce_variadic(const Ts&... q)
{
    constexpr size_t n = sizeof...(Ts);
    auto tup = std::tie(q...);
    if constexpr (n == 1) {
        return std::get<0>(tup);
    } else {
        using L = levels_of<List>;
        constexpr TagSet tags = [] {
            if constexpr (L::n == 0) return TagSet{};
            else {
                constexpr Fmt f = L::value[layer < L::n ? layer : L::n - 1][0];
                return TagSet{true, true, true, true, true, false, f.I, f.F, f.Q, f.O, f.S};
            }
        }();
        auto rest = [&]<size_t... I>(std::index_sequence<I...>) {
            return reduce_variadic<layer + 1, List>(add_node<tags>(std::get<2 * I>(tup), std::get<2 * I + 1>(tup))...);
        }(std::make_index_sequence<n / 2>{});
        if constexpr (n % 2 == 0) return rest;
        else return add_node<tags>(rest, std::get<n - 1>(tup));
    }
}
} // namespace detail

template <typename... Levels, detail::RealScalar T0, detail::RealScalar T1, detail::RealScalar... Ts>
auto Qreduce(const T0& q0, const T1& q1, const Ts&... qs)
{
    using list = typename detail::as_list<Levels...>::type;
    static_assert(detail::levels_of<list>::all_real, "Qreduce of real scalars needs real level types");
    return detail::reduce_variadic<0, list>(q0, q1, qs...);
}

} // namespace QuBLAS_amd
