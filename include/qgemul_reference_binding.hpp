// qgemul_reference_binding.hpp — the stub a QuBLAS maintainer would add to the reference header's
// "BLAS" section (after Reducer/Qreduce, /root/reference/include/QuBLAS.h:4899-5018) to get the README's
//     Qgemul<QgemulAddArgs<list>, QgemulMulArgs<type1>, QgemulTransposedA<true>>(m3, m1, m1);
// (readme.md:84-87) running on an MI355X through libqugemm.so.
//
// It must be included AFTER the reference's own QuBLAS.h: it contains no arithmetic and no tag
// machinery of its own — every result format is obtained from the reference's types by
// decltype on the reference's own Qmul / Qadd / Qsub front-ends (QuBLAS.h:3980-3999) and on the
// member typedefs of its complex multipliers (:3429-3435, :3513-3520), so the lowering cannot
// drift from the header it extends.  What it adds: the three Qgemul tags, the compile-time
// walk over the tree levels (Reducer::ReducerTypeSelector semantics, :4906-4921, :4966), and one
// call of the C-ABI entry point with the tensors' contiguous storage (Qu_s<dim<…>,T>::data.data(),
// column-major, :2680-2713).
//
// The repository's standalone header include/QuBLAS_amd.h provides the same Qgemul on its own
// minimal tag API for users who do not have the reference header.
#pragma once
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include "qgemul.h"

inline namespace QuBLAS {

template <typename... Args> struct QgemulAddArgs {};
template <typename... Args> struct QgemulMulArgs {};
template <bool Value> struct QgemulTransposedA { static constexpr bool value = Value; };

namespace qgemul_detail {

template <class T> constexpr qfmt fmt_of()
{
    return qfmt{int16_t(T::intB), int16_t(T::fracB), uint8_t(T::isS), uint8_t(T::QuM), uint8_t(T::OfM), 0};
}
template <class T, bool = T::is_complex> struct parts { using re = T; using im = T; };
template <class T> struct parts<T, true> { using re = typename T::realType; using im = typename T::imagType; };

// ---- optional, order-free Qgemul tags (same convention as every other tag pack in the header)
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

template <class List> struct mul_with;
template <class... Tags> struct mul_with<TypeList<Tags...>> {
    template <class X, class Y> static auto go(const X& x, const Y& y) { return Qmul<Tags...>(x, y); }
};

// ---- product sub-operation formats
template <class EA, class EB, class MulList> struct slots;
template <class EA, class EB, class... Tags>
    requires(!EA::is_complex)
struct slots<EA, EB, TypeList<Tags...>> {
    using prod_t = decltype(Qmul<Tags...>(std::declval<EA>(), std::declval<EB>()));
    static constexpr int cmul = QG_CMUL_NONE;
    static void fill(qgemul_desc& d) { d.mul[QG_MUL_REAL] = fmt_of<prod_t>(); }
};
template <class EA, class EB, class... Args>
    requires(EA::is_complex)
struct slots<EA, EB, TypeList<BasicComplexMul<Args...>>> {
    using S = Qmul_s<EA, EB, BasicComplexMul<Args...>>;
    using a_t = typename EA::realType; using b_t = typename EA::imagType;
    using c_t = typename EB::realType; using d_t = typename EB::imagType;
    using ac_t = decltype(Qmul<typename S::mulACType>(std::declval<a_t>(), std::declval<c_t>()));
    using bd_t = decltype(Qmul<typename S::mulBDType>(std::declval<b_t>(), std::declval<d_t>()));
    using ad_t = decltype(Qmul<typename S::mulADType>(std::declval<a_t>(), std::declval<d_t>()));
    using bc_t = decltype(Qmul<typename S::mulBCType>(std::declval<b_t>(), std::declval<c_t>()));
    using re_t = decltype(Qsub<typename S::subACBDType>(std::declval<ac_t>(), std::declval<bd_t>()));
    using im_t = decltype(Qadd<typename S::addADBCType>(std::declval<ad_t>(), std::declval<bc_t>()));
    using prod_t = Qcomplex<re_t, im_t>;
    static constexpr int cmul = QG_CMUL_BASIC;
    static void fill(qgemul_desc& d)
    {
        d.mul[QG_B_AC] = fmt_of<ac_t>(); d.mul[QG_B_BD] = fmt_of<bd_t>(); d.mul[QG_B_AD] = fmt_of<ad_t>();
        d.mul[QG_B_BC] = fmt_of<bc_t>(); d.mul[QG_B_RE] = fmt_of<re_t>(); d.mul[QG_B_IM] = fmt_of<im_t>();
    }
};
template <class EA, class EB>
    requires(EA::is_complex)
struct slots<EA, EB, TypeList<>> : slots<EA, EB, TypeList<BasicComplexMul<>>> {}; // QuBLAS.h:3422-3424
template <class EA, class EB, class... Args>
    requires(EA::is_complex)
struct slots<EA, EB, TypeList<TFComplexMul<Args...>>> {
    using S = Qmul_s<EA, EB, TFComplexMul<Args...>>;
    using a_t = typename EA::realType; using b_t = typename EA::imagType;
    using c_t = typename EB::realType; using d_t = typename EB::imagType;
    using ab_t = decltype(Qadd<typename S::addabType>(std::declval<a_t>(), std::declval<b_t>()));
    using cd_t = decltype(Qadd<typename S::addcdType>(std::declval<c_t>(), std::declval<d_t>()));
    using ba_t = decltype(Qsub<typename S::subbaType>(std::declval<b_t>(), std::declval<a_t>()));
    using A_t = decltype(Qmul<typename S::mulabcType>(std::declval<ab_t>(), std::declval<c_t>()));
    using B_t = decltype(Qmul<typename S::mulbadType>(std::declval<cd_t>(), std::declval<b_t>()));
    using C_t = decltype(Qmul<typename S::mulcdbType>(std::declval<ba_t>(), std::declval<d_t>()));
    using re_t = decltype(Qsub<typename S::subABType>(std::declval<A_t>(), std::declval<B_t>()));
    using im_t = decltype(Qsub<typename S::subBCType>(std::declval<B_t>(), std::declval<C_t>()));
    using prod_t = Qcomplex<re_t, im_t>;
    static constexpr int cmul = QG_CMUL_TF;
    static void fill(qgemul_desc& d)
    {
        d.mul[QG_T_AB] = fmt_of<ab_t>(); d.mul[QG_T_CD] = fmt_of<cd_t>(); d.mul[QG_T_BA] = fmt_of<ba_t>();
        d.mul[QG_T_A] = fmt_of<A_t>(); d.mul[QG_T_B] = fmt_of<B_t>(); d.mul[QG_T_C] = fmt_of<C_t>();
        d.mul[QG_T_RE] = fmt_of<re_t>(); d.mul[QG_T_IM] = fmt_of<im_t>();
    }
};

// ---- tree levels: type of level l as Reducer selects it, formats of the add and of the buffer.
// The walk stops early once the buffer type repeats (all further levels are identical), so the
// template depth is bounded by the length of the level list, not by log2 K.
template <size_t L, class Prev, class List> struct level_sel;
template <size_t L, class Prev> struct level_sel<L, Prev, TypeList<>> { using tag = std::nullptr_t; using buf = Prev; };
template <size_t L, class Prev, class... Ls> struct level_sel<L, Prev, TypeList<Ls...>> {
    using tag = TypeAt<(L >= sizeof...(Ls) ? sizeof...(Ls) - 1 : L), TypeList<Ls...>>;
    using buf = tag;
};
template <size_t L, class Prev, class List>
void fill_levels(qgemul_desc& d)
{
    using sel = level_sel<L, Prev, List>;
    using add_t = decltype(Qadd<typename sel::tag>(std::declval<Prev>(), std::declval<Prev>()));
    using buf_t = typename sel::buf;
    constexpr size_t n_list = List::size;
    constexpr bool steady = std::is_same_v<buf_t, Prev> && (L + 1 >= n_list);
    const uint32_t last = steady ? d.n_levels : L + 1;
    for (uint32_t l = L; l < last && l < d.n_levels; ++l) {
        d.level_add[0][l] = fmt_of<typename parts<add_t>::re>();
        d.level_add[1][l] = fmt_of<typename parts<add_t>::im>();
        d.level[0][l] = fmt_of<typename parts<buf_t>::re>();
        d.level[1][l] = fmt_of<typename parts<buf_t>::im>();
    }
    if constexpr (!steady && L + 1 < QG_MAX_LEVELS) {
        if (L + 1 < d.n_levels) fill_levels<L + 1, buf_t, List>(d);
    }
}

} // namespace qgemul_detail

// lower a Qgemul call on reference tensor types to the C-ABI descriptor (no device access)
template <typename... Tags, size_t CM, size_t CN, size_t AR, size_t AC, size_t BK, size_t BN, class EC, class EA, class EB>
qgemul_desc Qgemul_lower_types(std::type_identity<Qu_s<dim<CM, CN>, EC>>, std::type_identity<Qu_s<dim<AR, AC>, EA>>,
                               std::type_identity<Qu_s<dim<BK, BN>, EB>>)
{
    using namespace qgemul_detail;
    constexpr bool ta = pick_ta<Tags...>::value;
    constexpr size_t M = ta ? AC : AR, K = ta ? AR : AC;
    static_assert(M == CM && BN == CN && BK == K, "Qgemul: C is MxN, A is MxK (KxM when transposed), B is KxN");
    static_assert(EA::is_complex == EB::is_complex && EA::is_complex == EC::is_complex, "Qgemul: all real or all complex");
    using mul_list = typename pick_mul<Tags...>::type;
    using add_list = typename pick_add<Tags...>::type;
    using S = slots<EA, EB, mul_list>;
    qgemul_desc d{};
    d.abi = QGEMUL_ABI_VERSION;
    d.transA = ta;
    d.is_complex = EA::is_complex;
    d.cmul = uint8_t(S::cmul);
    d.M = int64_t(M); d.N = int64_t(CN); d.K = int64_t(K);
    d.a[0] = fmt_of<typename parts<EA>::re>(); d.a[1] = fmt_of<typename parts<EA>::im>();
    d.b[0] = fmt_of<typename parts<EB>::re>(); d.b[1] = fmt_of<typename parts<EB>::im>();
    d.c[0] = fmt_of<typename parts<EC>::re>(); d.c[1] = fmt_of<typename parts<EC>::im>();
    S::fill(d);
    for (size_t k = K; k > 1; k = (k + 1) / 2) ++d.n_levels;
    if (d.n_levels) fill_levels<0, typename S::prod_t, add_list>(d);
    return d;
}

template <typename... Tags, class TC, class TA, class TB>
qgemul_desc Qgemul_lower(const TC&, const TA&, const TB&)
{
    return Qgemul_lower_types<Tags...>(std::type_identity<TC>{}, std::type_identity<TA>{}, std::type_identity<TB>{});
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

// the README entry point: C = A' * B with per-product and per-tree-node quantisation
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

// ---- element-wise operators after the GEMM (the header's lazy tensor operators, QuBLAS.h:3780-3877, :4079-4100) ----
//     Qgemul<…>(C, A, B);   Qu<dim<M,N>, T1> t = Qmul<t1…>(C, s);   Qu<dim<M,N>, DT> D = Qadd<t2…>(t, Bias);
// as ONE call that never stores C or t:
//     Qgemul<…, QgemulResult<CT>>(D, A, B, ThenMul<T1, t1…>(s), ThenAdd<void, t2…>(Bias));
// Every operator's result type is decltype of the header's own scalar Qmul / Qadd / Qsub<tags…> on the running element
// type and the operand's element type, so the chain resolves exactly as the three statements above do.
// After a COMPLEX Qgemul the same spelling takes complex operands for ThenAdd / ThenSub / ThenRsub (realT<…> / imagT<…>
// tags, QuBLAS.h:3549-3589) and real operands for all four (:3604-3707); the chain is lowered to the two part-wise
// chains of qgemul_epilogue_cplx.  complex x complex ThenMul does not compile (it mixes the parts).
template <class CT> struct QgemulResult {};

namespace qgemul_detail {
template <int OP, bool XFIRST, class Into, class Operand, typename... Tags>
struct EwStage {
    const Operand& e;
    static constexpr int op = OP;
    static constexpr bool x_first = XFIRST;
    static constexpr bool scalar = isScalar<Operand>;
    template <class O, bool = isScalar<O>> struct elem { using type = O; };
    template <class O> struct elem<O, false> { using type = typename O::elem_t; };
    using e_t = typename elem<Operand>::type;
    template <class X> static auto apply(const X& x, const e_t& y)
    {
        if constexpr (OP == QG_EW_MUL) { if constexpr (XFIRST) return Qmul<Tags...>(x, y); else return Qmul<Tags...>(y, x); }
        else if constexpr (OP == QG_EW_ADD) { if constexpr (XFIRST) return Qadd<Tags...>(x, y); else return Qadd<Tags...>(y, x); }
        else { if constexpr (XFIRST) return Qsub<Tags...>(x, y); else return Qsub<Tags...>(y, x); }
    }
    template <class X> using r_t = decltype(apply(std::declval<X>(), std::declval<e_t>()));
    template <class X> using next_t = std::conditional_t<std::is_void_v<Into>, r_t<X>, Into>;
};
template <class... Tags> struct pick_result { using type = void; };
template <class CT, class... Rest> struct pick_result<QgemulResult<CT>, Rest...> { using type = CT; };
template <class T, class... Rest> struct pick_result<T, Rest...> : pick_result<Rest...> {};

// complex chains: stage k of the chain of part P (0 = real parts, 1 = imaginary parts) — include/qgemul.h's table
template <class X> void fill_chain_cplx(qgemul_epilogue_cplx&, uint32_t) {}
template <class X, class S0, class... Ss>
void fill_chain_cplx(qgemul_epilogue_cplx& ep, uint32_t k)
{
    using e_t = typename S0::e_t;
    using r_t = typename S0::template r_t<X>;
    using n_t = typename S0::template next_t<X>;
    static_assert(X::is_complex && r_t::is_complex && n_t::is_complex, "a complex chain runs on complex tensors");
    static_assert(!(e_t::is_complex && S0::op == QG_EW_MUL), "complex x complex multiplication mixes the parts: not an element-wise stage");
    ep.e_complex[k] = e_t::is_complex;
    qgemul_ew_stage& re = ep.part[0].stage[k];
    qgemul_ew_stage& im = ep.part[1].stage[k];
    re.op = im.op = uint8_t(S0::op);
    re.x_first = im.x_first = S0::x_first;
    re.e_scalar = im.e_scalar = S0::scalar;
    re.e = fmt_of<typename parts<e_t>::re>();
    im.e = fmt_of<typename parts<e_t>::im>();
    re.r = fmt_of<typename r_t::realType>();
    im.r = fmt_of<typename r_t::imagType>();
    re.t = fmt_of<typename n_t::realType>();
    im.t = fmt_of<typename n_t::imagType>();
    if constexpr (!e_t::is_complex && S0::op != QG_EW_MUL) {
        if constexpr (S0::op == QG_EW_ADD || S0::x_first) im.op = QG_EW_PASS;   // the imaginary part is carried over (QuBLAS.h:3654, :3670, :3701)
        im.e_scalar = 1;                                                        // real - complex: the zero of the operand's type (:3686)
    }
    fill_chain_cplx<n_t, Ss...>(ep, k + 1);
}

template <class X> void fill_chain(qgemul_epilogue&, uint32_t) {}
template <class X, class S0, class... Ss>
void fill_chain(qgemul_epilogue& ep, uint32_t k)
{
    using r_t = typename S0::template r_t<X>;
    using n_t = typename S0::template next_t<X>;
    ep.stage[k].op = uint8_t(S0::op);
    ep.stage[k].x_first = S0::x_first;
    ep.stage[k].e_scalar = S0::scalar;
    ep.stage[k].e = fmt_of<typename S0::e_t>();
    ep.stage[k].r = fmt_of<r_t>();
    ep.stage[k].t = fmt_of<n_t>();
    fill_chain<n_t, Ss...>(ep, k + 1);
}
} // namespace qgemul_detail

template <class Into = void, typename... Tags, class Operand> auto ThenMul(const Operand& e) { return qgemul_detail::EwStage<QG_EW_MUL, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenAdd(const Operand& e) { return qgemul_detail::EwStage<QG_EW_ADD, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenSub(const Operand& e) { return qgemul_detail::EwStage<QG_EW_SUB, true, Into, Operand, Tags...>{e}; }
template <class Into = void, typename... Tags, class Operand> auto ThenRsub(const Operand& e) { return qgemul_detail::EwStage<QG_EW_SUB, false, Into, Operand, Tags...>{e}; }

template <typename... Tags, class TD, class... Stages>
qgemul_epilogue Qgemul_lower_epilogue(const TD&, const Stages&...)
{
    using CT = typename qgemul_detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>: the element type of the Qgemul result");
    static_assert(sizeof...(Stages) <= QG_MAX_EW, "at most QG_MAX_EW element-wise operators");
    static_assert(!CT::is_complex && !TD::elem_t::is_complex, "a complex chain is lowered by Qgemul_lower_epilogue_cplx");
    qgemul_epilogue ep{};
    ep.n_stages = sizeof...(Stages);
    qgemul_detail::fill_chain<CT, Stages...>(ep, 0);
    ep.d = qgemul_detail::fmt_of<typename TD::elem_t>();
    return ep;
}

template <typename... Tags, class TD, class... Stages>
qgemul_epilogue_cplx Qgemul_lower_epilogue_cplx(const TD&, const Stages&...)
{
    using CT = typename qgemul_detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>: the element type of the Qgemul result");
    static_assert(sizeof...(Stages) <= QG_MAX_EW, "at most QG_MAX_EW element-wise operators");
    static_assert(CT::is_complex && TD::elem_t::is_complex, "a complex chain runs from a complex Qgemul result into a complex tensor");
    qgemul_epilogue_cplx ep{};
    ep.part[0].n_stages = ep.part[1].n_stages = sizeof...(Stages);
    qgemul_detail::fill_chain_cplx<CT, Stages...>(ep, 0);
    ep.part[0].d = qgemul_detail::fmt_of<typename TD::elem_t::realType>();
    ep.part[1].d = qgemul_detail::fmt_of<typename TD::elem_t::imagType>();
    return ep;
}

template <typename... Tags, class TD, class TA, class TB, class S0, class... Stages>
void Qgemul(TD& D, const TA& A, const TB& B, const S0& s0, const Stages&... st)
{
    using CT = typename qgemul_detail::pick_result<Tags...>::type;
    static_assert(!std::is_void_v<CT>, "Qgemul with element-wise operators needs QgemulResult<CT>");
    qgemul_desc d = Qgemul_lower_types<Tags...>(std::type_identity<Qu_s<typename TD::size, CT>>{}, std::type_identity<TA>{}, std::type_identity<TB>{});
    d.flags |= QgemulDescFlags();
    auto ptr = [](const auto& stage) -> const void* {
        // a scalar is one element as the tensors store them (ArbiInt<N>::data, QuBLAS.h:353; {real, imag} for a complex one, :2512-2513)
        if constexpr (std::remove_cvref_t<decltype(stage)>::scalar) return &stage.e;
        else return stage.e.data.data();
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

} // namespace QuBLAS
