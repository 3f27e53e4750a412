// Compiles include/QuBLAS_amd.h on its own (no reference header) and prints the descriptors it
// lowers for the same tag combinations as ref_binding_probe.cpp (names match tests/golden).
#include "QuBLAS_amd.h"
#include "desc_json.hpp"

using namespace QuBLAS_amd;

using e88z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
using e43 = Qu<intBits<4>, fracBits<3>>;
using w16 = Qu<intBits<16>, fracBits<3>>;

template <class EC, class EA, class EB, size_t M, size_t N, size_t K, bool TA, class... Tags>
void probe(const char* name)
{
    Qu<dim<M, N>, EC> C;
    std::conditional_t<TA, Qu<dim<K, M>, EA>, Qu<dim<M, K>, EA>> A;
    Qu<dim<K, N>, EB> B;
    print_desc(name, Qgemul_lower<Tags...>(C, A, B));
}

int main()
{
    static_assert(sizeof(e43) == 4 && sizeof(Qu<intBits<30>, fracBits<30>>) == 8, "host element layout");
    static_assert(std::is_same_v<Qu<fracBits<3>, intBits<4>>, e43>, "tags are order-free");
    probe<e88z, e88z, e88z, 4, 4, 4, false, QgemulAddArgs<e88z>, QgemulMulArgs<e88z>>("c1_nn_classT");
    probe<e88z, e88z, e88z, 4, 4, 4, true, QgemulTransposedA<true>, QgemulMulArgs<e88z>, QgemulAddArgs<TypeList<e88z>>>("c1_tn_classT");
    probe<e88z, e88z, e88z, 4, 4, 4, false>("c1_nn_default");
    probe<e88z, e88z, e88z, 4, 4, 4, false, QgemulMulArgs<intBits<17>, fracBits<16>>, QgemulAddArgs<Qu<intBits<29>, fracBits<16>>>>("c1_nn_classL");
    probe<w16, e43, e43, 33, 17, 128, false, QgemulMulArgs<intBits<9>, fracBits<6>>, QgemulAddArgs<Qu<intBits<19>, fracBits<6>>>>("e43_L_33x17x128_full_wideC");
    {
        using n63 = Qu<intBits<6>, fracBits<-3>>;
        probe<Qu<intBits<16>, fracBits<-3>>, n63, n63, 8, 8, 64, true, QgemulTransposedA<true>, QgemulMulArgs<FullPrec>, QgemulAddArgs<Qu<intBits<20>, fracBits<-6>>>>(
            "n63_fullprec_tn_8x8x64_full");
        using u44 = Qu<intBits<4>, fracBits<4>, isSigned<false>>;
        probe<w16, e43, u44, 8, 8, 64, false>("mixed_e43_u44_default_8x8x64_full");
        probe<w16, e88z, e43, 8, 8, 64, false>("mixed_e88z_e43_default_8x8x64_small");
    }
    {
        using t1 = Qu<intBits<6>, fracBits<5>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using t2 = Qu<intBits<8>, fracBits<4>, QuMode<RND::ZERO>, OfMode<SAT::TCPL>>;
        using t3 = Qu<intBits<9>, fracBits<2>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>;
        using pm = Qu<intBits<5>, fracBits<4>, QuMode<RND::INF>, OfMode<SAT::TCPL>>;
        probe<w16, e43, e43, 8, 8, 64, false, QgemulMulArgs<pm>, QgemulAddArgs<TypeList<t1, t2, t3>>>("e43_levels3_8x8x64_full");
        probe<e43, e43, e43, 8, 8, 16, false, QgemulMulArgs<QuMode<RND::POS_INF>, fracBits<2>>, QgemulAddArgs<t2, t1>>("e43_levels2_8x8x16_full");
        probe<w16, e43, e43, 4, 4, 5, false, QgemulAddArgs<t1, t2>>("e43_K5");
        using type1 = Qu<isSigned<true>, intBits<6>, fracBits<3>, OfMode<SAT::ZERO>>;
        using type2 = Qu<intBits<6>, fracBits<-3>>;
        using list = TypeList<type1, type2>;
        probe<type1, type1, type1, 4, 4, 4, true, QgemulAddArgs<list>, QgemulMulArgs<type1>, QgemulTransposedA<true>>("readme_list_tn_4x4x4");
    }
    {
        using r55 = Qu<intBits<5>, fracBits<5>>;
        using c55 = Qcomplex<r55, r55>;
        using rw = Qu<intBits<18>, fracBits<6>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
        using cw = Qcomplex<rw, rw>;
        using r63 = Qu<intBits<6>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
        using i63n = Qu<intBits<6>, fracBits<-3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
        using c5 = Qcomplex<r63, i63n>;
        using tA = Qu<intBits<9>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using tB = Qu<intBits<7>, fracBits<2>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>;
        using tC = Qu<intBits<10>, fracBits<5>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
        using tD = Qu<intBits<8>, fracBits<3>, QuMode<RND::INF>, OfMode<SAT::TCPL>>;
        using TFmix = TFComplexMul<abT<tA>, cdT<tD>, abcT<tC>, cdbT<tB>, badT<tA>, ABT<tD>, BCT<tC>>;
        probe<cw, c55, c55, 8, 8, 16, false, QgemulMulArgs<TFmix>>("c55_tf_mixedtags_8x8x16_full");
        probe<cw, c55, c5, 8, 8, 16, true, QgemulMulArgs<TFmix>, QgemulTransposedA<true>>("c55_c5_tf_mixedtags_tn_8x8x16_small");
        using Bmix = BasicComplexMul<acT<tA>, bdT<tB>, adT<tC>, bcT<tD>, acbdT<tC>, adbcT<tA>>;
        probe<cw, c55, c55, 8, 8, 16, false, QgemulMulArgs<Bmix>>("c55_basic_mixedtags_8x8x16_full");
        using l1 = Qcomplex<Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, Qu<intBits<11>, fracBits<6>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>>;
        using l2 = Qcomplex<Qu<intBits<16>, fracBits<2>, QuMode<RND::ZERO>>, Qu<intBits<16>, fracBits<3>, QuMode<RND::INF>, OfMode<WRP::TCPL>>>;
        probe<cw, c55, c55, 8, 8, 32, false, QgemulMulArgs<TFComplexMul<>>, QgemulAddArgs<l1, l2>>("c55_tf_levels2_8x8x32_full");
        probe<cw, c55, c55, 4, 4, 8, false, QgemulMulArgs<BasicComplexMul<intBits<12>, OfMode<SAT::ZERO>, bdT<tB>>>>("c55_basic_loosetags_4x4x8_full");
        probe<cw, c55, c55, 8, 8, 64, false>("c55_basic_default_8x8x64_full_wideC");
        probe<c5, c5, c5, 8, 8, 64, false, QgemulMulArgs<TFComplexMul<>>>("c5_tf_default_8x8x64_full");
        using t146 = Qu<intBits<14>, fracBits<6>>;
        using c146 = Qcomplex<t146, t146>;
        using TFall = TFComplexMul<abT<t146>, cdT<t146>, baT<t146>, abcT<t146>, cdbT<t146>, badT<t146>, ABT<t146>, BCT<t146>>;
        probe<c146, c146, c146, 1, 1, 1, false, QgemulMulArgs<TFall>>("tf_quirk_baT_1x1x1");
    }
    {   // Qreduce lowers onto the Qgemul path: result type follows the reducer's rule (compile-time only here)
        using type1 = Qu<isSigned<true>, intBits<6>, fracBits<3>, OfMode<SAT::ZERO>>;
        using type2 = Qu<intBits<6>, fracBits<-3>>;
        using R1 = detail::reduce_result<e43, 4, TypeList<type2>>::type;
        static_assert(std::is_same_v<R1, type2>);
        using R2 = detail::reduce_result<e43, 16, TypeList<type1, type2>>::type;
        static_assert(std::is_same_v<R2, type2>);
        using R3 = detail::reduce_result<e43, 2, TypeList<type1, type2>>::type;
        static_assert(std::is_same_v<R3, type1>);
        using R4 = detail::reduce_result<e43, 1, TypeList<type1>>::type;
        static_assert(std::is_same_v<R4, e43>);
        using R5 = detail::reduce_result<e43, 1000, TypeList<>>::type;
        static_assert(std::is_same_v<R5, e43>);
    }
    // value construction honours the type's modes (Qu_s(double), QuBLAS.h:2387-2393): 20 -> 16 in int<6,-3>
    Qu<intBits<6>, fracBits<-3>> q2 = 20;
    if (q2.toDouble() != 16.0) return 1;
    Qu<dim<4, 4>, e43> m1 = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 15.5};
    if (m1[1, 2].toDouble() != 10.0) return 2;  // column-major, like the reference
    {   // element-wise chains: the operators of three golden cases (tests/golden/ref_eltwise_*)
        using c238 = Qu<intBits<23>, fracBits<8>>;
        using b106 = Qu<intBits<10>, fracBits<6>>;
        using s34 = Qu<intBits<3>, fracBits<4>>;
        using d124 = Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using d88z = Qu<intBits<8>, fracBits<8>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
        using d62w = Qu<intBits<6>, fracBits<2>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
        Qu<dim<4, 4>, d124> D;
        Qu<dim<4, 4>, b106> Bias;
        s34 s;
        print_epilogue("scale_then_bias", Qgemul_lower_epilogue<QgemulResult<c238>>(D, ThenMul<Qu<intBits<24>, fracBits<8>>, intBits<24>, fracBits<8>>(s), ThenAdd<>(Bias)));
        print_epilogue("scale_into_narrow_then_bias", Qgemul_lower_epilogue<QgemulResult<c238>>(D, ThenMul<d88z>(s), ThenAdd<void, d124>(Bias)));
        Qu<dim<4, 4>, d62w> D2;
        print_epilogue("sub_efirst_tags_wrap", Qgemul_lower_epilogue<QgemulResult<c238>>(D2, ThenRsub<void, fracBits<2>, QuMode<RND::CONV>>(Bias)));
    }
    {   // complex chains: the operators of five golden cases (tests/golden/ref_cplx_eltwise_*)
        using r206 = Qu<intBits<20>, fracBits<6>>;
        using r54 = Qu<intBits<5>, fracBits<4>>;
        using r32 = Qu<intBits<3>, fracBits<2>>;
        using s22 = Qu<intBits<2>, fracBits<2>>;
        using r104 = Qu<intBits<10>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using r82z = Qu<intBits<8>, fracBits<2>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
        using r73w = Qu<intBits<7>, fracBits<3>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
        using r91s = Qu<intBits<9>, fracBits<1>, QuMode<TRN::SMGN>, OfMode<SAT::SMGN>>;
        using cw = Qcomplex<r206, r206>;
        using cb = Qcomplex<r54, r32>;
        using cd = Qcomplex<r104, r82z>;
        using cq = Qcomplex<r73w, r91s>;
        Qu<dim<4, 4>, cd> D;
        Qu<dim<4, 4>, cq> Dq;
        Qu<dim<4, 4>, cb> Bias;
        Qu<dim<4, 4>, r32> Off;
        cb cs;
        s22 s;
        r54 rs;
        print_epilogue_cplx("cadd_realT_imagT", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(D, ThenAdd<void, realT<r104>, imagT<intBits<12>, OfMode<SAT::ZERO>>>(Bias)));
        print_epilogue_cplx("csub_efirst_scalar_two_type_form", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(D, ThenRsub<void, r104, r82z>(cs)));
        print_epilogue_cplx("cmul_real_scalar_realT_imagT", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(D, ThenMul<void, realT<intBits<22>, fracBits<6>>, imagT<r104>>(s)));
        print_epilogue_cplx("real_scalar_minus_complex_fulltag", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(Dq, ThenRsub<void, r104>(rs)));
        print_epilogue_cplx("scale_cbias_real_sub", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(Dq, ThenMul<cw>(s), ThenAdd<cd, realT<r104>>(Bias), ThenSub<>(Off)));
        print_epilogue_cplx("cbias_then_real_minus", Qgemul_lower_epilogue_cplx<QgemulResult<cw>>(Dq, ThenAdd<cw>(Bias), ThenRsub<>(rs)));
    }
    return 0;
}
