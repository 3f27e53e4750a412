// ref_cases_cplx.cpp — TEST INFRASTRUCTURE: complex golden GEMM cases (BASELINE.json configuration 5
// formats) evaluated with the reference header's BasicComplexMul / TFComplexMul, complex Qadd tree
// nodes and part-wise converting constructor (see ref_driver.hpp).
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

// configuration 5: Qcomplex<int<6,3>, int<6,-3>>, "RND + SAT"
using r63 = Qu<intBits<6>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using i63n = Qu<intBits<6>, fracBits<-3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using c5 = Qcomplex<r63, i63n>;
// a complex type with headroom so results are not all saturated
using rw = Qu<intBits<18>, fracBits<6>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using cw = Qcomplex<rw, rw>;
// symmetrical small complex operands
using r55 = Qu<intBits<5>, fracBits<5>>;
using c55 = Qcomplex<r55, r55>;
using t146 = Qu<intBits<14>, fracBits<6>>;
using c146 = Qcomplex<t146, t146>;

static Inputs syn(int dist, uint64_t sa = 11, uint64_t sb = 12)
{
    Inputs in;
    in.dist = dist; in.seedA = sa; in.seedB = sb;
    return in;
}

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    switch (part) {
    case 0: { // default sub-op tags, Basic (the default multiplier, QuBLAS.h:3422-3424) and TF
        run_case<c5, c5, c5, TypeList<>, TypeList<>, false, 8, 8, 64>("c5_basic_default_8x8x64_full", syn(0), out);
        run_case<c5, c5, cw, TypeList<>, TypeList<>, false, 8, 8, 64>("c5_basic_default_8x8x64_small_wideC", syn(1), out);
        run_case<c5, c5, c5, TypeList<TFComplexMul<>>, TypeList<>, false, 8, 8, 64>("c5_tf_default_8x8x64_full", syn(0), out);
        run_case<c5, c5, cw, TypeList<TFComplexMul<>>, TypeList<>, true, 8, 8, 64>("c5_tf_default_tn_8x8x64_small_wideC", syn(1), out);
        run_case<c55, c55, cw, TypeList<TFComplexMul<>>, TypeList<>, false, 8, 8, 64>("c55_tf_default_8x8x64_full_wideC", syn(0), out);
        run_case<c55, c55, cw, TypeList<BasicComplexMul<>>, TypeList<>, false, 8, 8, 64>("c55_basic_default_8x8x64_full_wideC", syn(0), out);
        break;
    }
    case 1: { // sub-op tags, the two TF quirks, complex level types
        // quirk input of SURVEY.md §8-a11: (-60+56i)(1+i) with every tag int<14,6>
        Inputs q;
        q.synthetic = false;
        q.A = {-60 * 64, 56 * 64};
        q.B = {1 * 64, 1 * 64};
        using TFall = TFComplexMul<abT<t146>, cdT<t146>, baT<t146>, abcT<t146>, cdbT<t146>, badT<t146>, ABT<t146>, BCT<t146>>;
        run_case<c146, c146, c146, TypeList<TFall>, TypeList<>, false, 1, 1, 1>("tf_quirk_baT_1x1x1", q, out);
        using Ball = BasicComplexMul<acT<t146>, bdT<t146>, adT<t146>, bcT<t146>, acbdT<t146>, adbcT<t146>>;
        run_case<c146, c146, c146, TypeList<Ball>, TypeList<>, false, 1, 1, 1>("basic_exact_1x1x1", q, out);
        // distinct tags per sub-op: exposes the crossed cdbT/badT use (QuBLAS.h:3525-3526)
        using tA = Qu<intBits<9>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using tB = Qu<intBits<7>, fracBits<2>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>;
        using tC = Qu<intBits<10>, fracBits<5>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
        using tD = Qu<intBits<8>, fracBits<3>, QuMode<RND::INF>, OfMode<SAT::TCPL>>;
        using TFmix = TFComplexMul<abT<tA>, cdT<tD>, abcT<tC>, cdbT<tB>, badT<tA>, ABT<tD>, BCT<tC>>;
        run_case<c55, c55, cw, TypeList<TFmix>, TypeList<>, false, 8, 8, 16>("c55_tf_mixedtags_8x8x16_full", syn(0), out);
        run_case<c55, c5, cw, TypeList<TFmix>, TypeList<>, true, 8, 8, 16>("c55_c5_tf_mixedtags_tn_8x8x16_small", syn(1), out);
        using Bmix = BasicComplexMul<acT<tA>, bdT<tB>, adT<tC>, bcT<tD>, acbdT<tC>, adbcT<tA>>;
        run_case<c55, c55, cw, TypeList<Bmix>, TypeList<>, false, 8, 8, 16>("c55_basic_mixedtags_8x8x16_full", syn(0), out);
        // loose tags inside the wrapper apply to every sub-op that has no tag of its own
        run_case<c55, c55, cw, TypeList<BasicComplexMul<intBits<12>, OfMode<SAT::ZERO>, bdT<tB>>>, TypeList<>, false, 4, 4, 8>("c55_basic_loosetags_4x4x8_full", syn(0), out);
        // complex level types: the add itself is a default merge, the level buffer converts (QuBLAS.h:4966)
        using l1 = Qcomplex<Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, Qu<intBits<11>, fracBits<6>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>>;
        using l2 = Qcomplex<Qu<intBits<16>, fracBits<2>, QuMode<RND::ZERO>>, Qu<intBits<16>, fracBits<3>, QuMode<RND::INF>, OfMode<WRP::TCPL>>>;
        run_case<c55, c55, cw, TypeList<TFComplexMul<>>, TypeList<l1, l2>, false, 8, 8, 32>("c55_tf_levels2_8x8x32_full", syn(0), out);
        run_case<c55, c55, cw, TypeList<>, TypeList<l1>, false, 4, 4, 7>("c55_basic_levels1_K7_small", syn(1), out);
        break;
    }
    case 2: { // long reduction (configuration 5's K) and a linear-class variant
        run_case<c5, c5, c5, TypeList<TFComplexMul<>>, TypeList<>, false, 2, 2, 2048>("c5_tf_default_2x2x2048_small", syn(1), out);
        using m = Qu<intBits<16>, fracBits<6>>;
        using s = Qu<intBits<8>, fracBits<3>>;
        using TFL = TFComplexMul<abT<s>, cdT<s>, abcT<m>, cdbT<m>, badT<m>, ABT<Qu<intBits<17>, fracBits<6>>>, BCT<Qu<intBits<17>, fracBits<6>>>>;
        using lw = Qcomplex<Qu<intBits<30>, fracBits<6>>, Qu<intBits<30>, fracBits<6>>>;
        run_case<c5, c5, cw, TypeList<TFL>, TypeList<lw>, false, 2, 2, 2048>("c5_tf_L_2x2x2048_full_wideC", syn(0), out);
        run_case<c5, c5, c5, TypeList<TFL>, TypeList<lw>, true, 4, 4, 64>("c5_tf_L_tn_4x4x64_full", syn(0), out);
        break;
    }
    case 3: { // complex LINEAR class: BasicComplexMul with exact sub-ops and headroom in every level.
              // (TFComplexMul can never be linear: its (b-a) is always formed in the default-merged format,
              //  QuBLAS.h:3515, whose intBits = max(Ia, Ib) cannot hold b-a over the operands' full range.)
        using BL = BasicComplexMul<acT<Qu<intBits<14>, fracBits<6>>>, bdT<Qu<intBits<14>, fracBits<-6>>>, adT<Qu<intBits<14>, fracBits<0>>>,
                                   bcT<Qu<intBits<14>, fracBits<0>>>, acbdT<Qu<intBits<15>, fracBits<6>>>, adbcT<Qu<intBits<15>, fracBits<0>>>>;
        using lw = Qcomplex<Qu<intBits<30>, fracBits<6>>, Qu<intBits<30>, fracBits<0>>>;
        using cn = Qcomplex<Qu<intBits<9>, fracBits<2>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, Qu<intBits<7>, fracBits<-1>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>>;
        run_case<c5, c5, cw, TypeList<BL>, TypeList<lw>, false, 8, 8, 64>("c5_basic_L_8x8x64_full_wideC", syn(0), out);
        run_case<c5, c5, cn, TypeList<BL>, TypeList<lw>, true, 5, 7, 128>("c5_basic_L_tn_5x7x128_full_narrowC", syn(0), out);
        run_case<c5, c5, c5, TypeList<BL>, TypeList<lw>, false, 2, 2, 2048>("c5_basic_L_2x2x2048_full", syn(0), out);
        run_case<c5, c5, cw, TypeList<BL>, TypeList<lw>, false, 3, 3, 100>("c5_basic_L_K100_small_wideC", syn(1), out);
        break;
    }
    default:
        return 2;
    }
    return 0;
}
