// ref_cases_real.cpp — TEST INFRASTRUCTURE: real-valued golden GEMM cases evaluated with the
// reference header's own Qmul / Qreduce / converting constructor (see ref_driver.hpp).
// Usage: ref_cases_real <part> > out.jsonl   (part selects a slice so the slices build in parallel)
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

// element types of the BASELINE.json configurations
using e88z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>; // c1, c3
using e43 = Qu<intBits<4>, fracBits<3>>;                                                        // c2, c4
using u44 = Qu<intBits<4>, fracBits<4>, isSigned<false>>;
using n63 = Qu<intBits<6>, fracBits<-3>>;
using w16 = Qu<intBits<16>, fracBits<3>>; // wide C so that parity is not hidden by saturation

// class-L tags of SURVEY.md §8-d
using L2mul = TypeList<intBits<9>, fracBits<6>>;
using L2add = TypeList<Qu<intBits<19>, fracBits<6>>>;
using L4add = TypeList<Qu<intBits<21>, fracBits<6>>>;
using L3mul = TypeList<intBits<17>, fracBits<16>>;
using L3add = TypeList<Qu<intBits<29>, fracBits<16>>>;

static Inputs one_to_sixteen()
{
    Inputs in;
    in.synthetic = false;
    for (int v = 1; v <= 16; ++v) { in.A.push_back(v * 256); in.B.push_back(v * 256); }
    return in;
}
static Inputs syn(int dist, uint64_t sa = 1, uint64_t sb = 2)
{
    Inputs in;
    in.dist = dist; in.seedA = sa; in.seedB = sb;
    return in;
}

// every QuMode x OfMode on C (the GEMM epilogue = converting constructor)
template <class Q, class O>
using c43 = Qu<intBits<4>, fracBits<3>, QuMode<Q>, OfMode<O>>;

template <class O>
static void epilogue_row(const char* oname, FILE* out)
{
    auto nm = [&](const char* q) { static std::string s; s = std::string("epi_e43_L_8x8x64_") + q + "_" + oname; return s.c_str(); };
    run_case<e43, e43, c43<RND::POS_INF, O>, L2mul, L2add, false, 8, 8, 64>(nm("POS_INF"), syn(0), out);
    run_case<e43, e43, c43<RND::NEG_INF, O>, L2mul, L2add, false, 8, 8, 64>(nm("NEG_INF"), syn(0), out);
    run_case<e43, e43, c43<RND::ZERO, O>, L2mul, L2add, false, 8, 8, 64>(nm("ZERO"), syn(0), out);
    run_case<e43, e43, c43<RND::INF, O>, L2mul, L2add, false, 8, 8, 64>(nm("INF"), syn(0), out);
    run_case<e43, e43, c43<RND::CONV, O>, L2mul, L2add, false, 8, 8, 64>(nm("CONV"), syn(0), out);
    run_case<e43, e43, c43<TRN::TCPL, O>, L2mul, L2add, false, 8, 8, 64>(nm("TCPL"), syn(0), out);
    run_case<e43, e43, c43<TRN::SMGN, O>, L2mul, L2add, false, 8, 8, 64>(nm("SMGN"), syn(0), out);
    // the same with small inputs so that results are not all saturated
    run_case<e43, e43, c43<RND::CONV, O>, L2mul, L2add, false, 8, 8, 64>(nm("CONV_small"), syn(1), out);
    run_case<e43, e43, c43<TRN::SMGN, O>, L2mul, L2add, false, 8, 8, 64>(nm("SMGN_small"), syn(1), out);
}

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    switch (part) {
    case 0: // configuration 1 (README example shapes) and its class-L sibling
        run_case<e88z, e88z, e88z, TypeList<e88z>, TypeList<e88z>, false, 4, 4, 4>("c1_nn_classT", one_to_sixteen(), out);
        run_case<e88z, e88z, e88z, TypeList<e88z>, TypeList<e88z>, true, 4, 4, 4>("c1_tn_classT", one_to_sixteen(), out);
        run_case<e88z, e88z, e88z, TypeList<>, TypeList<>, false, 4, 4, 4>("c1_nn_default", one_to_sixteen(), out);
        run_case<e88z, e88z, e88z, L3mul, L3add, false, 4, 4, 4>("c1_nn_classL", one_to_sixteen(), out);
        run_case<e88z, e88z, e88z, L3mul, L3add, true, 4, 4, 4>("c1_tn_classL", one_to_sixteen(), out);
        run_case<e88z, e88z, e88z, TypeList<>, TypeList<>, false, 16, 16, 64>("e88z_default_16x16x64_full", syn(0), out);
        run_case<e88z, e88z, e88z, TypeList<>, TypeList<>, true, 16, 16, 64>("e88z_default_tn_16x16x64_small", syn(1), out);
        run_case<e88z, e88z, e88z, L3mul, L3add, false, 16, 16, 64>("e88z_L_16x16x64_full", syn(0), out);
        run_case<e88z, e88z, Qu<intBits<20>, fracBits<10>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, L3mul, L3add, true, 16, 16, 64>(
            "e88z_L_tn_16x16x64_wideC", syn(0), out);
        break;
    case 1: // configuration 2 formats, tree and linear, ragged shape
        run_case<e43, e43, e43, TypeList<>, TypeList<>, false, 33, 17, 128>("e43_default_33x17x128_full", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<>, false, 33, 17, 128>("e43_default_33x17x128_small_wideC", syn(1), out);
        run_case<e43, e43, w16, L2mul, L2add, false, 33, 17, 128>("e43_L_33x17x128_full_wideC", syn(0), out);
        run_case<e43, e43, e43, L2mul, L2add, true, 33, 17, 128>("e43_L_tn_33x17x128_full", syn(0), out);
        run_case<e43, e43, w16, L2mul, L2add, false, 16, 16, 512>("e43_L_16x16x512_full_wideC", syn(0), out);
        break;
    case 2: // epilogue modes (SAT::TCPL, SAT::ZERO)
        epilogue_row<SAT::TCPL>("SAT_TCPL", out);
        epilogue_row<SAT::ZERO>("SAT_ZERO", out);
        break;
    case 3: // epilogue modes (SAT::SMGN, WRP::TCPL)
        epilogue_row<SAT::SMGN>("SAT_SMGN", out);
        epilogue_row<WRP::TCPL>("WRP_TCPL", out);
        break;
    case 4: // long reductions, blocked 512-leaf subtrees
        run_case<e43, e43, w16, L2mul, L2add, false, 8, 8, 1024>("e43_L_8x8x1024_full_wideC", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<>, false, 8, 8, 1024>("e43_default_8x8x1024_small_wideC", syn(1), out);
        run_case<e43, e43, e43, L2mul, L4add, true, 4, 4, 4096>("e43_L_tn_4x4x4096_full", syn(0), out);
        run_case<e88z, e88z, e88z, TypeList<>, TypeList<>, false, 4, 4, 4096>("e88z_default_4x4x4096_small", syn(1), out);
        run_case<e88z, e88z, e88z, L3mul, L3add, false, 4, 4, 4096>("e88z_L_4x4x4096_full", syn(0), out);
        run_case<e88z, e88z, Qu<intBits<24>, fracBits<8>>, L3mul, L3add, true, 4, 4, 4096>("e88z_L_tn_4x4x4096_full_wideC", syn(0), out);
        break;
    case 5: { // unsigned, negative fracBits, mixed operand formats, wrapping levels
        run_case<u44, u44, u44, TypeList<>, TypeList<>, false, 8, 8, 64>("u44_default_8x8x64_full", syn(0), out);
        run_case<u44, u44, Qu<intBits<14>, fracBits<4>, isSigned<false>>, TypeList<>, TypeList<>, false, 8, 8, 64>("u44_default_8x8x64_small_wideC", syn(1), out);
        run_case<u44, u44, Qu<intBits<14>, fracBits<8>, isSigned<false>>, TypeList<intBits<8>, fracBits<8>>, TypeList<Qu<intBits<14>, fracBits<8>, isSigned<false>>>, false, 8, 8,
                 64>("u44_L_8x8x64_full", syn(0), out);
        run_case<n63, n63, n63, TypeList<>, TypeList<>, false, 8, 8, 64>("n63_default_8x8x64_full", syn(0), out);
        run_case<n63, n63, Qu<intBits<16>, fracBits<-3>>, TypeList<FullPrec>, TypeList<Qu<intBits<20>, fracBits<-6>>>, true, 8, 8, 64>("n63_fullprec_tn_8x8x64_full", syn(0), out);
        run_case<e88z, e43, w16, TypeList<>, TypeList<>, false, 8, 8, 64>("mixed_e88z_e43_default_8x8x64_small", syn(1), out);
        run_case<e43, u44, w16, TypeList<>, TypeList<>, false, 8, 8, 64>("mixed_e43_u44_default_8x8x64_full", syn(0), out);
        using wl = Qu<intBits<5>, fracBits<3>, OfMode<WRP::TCPL>>;
        run_case<e43, e43, w16, TypeList<wl>, TypeList<wl>, false, 8, 8, 64>("e43_wrap_levels_8x8x64_full", syn(0), out);
        using ul = Qu<intBits<6>, fracBits<4>, isSigned<false>, OfMode<WRP::TCPL>>;
        run_case<e43, e43, w16, TypeList<ul>, TypeList<ul>, false, 8, 8, 64>("e43_unsigned_wrap_levels_8x8x64_full", syn(0), out);
        break;
    }
    case 6: { // per-level type lists with differing formats and modes; rounding products
        using t1 = Qu<intBits<6>, fracBits<5>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using t2 = Qu<intBits<8>, fracBits<4>, QuMode<RND::ZERO>, OfMode<SAT::TCPL>>;
        using t3 = Qu<intBits<9>, fracBits<2>, QuMode<TRN::SMGN>, OfMode<SAT::ZERO>>;
        using pm = Qu<intBits<5>, fracBits<4>, QuMode<RND::INF>, OfMode<SAT::TCPL>>;
        run_case<e43, e43, w16, TypeList<pm>, TypeList<t1, t2, t3>, false, 8, 8, 64>("e43_levels3_8x8x64_full", syn(0), out);
        run_case<e43, e43, w16, TypeList<pm>, TypeList<t1, t2, t3>, true, 8, 8, 64>("e43_levels3_tn_8x8x64_small", syn(1), out);
        run_case<e43, e43, e43, TypeList<QuMode<RND::POS_INF>, fracBits<2>>, TypeList<t2, t1>, false, 8, 8, 16>("e43_levels2_8x8x16_full", syn(0), out);
        run_case<e88z, e88z, e88z, TypeList<QuMode<RND::NEG_INF>, OfMode<SAT::SMGN>, intBits<10>>, TypeList<Qu<intBits<12>, fracBits<6>, QuMode<RND::POS_INF>>>, false, 8, 8,
                 32>("e88z_rnd_levels1_8x8x32_small", syn(1), out);
        // README shapes: type1 / type2 list
        using type1 = Qu<isSigned<true>, intBits<6>, fracBits<3>, OfMode<SAT::ZERO>>;
        using type2 = Qu<intBits<6>, fracBits<-3>>;
        run_case<type1, type1, type1, TypeList<type1>, TypeList<type1, type2>, true, 4, 4, 4>("readme_list_tn_4x4x4", syn(1), out);
        run_case<type1, type1, type1, TypeList<type1>, TypeList<type1, type2>, false, 8, 8, 32>("readme_list_8x8x32_small", syn(1), out);
        break;
    }
    case 7: { // reduction lengths that are not powers of two follow the vector overload (QuBLAS.h:4977-4980)
        using t1 = Qu<intBits<6>, fracBits<5>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using t2 = Qu<intBits<8>, fracBits<4>, QuMode<RND::ZERO>, OfMode<SAT::TCPL>>;
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 1>("e43_K1", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 2>("e43_K2", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 3>("e43_K3", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 5>("e43_K5", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 6>("e43_K6", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1, t2>, false, 4, 4, 7>("e43_K7", syn(0), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<t1>, true, 4, 4, 100>("e43_K100_tn", syn(1), out);
        run_case<e43, e43, w16, TypeList<>, TypeList<>, false, 3, 5, 37>("e43_K37_default", syn(1), out);
        run_case<e43, e43, w16, L2mul, L2add, false, 4, 4, 1000>("e43_L_K1000", syn(0), out);
        run_case<e88z, e88z, e88z, TypeList<>, TypeList<t2, t1>, false, 2, 2, 1000>("e88z_K1000_levels2_small", syn(1), out);
        break;
    }
    case 8: {   // C with WRP::TCPL_SAT<N> (the stub: the root goes into C's storage word unclamped)
        using csat = Qu<intBits<4>, fracBits<3>, OfMode<WRP::TCPL_SAT<2>>>;
        using csat2 = Qu<intBits<6>, fracBits<2>, QuMode<RND::CONV>, OfMode<WRP::TCPL_SAT<1>>>;
        run_case<e43, e43, csat, L2mul, L2add, false, 8, 8, 64>("e43_L_8x8x64_tcplsatC", syn(0), out);
        run_case<e43, e43, csat2, TypeList<>, TypeList<w16>, true, 9, 7, 37>("e43_T_tn_9x7x37_tcplsatC", syn(0), out);
        run_case<e88z, e88z, Qu<intBits<8>, fracBits<8>, OfMode<WRP::TCPL_SAT<4>>>, L3mul, L3add, false, 6, 6, 256>("e88z_L_6x6x256_tcplsatC", syn(0), out);
        break;
    }
    default:
        return 2;
    }
    return 0;
}
