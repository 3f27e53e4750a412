// ref_binding_demo.cpp — TEST INFRASTRUCTURE: the reference's OWN header and types driving the MI355X
// engine through include/qgemul_reference_binding.hpp.  Built in the build container (it needs
// /root/reference/include) into oracle/_ref/; the binary travels to the GPU box, where
// tests/test_gpu_cpp_dropin.py runs it and compares the printed C matrices with the golden vectors the
// reference's own Qmul/Qreduce produced (tests/golden).  Operand values come from ref_driver.hpp's copy
// of the synthetic generator, exactly as the golden records were generated.
#include "QuBLAS.h"
#include "qgemul_reference_binding.hpp"
#include "ref_driver.hpp"

#include <cstdio>

using namespace QuBLAS;

template <class T>
static void print_matrix(const char* name, const T& m, size_t n)
{
    std::printf("{\"name\":\"%s\",\"C\":[", name);
    for (size_t e = 0; e < n; ++e) std::printf("%s%lld", e ? "," : "", (long long)m.data[e].data.data);
    std::printf("]}\n");
}

template <class T>
static void fill_synth(T& m, size_t n, uint64_t seed, int dist)
{
    using E = typename T::elem_t;
    for (size_t e = 0; e < n; ++e) m.data[e].data.data = refdrv::synth<E>(seed, dist, e, 0);
}

int main()
{
    try {
        {   // configuration 1: 4x4x4 int<8,8> TCPL / SAT::ZERO, values 1..16, NN and TN (SURVEY.md §8-a known answer)
            using e88z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
            using mat = Qu<dim<4, 4>, e88z>;
            mat m1 = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0}, m3;
            Qgemul<QgemulAddArgs<e88z>, QgemulMulArgs<e88z>>(m3, m1, m1);
            print_matrix("c1_nn_classT", m3, 16);
            Qgemul<QgemulAddArgs<e88z>, QgemulMulArgs<e88z>, QgemulTransposedA<true>>(m3, m1, m1);
            print_matrix("c1_tn_classT", m3, 16);
        }
        {   // the README call, verbatim tags (readme.md:84-87), inputs as in the golden record
            using type1 = Qu<isSigned<true>, intBits<6>, fracBits<3>, OfMode<SAT::ZERO>>;
            using type2 = Qu<intBits<6>, fracBits<-3>>;
            using list = TypeList<type1, type2>;
            using matType = Qu<dim<4, 4>, type1>;
            matType a, b, m3;
            fill_synth(a, 16, 1, 1);
            fill_synth(b, 16, 2, 1);
            Qgemul<QgemulAddArgs<list>, QgemulMulArgs<type1>, QgemulTransposedA<true>>(m3, a, b);
            print_matrix("readme_list_tn_4x4x4", m3, 16);
        }
        {   // linear class on the MFMA path, ragged shape, wide C
            using e43 = Qu<intBits<4>, fracBits<3>>;
            using w16 = Qu<intBits<16>, fracBits<3>>;
            Qu<dim<33, 128>, e43> a;
            Qu<dim<128, 17>, e43> b;
            Qu<dim<33, 17>, w16> c;
            fill_synth(a, 33 * 128, 1, 0);
            fill_synth(b, 128 * 17, 2, 0);
            Qgemul<QgemulMulArgs<intBits<9>, fracBits<6>>, QgemulAddArgs<Qu<intBits<19>, fracBits<6>>>>(c, a, b);
            print_matrix("e43_L_33x17x128_full_wideC", c, 33 * 17);
        }
        {   // element-wise operators after the GEMM through the binding's Then* front-ends, with the operands of the golden
            // record "scale_then_bias" (tests/golden/ref_eltwise_2, oracle/ref_cases_eltwise.cpp case2: seeds 31/32/33,
            // dist 1): X * 1 with K = 1 reproduces the record's tensor, so D must be the record's D
            using c238 = Qu<intBits<23>, fracBits<8>>;
            using one_t = Qu<intBits<1>, fracBits<0>, isSigned<false>>;
            using s34 = Qu<intBits<3>, fracBits<4>>;
            using b106 = Qu<intBits<10>, fracBits<6>>;
            using t1 = Qu<intBits<24>, fracBits<8>>;
            using d124 = Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
            Qu<dim<96, 1>, c238> x;
            Qu<dim<1, 1>, one_t> one;
            Qu<dim<96, 1>, b106> bias;
            Qu<dim<96, 1>, d124> d;
            fill_synth(x, 96, 31, 1);
            one.data[0].data.data = 1;
            fill_synth(bias, 96, 33, 1);
            s34 s;
            s.data.data = refdrv::synth<s34>(32, 1, 0, 0);
            Qgemul<QgemulMulArgs<c238>, QgemulResult<c238>>(d, x, one, ThenMul<t1, intBits<24>, fracBits<8>>(s), ThenAdd<>(bias));
            std::printf("{\"epilogue\":\"scale_then_bias\",\"D\":[");
            for (size_t e = 0; e < 96; ++e) std::printf("%s%lld", e ? "," : "", (long long)d.data[e].data.data);
            std::printf("]}\n");
        }
        {   // the same after a COMPLEX Qgemul: golden record "scale_cbias_real_sub" (tests/golden/ref_cplx_eltwise_2,
            // oracle/ref_cases_cplx_eltwise.cpp case3: seeds 81 .. 84, dist 1).  X * (1 + 0i) with K = 1 and every product
            // sub-operation in C's own part format reproduces the record's tensor; the three operators are a real scalar
            // scale, a complex bias with a realT<> tag and a real tensor subtraction (the imaginary part is carried over).
            using r206 = Qu<intBits<20>, fracBits<6>>;
            using r104 = Qu<intBits<10>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
            using cw = Qcomplex<r206, r206>;
            using cb = Qcomplex<Qu<intBits<5>, fracBits<4>>, Qu<intBits<3>, fracBits<2>>>;
            using cd = Qcomplex<r104, Qu<intBits<8>, fracBits<2>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>>;
            using cq = Qcomplex<Qu<intBits<7>, fracBits<3>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>, Qu<intBits<9>, fracBits<1>, QuMode<TRN::SMGN>, OfMode<SAT::SMGN>>>;
            using s22 = Qu<intBits<2>, fracBits<2>>;
            using r32 = Qu<intBits<3>, fracBits<2>>;
            using one_t = Qu<intBits<1>, fracBits<0>, isSigned<false>>;
            constexpr size_t n = 48;
            Qu<dim<n, 1>, cw> x;
            Qu<dim<1, 1>, Qcomplex<one_t, one_t>> one;
            Qu<dim<n, 1>, cb> bias;
            Qu<dim<n, 1>, r32> off;
            Qu<dim<n, 1>, cq> d;
            for (size_t e = 0; e < n; ++e) {
                refdrv::set_raw(x.data[e], refdrv::synth<r206>(81, 1, e, 0), refdrv::synth<r206>(81, 1, e, 1));
                refdrv::set_raw(bias.data[e], refdrv::synth<cb::realType>(83, 1, e, 0), refdrv::synth<cb::imagType>(83, 1, e, 1));
                off.data[e].data.data = refdrv::synth<r32>(84, 1, e, 0);
            }
            refdrv::set_raw(one.data[0], 1, 0);
            s22 s;
            s.data.data = refdrv::synth<s22>(82, 1, 0, 0);
            Qgemul<QgemulMulArgs<BasicComplexMul<acT<r206>, bdT<r206>, adT<r206>, bcT<r206>, acbdT<r206>, adbcT<r206>>>, QgemulResult<cw>>(
                d, x, one, ThenMul<cw>(s), ThenAdd<cd, realT<r104>>(bias), ThenSub<>(off));
            std::printf("{\"epilogue_cplx\":\"scale_cbias_real_sub\",\"Dre\":[");
            for (size_t e = 0; e < n; ++e) std::printf("%s%lld", e ? "," : "", (long long)d.data[e].real.data.data);
            std::printf("],\"Dim\":[");
            for (size_t e = 0; e < n; ++e) std::printf("%s%lld", e ? "," : "", (long long)d.data[e].imag.data.data);
            std::printf("]}\n");
        }
    } catch (const std::exception& e) {
        std::printf("{\"error\":\"%s\"}\n", e.what());
        return 3;
    }
    return 0;
}
