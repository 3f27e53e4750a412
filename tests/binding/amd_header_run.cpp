// Runs Qgemul and Qreduce through include/QuBLAS_amd.h on the GPU (compiled by tests/test_gpu_cpp_dropin.py
// on the GPU box with clang++ -std=c++23 and linked against qublas_amd/libqugemm.so).
#include "QuBLAS_amd.h"

#include <cstdio>

using namespace QuBLAS_amd;

template <class T>
static void print_matrix(const char* name, const T& m)
{
    std::printf("{\"name\":\"%s\",\"C\":[", name);
    for (size_t e = 0; e < m.data.size(); ++e) std::printf("%s%lld", e ? "," : "", (long long)m.data[e].data);
    std::printf("]}\n");
}

int main()
{
#ifdef QUBLAS_TEST_ALL_DEVICES
    QgemulRunFlags() |= QG_OPT_ALL_DEVICES;   // every Qgemul<...> below is row-sharded over all visible gfx950 devices (qgemul_run_sharded)
#endif
    try {
        using e88z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
        using mat = Qu<dim<4, 4>, e88z>;
        mat m1 = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0}, m3;
        Qgemul<QgemulAddArgs<e88z>, QgemulMulArgs<e88z>>(m3, m1, m1);
        print_matrix("c1_nn_classT", m3);
        Qgemul<QgemulTransposedA<true>, QgemulMulArgs<e88z>, QgemulAddArgs<TypeList<e88z>>>(m3, m1, m1);
        print_matrix("c1_tn_classT", m3);
        Qgemul<>(m3, m1, m1);
        print_matrix("c1_nn_default", m3);
        Qgemul<QgemulMulArgs<intBits<17>, fracBits<16>>, QgemulAddArgs<Qu<intBits<29>, fracBits<16>>>>(m3, m1, m1);
        print_matrix("c1_nn_classL", m3);
        // Qreduce of a vector: 1+2+3+4 in int<4,3> reduced into Qu<intBits<8>,fracBits<3>> = 10.0 -> raw 80
        using e43 = Qu<intBits<4>, fracBits<3>>;
        Qu<dim<4>, e43> v = {1.0, 2.0, 3.0, 4.0};
        auto r = Qreduce<Qu<intBits<8>, fracBits<3>>>(v);
        std::printf("{\"name\":\"qreduce\",\"C\":[%lld]}\n", (long long)r.data);
        // a signed SAT::SMGN element type whose raw minimum -2^W is present: the reference's Qreduce adds it as it is (the leaf
        // format of the lowering is the element's with SAT::TCPL; tests/golden/ref_scalar_7)
        using sm = Qu<intBits<3>, fracBits<4>, OfMode<SAT::SMGN>>;
        Qu<dim<16>, sm> w;
        for (size_t i = 0; i < 16; ++i) w[i].fill(int64_t((i * 37) % 256) - 128);
        w[0].fill(-128);
        w[8].fill(-128);
        auto r1 = Qreduce<Qu<intBits<4>, fracBits<6>, OfMode<SAT::ZERO>>>(w);
        auto r2 = Qreduce<>(w);
        std::printf("{\"name\":\"qreduce_smgn\",\"C\":[%lld,%lld]}\n", (long long)r1.data, (long long)r2.data);
        // the VARIADIC overload (readme.md:62): scalars of two types, the reference tables' inputs of seed 1 (tests/golden/ref_scalar_10)
        {
            using t1 = Qu<intBits<4>, fracBits<3>>;
            using t2 = Qu<intBits<6>, fracBits<1>, QuMode<RND::POS_INF>, OfMode<SAT::SMGN>>;
            using nar = Qu<intBits<5>, fracBits<2>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
            using wide = Qu<intBits<12>, fracBits<5>, QuMode<RND::ZERO>>;
            t1 a, c, e, g; t2 b, d, f;
            a.fill(17); b.fill(120); c.fill(-15); d.fill(96); e.fill(-55);
            f.fill(-25); g.fill(-12);
            auto v4 = Qreduce<t1>(a, b, a, b);
            auto v3 = Qreduce<>(a, b, c);
            auto v5 = Qreduce<nar, wide>(a, b, c, d, e);
            auto v6 = Qreduce<nar, wide>(a, b, c, d, e, f);
            auto v7 = Qreduce<nar, wide>(a, b, c, d, e, f, g);
            auto v7n = Qreduce<nar>(a, b, c, d, e, f, g);
            auto v5l = Qreduce<TypeList<wide, nar>>(a, b, c, d, e);
            std::printf("{\"name\":\"qreduce_variadic\",\"C\":[%lld,%lld,%lld,%lld,%lld,%lld,%lld],\"F\":[%d,%d,%d,%d,%d,%d,%d]}\n", (long long)v4.data, (long long)v3.data,
                        (long long)v5.data, (long long)v6.data, (long long)v7.data, (long long)v7n.data, (long long)v5l.data, decltype(v4)::fracB, decltype(v3)::fracB,
                        decltype(v5)::fracB, decltype(v6)::fracB, decltype(v7)::fracB, decltype(v7n)::fracB, decltype(v5l)::fracB);
        }
    } catch (const std::exception& e) {
        std::printf("{\"error\":\"%s\"}\n", e.what());
        return 3;
    }
    return 0;
}
