// Runs Qgemul with element-wise operators (ThenMul / ThenAdd / ThenRsub) through include/QuBLAS_amd.h on the GPU
// (compiled by tests/test_gpu_cpp_dropin.py with clang++ -std=c++23, linked against qublas_amd/libqugemm.so).
// Prints the lowered epilogue and D; the test recomputes D with the oracle from the same raw inputs.
#include "QuBLAS_amd.h"

#include <cstdio>

using namespace QuBLAS_amd;

static void print_fmt(const char* key, qfmt f) { std::printf("\"%s\":[%d,%d,%d,%d,%d]", key, f.I, f.F, f.S, f.Q, f.O); }

int main()
{
    try {
        using e88 = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
        using ct = Qu<intBits<15>, fracBits<8>>;                      // the Qgemul result's element type
        using t1 = Qu<intBits<16>, fracBits<8>>;                      // tensor the scaled result is assigned to
        using bt = Qu<intBits<10>, fracBits<6>>;
        using st = Qu<intBits<3>, fracBits<3>>;
        using dt = Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        constexpr size_t M = 24, N = 10, K = 40;
        Qu<dim<M, K>, e88> A;
        Qu<dim<K, N>, e88> B;
        Qu<dim<M, N>, bt> Bias;
        Qu<dim<M, N>, dt> D;
        for (size_t i = 0; i < M * K; ++i) A[i].fill(int64_t((i * 2654435761ull) % 8192) - 4096);
        for (size_t i = 0; i < K * N; ++i) B[i].fill(int64_t((i * 40503ull + 7) % 8192) - 4096);
        for (size_t i = 0; i < M * N; ++i) Bias[i].fill(int64_t((i * 97ull) % 131072) - 65536);
        st s;
        s.fill(13);
        st off;
        off.fill(-20);
        Qgemul<QgemulMulArgs<intBits<17>, fracBits<16>>, QgemulAddArgs<Qu<intBits<29>, fracBits<16>>>, QgemulResult<ct>>(
            D, A, B, ThenMul<t1, intBits<16>, fracBits<8>>(s), ThenAdd<>(Bias), ThenRsub<void, QuMode<RND::CONV>>(off));
        const qgemul_epilogue ep = Qgemul_lower_epilogue<QgemulResult<ct>>(D, ThenMul<t1, intBits<16>, fracBits<8>>(s), ThenAdd<>(Bias),
                                                                         ThenRsub<void, QuMode<RND::CONV>>(off));
        std::printf("{\"name\":\"scale_bias_rsub\",\"M\":%zu,\"N\":%zu,\"K\":%zu,\"n_stages\":%u,", M, N, K, ep.n_stages);
        for (uint32_t k = 0; k < ep.n_stages; ++k) {
            char key[8];
            std::snprintf(key, sizeof key, "r%u", k); print_fmt(key, ep.stage[k].r); std::printf(",");
            std::snprintf(key, sizeof key, "t%u", k); print_fmt(key, ep.stage[k].t); std::printf(",");
            std::printf("\"op%u\":[%d,%d,%d],", k, ep.stage[k].op, ep.stage[k].x_first, ep.stage[k].e_scalar);
        }
        print_fmt("d", ep.d);
        std::printf(",\"D\":[");
        for (size_t e = 0; e < D.data.size(); ++e) std::printf("%s%lld", e ? "," : "", (long long)D.data[e].data);
        std::printf("]}\n");
        {   // the same after a COMPLEX Qgemul (configuration 5's element type, TFComplexMul): a real scalar scale, a complex
            // bias tensor with per-part tags, and `real tensor - x` (the imaginary part becomes 0 - x.imag)
            using r63 = Qu<intBits<6>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
            using r6n3 = Qu<intBits<6>, fracBits<-3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
            using c5 = Qcomplex<r63, r6n3>;
            using cw = Qcomplex<Qu<intBits<18>, fracBits<6>>, Qu<intBits<18>, fracBits<6>>>;
            using cbias = Qcomplex<Qu<intBits<5>, fracBits<4>>, Qu<intBits<3>, fracBits<2>>>;
            using cdst = Qcomplex<Qu<intBits<10>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, Qu<intBits<20>, fracBits<12>>>;   // {int32; int64}
            using s22 = Qu<intBits<2>, fracBits<2>>;
            using r54 = Qu<intBits<5>, fracBits<4>>;
            constexpr size_t cM = 12, cN = 9, cK = 32;
            Qu<dim<cM, cK>, c5> cA;
            Qu<dim<cK, cN>, c5> cB;
            Qu<dim<cM, cN>, cbias> cBias;
            Qu<dim<cM, cN>, r54> cOff;
            Qu<dim<cM, cN>, cdst> cD;
            for (size_t i = 0; i < cM * cK; ++i) { cA[i].real.fill(int64_t((i * 37ull) % 1024) - 512); cA[i].imag.fill(int64_t((i * 11ull + 3) % 16) - 8); }
            for (size_t i = 0; i < cK * cN; ++i) { cB[i].real.fill(int64_t((i * 53ull + 1) % 1024) - 512); cB[i].imag.fill(int64_t((i * 7ull) % 16) - 8); }
            for (size_t i = 0; i < cM * cN; ++i) { cBias[i].real.fill(int64_t((i * 29ull) % 1024) - 512); cBias[i].imag.fill(int64_t((i * 13ull) % 64) - 32); cOff[i].fill(int64_t((i * 41ull) % 1024) - 512); }
            s22 cs;
            cs.fill(-7);
            Qgemul<QgemulMulArgs<TFComplexMul<>>, QgemulResult<cw>>(
                cD, cA, cB, ThenMul<cw, realT<intBits<18>, fracBits<6>>, imagT<intBits<18>, fracBits<6>>>(cs), ThenAdd<void, imagT<intBits<19>>>(cBias), ThenRsub<>(cOff));
            std::printf("{\"name\":\"cplx_scale_cbias_rsub\",\"M\":%zu,\"N\":%zu,\"K\":%zu,\"elem_bytes\":%zu,\"Dre\":[", cM, cN, cK, sizeof(cdst));
            for (size_t e = 0; e < cD.data.size(); ++e) std::printf("%s%lld", e ? "," : "", (long long)cD.data[e].real.data);
            std::printf("],\"Dim\":[");
            for (size_t e = 0; e < cD.data.size(); ++e) std::printf("%s%lld", e ? "," : "", (long long)cD.data[e].imag.data);
            std::printf("]}\n");
        }
    } catch (const std::exception& e) {
        std::printf("{\"error\":\"%s\"}\n", e.what());
        return 3;
    }
    return 0;
}
