// ref_cases_cplx_bitstream.cpp — TEST INFRASTRUCTURE: golden strings for the BitStream export of a COMPLEX tensor,
// produced by the reference header's own BitStream<tensorProcessT, elemProcessT>(tensor) (QuBLAS.h:4811-4827; a complex
// element prints as "(re, im)", :2553-2556) on tensors with synthetic raw values.  Each record: the two part formats,
// tensor dims, raw values per part in storage order, the two processing tags (0 = l2r, k = r2l<k>) and the string the
// reference returned.
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

template <class P> struct chunk_of { static constexpr size_t v = 0; };
template <size_t... k> struct chunk_of<r2l<k...>> { static constexpr size_t v = r2l<k...>::index; };

template <class T, class TP, class EP, size_t... dims>
void one(const char* name, uint64_t seed, int dist, FILE* out)
{
    using tensor_t = Qu_s<dim<dims...>, T>;
    tensor_t t;
    constexpr size_t n = tensor_t::elemSize;
    std::vector<int64_t> re(n), im(n);
    for (size_t i = 0; i < n; ++i) {
        re[i] = synth<typename T::realType>(seed, dist, i, 0);
        im[i] = synth<typename T::imagType>(seed, dist, i, 1);
        set_raw(t[i], re[i], im[i]);
    }
    std::string s = BitStream<TP, EP>(t);
    std::fprintf(out, "{\"name\":\"%s\",\"fmt\":%s,\"n\":%zu,\"rows\":%zu,\"tensor_chunk\":%zu,\"elem_chunk\":%zu,\"Xre\":[", name,
                 fmt2_json<T>().c_str(), n, dim<dims...>::template dimAt<0>, chunk_of<TP>::v, chunk_of<EP>::v);
    for (size_t i = 0; i < n; ++i) std::fprintf(out, "%s%lld", i ? "," : "", (long long)re[i]);
    std::fprintf(out, "],\"Xim\":[");
    for (size_t i = 0; i < n; ++i) std::fprintf(out, "%s%lld", i ? "," : "", (long long)im[i]);
    std::fprintf(out, "],\"bits\":\"%s\"}\n", s.c_str());
}

using c5 = Qcomplex<Qu<intBits<6>, fracBits<3>>, Qu<intBits<6>, fracBits<-3>>>;                  // 10 + 4 + 4 = 18 characters
using c88 = Qcomplex<Qu<intBits<8>, fracBits<8>>, Qu<intBits<8>, fracBits<8>>>;                   // 17 + 17 + 4 = 38
using cu = Qcomplex<Qu<intBits<4>, fracBits<4>, isSigned<false>>, Qu<intBits<5>, fracBits<2>>>;  // 8 + 8 + 4 = 20
using cm = Qcomplex<Qu<intBits<20>, fracBits<6>>, Qu<intBits<30>, fracBits<10>>>;                 // 27 + 41 + 4 = 72, int64 imaginary part

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    if (part != 0) return 2;
    one<c5, l2r, l2r, 2, 3>("c5_l2r_l2r", 1, 0, out);
    one<c5, r2l<1>, l2r, 2, 3>("c5_r2l1_l2r", 2, 0, out);
    one<c5, r2l<2>, r2l<>, 4, 3>("c5_r2l2_r2l_default_chars_reversed", 3, 0, out);
    one<c5, l2r, r2l<6>, 5>("c5_vector_l2r_r2l6", 4, 0, out);
    one<c5, r2l<3>, r2l<9>, 3, 3>("c5_r2l3_r2l9_halves_swapped", 5, 0, out);
    one<c88, l2r, l2r, 4, 4>("c88_l2r_l2r", 6, 0, out);
    one<c88, r2l<4>, r2l<2>, 4, 4>("c88_r2l4_r2l2", 7, 0, out);
    one<c88, r2l<1>, r2l<19>, 4, 4>("c88_r2l1_r2l19", 8, 1, out);
    one<cu, l2r, r2l<4>, 3, 5>("cu_l2r_r2l4_nibbles", 9, 0, out);
    one<cu, r2l<5>, r2l<10>, 3, 5>("cu_r2l5_r2l10", 10, 0, out);
    one<cm, l2r, l2r, 3, 3>("cm_l2r_l2r_int64_part", 11, 0, out);
    one<cm, r2l<3>, r2l<8>, 3, 3>("cm_r2l3_r2l8", 12, 0, out);
    one<c88, r2l<16>, r2l<1>, 16, 16>("c88_16x16_r2l16_r2l1", 13, 0, out);
    return 0;
}
