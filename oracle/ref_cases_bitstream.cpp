// ref_cases_bitstream.cpp — TEST INFRASTRUCTURE: golden strings for the BitStream export of a tensor,
// produced by the reference header's own BitStream<tensorProcessT, elemProcessT>(tensor) (QuBLAS.h:4811-4827) on tensors
// with synthetic raw values.  Each record: element format, tensor dims, raw values in storage order, the two
// processing tags (0 = l2r, k = r2l<k>) and the string the reference returned.
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
    std::vector<int64_t> raw(n);
    for (size_t i = 0; i < n; ++i) {
        raw[i] = synth<T>(seed, dist, i, 0);
        set_raw(t[i], raw[i], 0);
    }
    std::string s = BitStream<TP, EP>(t);
    std::fprintf(out, "{\"name\":\"%s\",\"fmt\":%s,\"n\":%zu,\"rows\":%zu,\"tensor_chunk\":%zu,\"elem_chunk\":%zu,\"X\":[", name,
                 fmt_json<T>().c_str(), n, dim<dims...>::template dimAt<0>, chunk_of<TP>::v, chunk_of<EP>::v);
    for (size_t i = 0; i < n; ++i) std::fprintf(out, "%s%lld", i ? "," : "", (long long)raw[i]);
    std::fprintf(out, "],\"bits\":\"%s\"}\n", s.c_str());
}

using s50 = Qu<intBits<5>, fracBits<0>>;                       // main.cpp's demo type (6 chars per element)
using s88 = Qu<intBits<8>, fracBits<8>>;                       // 17 chars
using u44 = Qu<intBits<4>, fracBits<4>, isSigned<false>>;      // 8 chars, no sign bit in the string
using s43 = Qu<intBits<4>, fracBits<3>>;                       // 8 chars
using s238 = Qu<intBits<23>, fracBits<8>>;                     // 32 chars
using w40 = Qu<intBits<30>, fracBits<10>>;                     // 41 chars, 64-bit storage
using n63 = Qu<intBits<6>, fracBits<-3>>;                      // 4 chars

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    if (part != 0) return 2;
    {   // the demo of the reference's main.cpp: v = {1..6}, BitStream<r2l<1>, l2r>
        Qu_s<dim<2, 3>, s50> v = {1, 2, 3, 4, 5, 6};
        std::string s = BitStream<r2l<1>, l2r>(v);
        std::fprintf(out, "{\"name\":\"main_cpp_demo\",\"fmt\":%s,\"n\":6,\"rows\":2,\"tensor_chunk\":1,\"elem_chunk\":0,\"X\":[1,2,3,4,5,6],\"bits\":\"%s\"}\n",
                     fmt_json<s50>().c_str(), s.c_str());
    }
    one<s50, l2r, l2r, 2, 3>("s50_l2r_l2r", 1, 0, out);
    one<s50, r2l<1>, r2l<>, 2, 3>("s50_r2l1_r2l_default", 2, 0, out);
    one<s50, r2l<2>, r2l<3>, 4, 3>("s50_r2l2_r2l3", 3, 0, out);
    one<s50, l2r, r2l<2>, 5>("s50_vector_l2r_r2l2", 4, 0, out);
    one<s88, l2r, l2r, 4, 4>("s88_l2r_l2r", 5, 0, out);
    one<s88, r2l<4>, l2r, 4, 4>("s88_r2l4_l2r_columns_reversed", 6, 0, out);
    one<s88, r2l<1>, r2l<1>, 4, 4>("s88_r2l1_r2l1", 7, 0, out);
    one<u44, l2r, r2l<4>, 3, 5>("u44_l2r_r2l4_nibbles", 8, 0, out);
    one<u44, r2l<5>, r2l<2>, 3, 5>("u44_r2l5_r2l2", 9, 0, out);
    one<s43, r2l<8>, r2l<8>, 8, 4>("s43_r2l8_r2l8_identity_elem", 10, 0, out);
    one<s238, l2r, r2l<8>, 4, 6>("s238_l2r_r2l8_bytes", 11, 0, out);
    one<s238, r2l<3>, r2l<16>, 4, 6>("s238_r2l3_r2l16", 12, 1, out);
    one<w40, l2r, l2r, 3, 3>("w40_l2r_l2r_int64", 13, 0, out);
    one<w40, r2l<1>, r2l<41>, 3, 3>("w40_r2l1_r2l41", 14, 0, out);
    one<n63, r2l<2>, r2l<2>, 4, 4>("n63_r2l2_r2l2", 15, 0, out);
    one<s88, l2r, l2r, 16, 16>("s88_16x16_l2r", 16, 0, out);
    one<s88, r2l<16>, r2l<1>, 16, 16>("s88_16x16_r2l16_r2l1", 17, 0, out);
    return 0;
}
