// ref_bench.cpp — TEST/BENCH INFRASTRUCTURE: times the REAL reference primitives (Qmul + vector
// Qreduce + converting assignment, composed as in ref_driver.hpp) on a bounded block of a
// BASELINE.json configuration, for bench.py's "cpu_baseline" (kind "reference").
// Built in the build container into oracle/_ref/ref_bench; the binary (not the header) travels
// to the GPU box.  Reducer keeps function-local statics (QuBLAS.h:4966, :4995), so parallelism
// is by PROCESS: bench.py starts one copy per core, each on its own row range.
//
//   ref_bench <variant> <row0> <rows> <cols> [dist]
//     variant c3T : 4096-deep int<8,8> TCPL/SAT::ZERO, default tags (tree class)
//             c3L : same operands, MulArgs<int 17,frac 16>, AddArgs<Qu<29,16>> (linear class)
//             c2L : 1024-deep int<4,3>, MulArgs<int 9,frac 6>, AddArgs<Qu<19,6>>
//   prints one JSON line: {"variant","rows","cols","K","macs","seconds","checksum"}
#include "ref_driver.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

using namespace refdrv;

using e88z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
using e43 = Qu<intBits<4>, fracBits<3>>;

template <class E, class MulList, class AddList, size_t K, size_t MTOT>
static int run(const char* name, size_t row0, size_t rows, size_t cols, int dist)
{
    std::vector<E> arow(K), bcol(K);
    uint64_t checksum = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (size_t j = 0; j < cols; ++j) {
        for (size_t k = 0; k < K; ++k) bcol[k].data.data = synth<E>(2, dist, k + j * K, 0);
        for (size_t i = row0; i < row0 + rows; ++i) {
            for (size_t k = 0; k < K; ++k) arow[k].data.data = synth<E>(1, dist, i + k * MTOT, 0);
            E c = ref_dot<E, E, E, MulList, AddList, K>(arow, bcol);
            checksum = checksum * 1099511628211ull + uint64_t(int64_t(c.data.data));
        }
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"variant\":\"%s\",\"rows\":%zu,\"cols\":%zu,\"K\":%zu,\"macs\":%.0f,\"seconds\":%.6f,\"checksum\":%llu}\n", name, rows,
                cols, K, double(rows) * double(cols) * double(K), s, (unsigned long long)checksum);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const char* v = argv[1];
    size_t row0 = std::strtoull(argv[2], 0, 10), rows = std::strtoull(argv[3], 0, 10), cols = std::strtoull(argv[4], 0, 10);
    int dist = argc > 5 ? std::atoi(argv[5]) : 0;
    if (!std::strcmp(v, "c3T")) return run<e88z, TypeList<>, TypeList<>, 4096, 4096>(v, row0, rows, cols, dist);
    if (!std::strcmp(v, "c3L"))
        return run<e88z, TypeList<intBits<17>, fracBits<16>>, TypeList<Qu<intBits<29>, fracBits<16>>>, 4096, 4096>(v, row0, rows, cols, dist);
    if (!std::strcmp(v, "c2L"))
        return run<e43, TypeList<intBits<9>, fracBits<6>>, TypeList<Qu<intBits<19>, fracBits<6>>>, 1024, 1024>(v, row0, rows, cols, dist);
    return 2;
}
