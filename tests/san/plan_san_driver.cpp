// TEST INFRASTRUCTURE: the host-side planner (qublas_amd/csrc/qg_plan.cpp — interval propagation, class L / class T, lowering of
// the element-wise chain) compiled for the CPU with AddressSanitizer + UndefinedBehaviorSanitizer, as the reference compiles its
// own tests (CMakeLists.txt:17,26), and driven with descriptors read from stdin: the raw bytes of qgemul_desc structs (and,
// after each, one qgemul_epilogue).  Prints status, class and kernel-selection flags per descriptor; any sanitizer report
// fails the run.  No GPU code is involved.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../qublas_amd/csrc/qg_plan.h"

int main()
{
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, stdin)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    const size_t rec = sizeof(qgemul_desc) + sizeof(qgemul_epilogue);
    size_t count = 0;
    for (size_t off = 0; off + rec <= buf.size(); off += rec, ++count) {
        qgemul_desc d;
        qgemul_epilogue ep;
        memcpy(&d, buf.data() + off, sizeof d);
        memcpy(&ep, buf.data() + off + sizeof d, sizeof ep);
        QAnalysis* an = new QAnalysis;
        qg_analyze(&d, an);
        int ep_st = -99, ep_bits = 0;
        if (an->status == QG_OK && !d.is_complex && ep.n_stages <= QG_MAX_EW) {
            QEpTable t;
            char why[96];
            ep_st = qg_analyze_ep(d.c[0], &ep, &t, &ep_bits, why, sizeof why);
        }
        printf("%zu %d %d %d %d %d %d %d %d %d\n", count, an->status, an->cls, an->max_bits, an->linear_ok ? 1 : 0, an->tree_fast_ok ? 1 : 0, ep_st,
               an->status == QG_OK ? an->fast_mode : -1, an->status == QG_OK ? an->cplx_fixed_ok : -1, an->status == QG_OK ? an->gemv_fixed : -1);
        delete an;
    }
    fprintf(stderr, "analysed %zu descriptors\n", count);
    return 0;
}
