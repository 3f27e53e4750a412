// shared by the header probes: print a qgemul_desc in the same shape as the golden records
#pragma once
#include <cstdio>
#include "qgemul.h"

inline void print_fmt(const qfmt& f) { std::printf("[%d,%d,%d,%d,%d]", f.I, f.F, f.S, f.Q, f.O); }
inline void print_desc(const char* name, const qgemul_desc& d)
{
    std::printf("{\"name\":\"%s\",\"M\":%lld,\"N\":%lld,\"K\":%lld,\"transA\":%d,\"is_complex\":%d,\"cmul\":%d,\"a\":[", name, (long long)d.M,
                (long long)d.N, (long long)d.K, d.transA, d.is_complex, d.cmul);
    print_fmt(d.a[0]); std::printf(","); print_fmt(d.a[1]);
    std::printf("],\"b\":["); print_fmt(d.b[0]); std::printf(","); print_fmt(d.b[1]);
    std::printf("],\"c\":["); print_fmt(d.c[0]); std::printf(","); print_fmt(d.c[1]);
    std::printf("],\"mul\":[");
    for (int i = 0; i < 8; ++i) { if (i) std::printf(","); print_fmt(d.mul[i]); }
    std::printf("],\"n_levels\":%u,\"level_add\":[", d.n_levels);
    for (unsigned l = 0; l < d.n_levels; ++l) { if (l) std::printf(","); std::printf("["); print_fmt(d.level_add[0][l]); std::printf(","); print_fmt(d.level_add[1][l]); std::printf("]"); }
    std::printf("],\"level\":[");
    for (unsigned l = 0; l < d.n_levels; ++l) { if (l) std::printf(","); std::printf("["); print_fmt(d.level[0][l]); std::printf(","); print_fmt(d.level[1][l]); std::printf("]"); }
    std::printf("]}\n");
}

// an element-wise chain in the shape of the golden records of tests/golden/ref_eltwise_*
inline void print_epilogue(const char* name, const qgemul_epilogue& ep)
{
    std::printf("{\"epilogue\":\"%s\",\"stages\":[", name);
    for (unsigned k = 0; k < ep.n_stages; ++k) {
        const qgemul_ew_stage& s = ep.stage[k];
        std::printf("%s{\"op\":%d,\"x_first\":%d,\"scalar\":%d,\"e\":", k ? "," : "", s.op, s.x_first, s.e_scalar);
        print_fmt(s.e); std::printf(",\"r\":"); print_fmt(s.r); std::printf(",\"t\":"); print_fmt(s.t); std::printf("}");
    }
    std::printf("],\"d\":"); print_fmt(ep.d); std::printf("}\n");
}

// a complex chain (qgemul_epilogue_cplx): per stage and part what tests/golden_io.py's cplx_eltwise_epilogue builds from a
// golden record of tests/golden/ref_cplx_eltwise_*
inline void print_epilogue_cplx(const char* name, const qgemul_epilogue_cplx& ep)
{
    std::printf("{\"epilogue_cplx\":\"%s\",\"stages\":[", name);
    for (unsigned k = 0; k < ep.part[0].n_stages; ++k) {
        std::printf("%s{\"e_complex\":%d,\"parts\":[", k ? "," : "", ep.e_complex[k]);
        for (int p = 0; p < 2; ++p) {
            const qgemul_ew_stage& s = ep.part[p].stage[k];
            std::printf("%s{\"op\":%d,\"x_first\":%d,\"scalar\":%d,\"e\":", p ? "," : "", s.op, s.x_first, s.e_scalar);
            print_fmt(s.e); std::printf(",\"r\":"); print_fmt(s.r); std::printf(",\"t\":"); print_fmt(s.t); std::printf("}");
        }
        std::printf("]}");
    }
    std::printf("],\"n\":[%u,%u],\"d\":[", ep.part[0].n_stages, ep.part[1].n_stages);
    print_fmt(ep.part[0].d); std::printf(","); print_fmt(ep.part[1].d); std::printf("]}\n");
}

