// ref_cases_eltwise.cpp — TEST INFRASTRUCTURE: golden vectors for the element-wise epilogue,
// produced by the reference header's own lazy tensor operators (Qmul / Qadd / Qsub on Qu<dim<…>> tensors and
// scalars) and tensor construction from the resulting expressions (see ref_driver.hpp for the conventions).
//
// Each case: a tensor X of the "C" element type with synthetic raw values stands for a Qgemul result; one to
// three operators are applied the way user code has to write them (one tensor per operator, because the tensor
// front-ends accept tensors, not expressions); the raw values of every operand and of the final tensor D are
// printed together with the resolved result format of every operator as the reference's types report it.
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

template <class T, size_t N>
Qu_s<dim<N>, T> make_tensor(uint64_t seed, int dist, std::vector<int64_t>& raw)
{
    Qu_s<dim<N>, T> t;
    raw.resize(N);
    for (size_t i = 0; i < N; ++i) {
        raw[i] = synth<T>(seed, dist, i, 0);
        set_raw(t[i], raw[i], 0);
    }
    return t;
}

static void print_vec(FILE* out, const char* key, const std::vector<int64_t>& v)
{
    std::fprintf(out, "\"%s\":[", key);
    for (size_t i = 0; i < v.size(); ++i) std::fprintf(out, "%s%lld", i ? "," : "", (long long)v[i]);
    std::fprintf(out, "]");
}

template <class T, size_t N>
std::vector<int64_t> raw_of(const Qu_s<dim<N>, T>& t)
{
    std::vector<int64_t> v(N);
    for (size_t i = 0; i < N; ++i) { int64_t im; get_raw(t[i], v[i], im); }
    return v;
}

// op codes of include/qgemul.h
enum { ADD = 1, SUB = 2, MUL = 3 };

template <int OP, bool XFIRST, class... Tags>
struct Op {
    template <class X, class E>
    static auto apply(const X& x, const E& e)
    {
        if constexpr (OP == MUL) { if constexpr (XFIRST) return Qmul<Tags...>(x, e); else return Qmul<Tags...>(e, x); }
        else if constexpr (OP == ADD) { if constexpr (XFIRST) return Qadd<Tags...>(x, e); else return Qadd<Tags...>(e, x); }
        else { if constexpr (XFIRST) return Qsub<Tags...>(x, e); else return Qsub<Tags...>(e, x); }
    }
    // the operator's scalar result type
    template <class XT, class ET>
    using res_t = decltype(apply(std::declval<XT>(), std::declval<ET>()));
    static constexpr int op = OP;
    static constexpr bool xfirst = XFIRST;
};

struct StageOut {
    std::string json;
};

// one stage: X tensor (elem XT) op E (tensor of ET, or scalar ET) -> tensor of TT
template <class OpT, class XT, class ET, class TT, bool SCALAR, size_t N>
Qu_s<dim<N>, TT> stage(const Qu_s<dim<N>, XT>& X, uint64_t seed, int dist, std::string& js)
{
    using r_t = typename OpT::template res_t<XT, ET>;
    std::vector<int64_t> eraw;
    char buf[256];
    std::snprintf(buf, sizeof buf, "{\"op\":%d,\"x_first\":%d,\"scalar\":%d,\"e\":%s,\"r\":%s,\"t\":%s,", OpT::op, int(OpT::xfirst),
                  int(SCALAR), fmt_json<ET>().c_str(), fmt_json<r_t>().c_str(), fmt_json<TT>().c_str());
    js += buf;
    if constexpr (SCALAR) {
        ET e;
        eraw = {synth<ET>(seed, dist, 0, 0)};
        set_raw(e, eraw[0], 0);
        Qu_s<dim<N>, TT> out = OpT::apply(X, e);
        js += "\"E\":[" + std::to_string((long long)eraw[0]) + "]}";
        return out;
    } else {
        auto E = make_tensor<ET, N>(seed, dist, eraw);
        Qu_s<dim<N>, TT> out = OpT::apply(X, E);
        js += "\"E\":[";
        for (size_t i = 0; i < N; ++i) js += (i ? "," : "") + std::to_string((long long)eraw[i]);
        js += "]}";
        return out;
    }
}

template <class CT, size_t N>
void header(FILE* out, const char* name, const std::vector<int64_t>& x)
{
    std::fprintf(out, "{\"name\":\"%s\",\"n\":%zu,\"c\":%s,", name, N, fmt_json<CT>().c_str());
    print_vec(out, "X", x);
}

template <class DT, size_t N>
void footer(FILE* out, const std::string& stages, const Qu_s<dim<N>, DT>& D)
{
    std::fprintf(out, ",\"stages\":[%s],\"d\":%s,", stages.c_str(), fmt_json<DT>().c_str());
    print_vec(out, "D", raw_of(D));
    std::fprintf(out, "}\n");
}

// one operator
template <class CT, class OpT, class ET, class DT, bool SCALAR, size_t N = 96>
void case1(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> x;
    auto X = make_tensor<CT, N>(21, dist, x);
    std::string js;
    auto D = stage<OpT, CT, ET, DT, SCALAR, N>(X, 22, dist, js);
    header<CT, N>(out, name, x);
    footer<DT, N>(out, js, D);
}

// two operators through an intermediate tensor of T1
template <class CT, class Op1, class E1, class T1, bool S1, class Op2, class E2, class DT, bool S2, size_t N = 96>
void case2(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> x;
    auto X = make_tensor<CT, N>(31, dist, x);
    std::string js;
    auto T = stage<Op1, CT, E1, T1, S1, N>(X, 32, dist, js);
    js += ",";
    auto D = stage<Op2, T1, E2, DT, S2, N>(T, 33, dist, js);
    header<CT, N>(out, name, x);
    footer<DT, N>(out, js, D);
}

template <class CT, class Op1, class E1, class T1, bool S1, class Op2, class E2, class T2, bool S2, class Op3, class E3, class DT, bool S3, size_t N = 96>
void case3(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> x;
    auto X = make_tensor<CT, N>(41, dist, x);
    std::string js;
    auto Ta = stage<Op1, CT, E1, T1, S1, N>(X, 42, dist, js);
    js += ",";
    auto Tb = stage<Op2, T1, E2, T2, S2, N>(Ta, 43, dist, js);
    js += ",";
    auto D = stage<Op3, T2, E3, DT, S3, N>(Tb, 44, dist, js);
    header<CT, N>(out, name, x);
    footer<DT, N>(out, js, D);
}

// pure conversion D = C (tensor converting constructor)
template <class CT, class DT, size_t N = 96>
void case0(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> x;
    auto X = make_tensor<CT, N>(51, dist, x);
    Qu_s<dim<N>, DT> D = X;
    header<CT, N>(out, name, x);
    footer<DT, N>(out, "", D);
}

// element types
using c238 = Qu<intBits<23>, fracBits<8>>;                                                    // bench C type
using c43 = Qu<intBits<4>, fracBits<3>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;                // config-2 C type with modes
using c163 = Qu<intBits<16>, fracBits<3>>;
using u44 = Qu<intBits<4>, fracBits<4>, isSigned<false>>;
using n63 = Qu<intBits<6>, fracBits<-3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using w40 = Qu<intBits<30>, fracBits<10>>;                                                    // 41 storage bits: int64 host elements
using b106 = Qu<intBits<10>, fracBits<6>>;
using s34 = Qu<intBits<3>, fracBits<4>>;
using d124 = Qu<intBits<12>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
using d88z = Qu<intBits<8>, fracBits<8>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
using d62w = Qu<intBits<6>, fracBits<2>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
using d55i = Qu<intBits<5>, fracBits<5>, QuMode<RND::INF>, OfMode<SAT::TCPL>>;
using d70n = Qu<intBits<7>, fracBits<0>, QuMode<RND::NEG_INF>, OfMode<SAT::ZERO>>;
using d91s = Qu<intBits<9>, fracBits<1>, QuMode<TRN::SMGN>, OfMode<SAT::SMGN>>;
using du8 = Qu<intBits<8>, fracBits<0>, isSigned<false>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    switch (part) {
    case 0: // single operators, default and explicit tags, every op / order / scalar-ness
        case1<c238, Op<ADD, true>, b106, d124, false>("add_tensor_default", 0, out);
        case1<c238, Op<ADD, true>, b106, c238, false>("add_tensor_default_sameD", 1, out);
        case1<c238, Op<ADD, true, d124>, b106, d124, false>("add_tensor_fulltag_identityD", 0, out);
        case1<c238, Op<ADD, false, intBits<20>, OfMode<SAT::ZERO>>, b106, d88z, false>("add_efirst_loosetags", 0, out);
        case1<c238, Op<ADD, true, FullPrec>, b106, w40, false>("add_fullprec", 0, out);
        case1<c238, Op<SUB, true>, b106, d124, false>("sub_xfirst", 0, out);
        case1<c238, Op<SUB, false>, b106, d124, false>("sub_efirst", 0, out);
        case1<c238, Op<SUB, false, fracBits<2>, QuMode<RND::CONV>>, b106, d62w, false>("sub_efirst_tags_wrap", 1, out);
        case1<c238, Op<MUL, true>, s34, d124, true>("mul_scalar_default", 1, out);
        case1<c238, Op<MUL, true, intBits<24>, fracBits<8>>, s34, d88z, true>("mul_scalar_tags", 0, out);
        case1<c238, Op<MUL, false, d55i>, s34, d55i, false>("mul_tensor_efirst_fulltag", 1, out);
        case1<c238, Op<MUL, true, FullPrec>, s34, w40, false>("mul_fullprec_wideD", 0, out);
        case1<c238, Op<ADD, true>, s34, d91s, true>("add_scalar", 0, out);
        case1<c238, Op<SUB, false>, s34, d70n, true>("sub_scalar_efirst", 1, out);
        break;
    case 1: // other C types: int8-class C with modes, unsigned, negative frac, 64-bit host elements
        case1<c43, Op<ADD, true>, c43, c43, false>("c43_add_same_type", 0, out);
        case1<c43, Op<MUL, true, intBits<6>, fracBits<5>>, u44, du8, false>("c43_mul_unsigned_tensor", 0, out);
        case1<u44, Op<SUB, true>, u44, d62w, false>("u44_sub_unsigned_default", 0, out);
        case1<u44, Op<SUB, false, isSigned<true>>, u44, d91s, true>("u44_sub_scalar_signed_tag", 0, out);
        case1<n63, Op<ADD, true>, c163, d124, false>("negfrac_add", 0, out);
        case1<n63, Op<MUL, true>, n63, c163, false>("negfrac_mul", 1, out);
        case1<w40, Op<ADD, true>, w40, w40, false>("w40_add_int64_elems", 0, out);
        case1<w40, Op<MUL, true, intBits<30>, fracBits<10>, QuMode<RND::CONV>>, s34, w40, true>("w40_mul_scalar_int64", 1, out);
        case1<c163, Op<SUB, true, QuMode<RND::INF>, fracBits<0>>, b106, d70n, false>("c163_sub_roundtag", 0, out);
        case0<c238, d124>("convert_only_narrow", 0, out);
        case0<c238, c238>("convert_only_identity", 0, out);
        case0<w40, d88z>("convert_only_int64_to_zero", 0, out);
        break;
    case 2: // chains through intermediate tensors (scale then bias, the README-style post-processing)
        case2<c238, Op<MUL, true, intBits<24>, fracBits<8>>, s34, Qu<intBits<24>, fracBits<8>>, true, Op<ADD, true>, b106, d124, false>("scale_then_bias", 1, out);
        case2<c238, Op<MUL, true>, s34, d88z, true, Op<ADD, true, d124>, b106, d124, false>("scale_into_narrow_then_bias", 0, out);
        case2<c238, Op<ADD, true>, b106, c238, false, Op<MUL, false, FullPrec>, s34, w40, false>("bias_then_mul_fullprec", 1, out);
        case2<c43, Op<SUB, false>, c43, d62w, false, Op<ADD, true, QuMode<RND::CONV>, OfMode<WRP::TCPL>>, u44, d55i, true>("c43_sub_then_add_wrap", 0, out);
        case3<c238, Op<MUL, true, intBits<24>, fracBits<8>>, s34, Qu<intBits<24>, fracBits<8>>, true, Op<ADD, true>, b106, d124, false, Op<SUB, false>, s34, d91s, true>("scale_bias_negate", 1, out);
        case3<c163, Op<ADD, true>, c163, c163, false, Op<ADD, true, FullPrec>, c163, Qu<intBits<17>, fracBits<3>>, false, Op<MUL, true, intBits<12>, fracBits<2>, QuMode<RND::ZERO>, OfMode<SAT::SMGN>>, u44, d62w, true>("three_stage_mixed", 0, out);
        break;
    default:
        return 2;
    }
    return 0;
}
