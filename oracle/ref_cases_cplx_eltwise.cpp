// ref_cases_cplx_eltwise.cpp — TEST INFRASTRUCTURE: golden vectors for element-wise operators on COMPLEX tensors
// (include/qgemul.h, qgemul_epilogue_cplx), produced by the reference header's own lazy tensor operators on
// Qu<dim<N>, Qcomplex<...>> tensors with complex and real, tensor and scalar operands, and by the tensors' converting
// construction from the resulting expressions (conventions: ref_driver.hpp).
//
// A tensor X of the complex "C" element type with synthetic raw values stands for a complex Qgemul result; one to
// three operators are applied the way user code writes them (one tensor per operator).  Printed per operator: the
// operand's format(s), the operator's scalar result type as the reference's types report it, the element type of the
// tensor it was assigned to, and the operand's raw values; then the raw values of the final tensor D, part by part.
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

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
    template <class XT, class ET>
    using res_t = decltype(apply(std::declval<XT>(), std::declval<ET>()));
    static constexpr int op = OP;
    static constexpr bool xfirst = XFIRST;
};

template <class T, size_t N>
Qu_s<dim<N>, T> make_tensor(uint64_t seed, int dist, std::vector<int64_t>& re, std::vector<int64_t>& im)
{
    Qu_s<dim<N>, T> t;
    re.resize(N);
    im.resize(N);
    for (size_t i = 0; i < N; ++i) {
        re[i] = synth<typename parts<T>::re>(seed, dist, i, 0);
        im[i] = is_cplx<T> ? synth<typename parts<T>::im>(seed, dist, i, 1) : 0;
        set_raw(t[i], re[i], im[i]);
    }
    return t;
}

static std::string vec_json(const char* key, const std::vector<int64_t>& v)
{
    std::string s = std::string("\"") + key + "\":[";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string((long long)v[i]);
    return s + "]";
}

// one operator: X (complex elements XT) op E (tensor or scalar of ET, complex or real) -> tensor of TT
template <class OpT, class XT, class ET, class TT, bool SCALAR, size_t N>
Qu_s<dim<N>, TT> stage(const Qu_s<dim<N>, XT>& X, uint64_t seed, int dist, std::string& js)
{
    using r_t = typename OpT::template res_t<XT, ET>;
    static_assert(is_cplx<r_t> && is_cplx<TT>);
    char buf[512];
    std::snprintf(buf, sizeof buf, "{\"op\":%d,\"x_first\":%d,\"scalar\":%d,\"e_complex\":%d,\"e\":%s,\"r\":%s,\"t\":%s,", OpT::op,
                  int(OpT::xfirst), int(SCALAR), int(is_cplx<ET>), fmt2_json<ET>().c_str(), fmt2_json<r_t>().c_str(), fmt2_json<TT>().c_str());
    js += buf;
    std::vector<int64_t> ere, eim;
    if constexpr (SCALAR) {
        ET e;
        ere = {synth<typename parts<ET>::re>(seed, dist, 0, 0)};
        eim = {is_cplx<ET> ? synth<typename parts<ET>::im>(seed, dist, 0, 1) : 0};
        set_raw(e, ere[0], eim[0]);
        Qu_s<dim<N>, TT> out = OpT::apply(X, e);
        js += vec_json("Ere", ere) + "," + vec_json("Eim", eim) + "}";
        return out;
    } else {
        auto E = make_tensor<ET, N>(seed, dist, ere, eim);
        Qu_s<dim<N>, TT> out = OpT::apply(X, E);
        js += vec_json("Ere", ere) + "," + vec_json("Eim", eim) + "}";
        return out;
    }
}

template <class CT, class DT, size_t N>
void emit(FILE* out, const char* name, const std::vector<int64_t>& xre, const std::vector<int64_t>& xim, const std::string& stages,
          const Qu_s<dim<N>, DT>& D)
{
    std::vector<int64_t> dre(N), dim_(N);
    for (size_t i = 0; i < N; ++i) get_raw(D[i], dre[i], dim_[i]);
    std::fprintf(out, "{\"name\":\"%s\",\"n\":%zu,\"c\":%s,%s,%s,\"stages\":[%s],\"d\":%s,%s,%s}\n", name, N, fmt2_json<CT>().c_str(),
                 vec_json("Xre", xre).c_str(), vec_json("Xim", xim).c_str(), stages.c_str(), fmt2_json<DT>().c_str(),
                 vec_json("Dre", dre).c_str(), vec_json("Dim", dim_).c_str());
}

template <class CT, class OpT, class ET, class DT, bool SCALAR, size_t N = 48>
void case1(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> xre, xim;
    auto X = make_tensor<CT, N>(61, dist, xre, xim);
    std::string js;
    auto D = stage<OpT, CT, ET, DT, SCALAR, N>(X, 62, dist, js);
    emit<CT, DT, N>(out, name, xre, xim, js, D);
}

template <class CT, class Op1, class E1, class T1, bool S1, class Op2, class E2, class DT, bool S2, size_t N = 48>
void case2(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> xre, xim;
    auto X = make_tensor<CT, N>(71, dist, xre, xim);
    std::string js;
    auto T = stage<Op1, CT, E1, T1, S1, N>(X, 72, dist, js);
    js += ",";
    auto D = stage<Op2, T1, E2, DT, S2, N>(T, 73, dist, js);
    emit<CT, DT, N>(out, name, xre, xim, js, D);
}

template <class CT, class Op1, class E1, class T1, bool S1, class Op2, class E2, class T2, bool S2, class Op3, class E3, class DT, bool S3, size_t N = 48>
void case3(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> xre, xim;
    auto X = make_tensor<CT, N>(81, dist, xre, xim);
    std::string js;
    auto Ta = stage<Op1, CT, E1, T1, S1, N>(X, 82, dist, js);
    js += ",";
    auto Tb = stage<Op2, T1, E2, T2, S2, N>(Ta, 83, dist, js);
    js += ",";
    auto D = stage<Op3, T2, E3, DT, S3, N>(Tb, 84, dist, js);
    emit<CT, DT, N>(out, name, xre, xim, js, D);
}

// pure conversion D = C (complex tensor converting constructor: part by part)
template <class CT, class DT, size_t N = 48>
void case0(const char* name, int dist, FILE* out)
{
    std::vector<int64_t> xre, xim;
    auto X = make_tensor<CT, N>(91, dist, xre, xim);
    Qu_s<dim<N>, DT> D = X;
    emit<CT, DT, N>(out, name, xre, xim, "", D);
}

// real part types
using r206 = Qu<intBits<20>, fracBits<6>>;
using r63 = Qu<intBits<6>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using r6n3 = Qu<intBits<6>, fracBits<-3>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;
using r54 = Qu<intBits<5>, fracBits<4>>;
using r32 = Qu<intBits<3>, fracBits<2>>;
using r104 = Qu<intBits<10>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
using r82z = Qu<intBits<8>, fracBits<2>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
using r73w = Qu<intBits<7>, fracBits<3>, QuMode<RND::ZERO>, OfMode<WRP::TCPL>>;
using r91s = Qu<intBits<9>, fracBits<1>, QuMode<TRN::SMGN>, OfMode<SAT::SMGN>>;
using u44 = Qu<intBits<4>, fracBits<4>, isSigned<false>>;
using w40 = Qu<intBits<30>, fracBits<10>>;                       // 41 storage bits: int64 host part
using s22 = Qu<intBits<2>, fracBits<2>>;
// complex element types
using cw = Qcomplex<r206, r206>;                                  // a wide complex Qgemul result
using c5 = Qcomplex<r63, r6n3>;                                   // configuration 5's element type
using cb = Qcomplex<r54, r32>;                                    // a bias-like operand
using cd = Qcomplex<r104, r82z>;
using cq = Qcomplex<r73w, r91s>;
using cm = Qcomplex<r206, w40>;                                   // 4-byte real part, 8-byte imaginary part
using cu = Qcomplex<u44, r54>;

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    switch (part) {
    case 0: // complex (+|-) complex: default tags, realT / imagT, loose tags, the two-type form, scalars, both orders
        case1<cw, Op<ADD, true>, cb, cd, false>("cadd_tensor_default", 0, out);
        case1<cw, Op<ADD, true>, cb, cw, false>("cadd_tensor_default_sameD", 1, out);
        case1<cw, Op<ADD, true, realT<r104>, imagT<intBits<12>, OfMode<SAT::ZERO>>>, cb, cd, false>("cadd_realT_imagT", 0, out);
        case1<cw, Op<ADD, false, realT<fracBits<2>, QuMode<RND::CONV>>>, cb, cq, false>("cadd_efirst_realT_only", 0, out);
        case1<cw, Op<ADD, true, intBits<14>, fracBits<3>, QuMode<RND::INF>>, cb, cd, false>("cadd_loose_tags_both_parts", 0, out);
        case1<cw, Op<ADD, true, FullPrec>, cb, cm, false>("cadd_fullprec", 0, out);
        case1<cw, Op<SUB, true>, cb, cd, false>("csub_xfirst", 0, out);
        case1<cw, Op<SUB, false>, cb, cd, false>("csub_efirst", 1, out);
        case1<cw, Op<SUB, false, r104, r82z>, cb, cd, true>("csub_efirst_scalar_two_type_form", 0, out);
        case1<cw, Op<ADD, true, r73w, r91s>, cb, cq, true>("cadd_scalar_two_type_form", 1, out);
        case1<c5, Op<ADD, true>, c5, c5, false>("c5_add_same_type", 0, out);
        case1<c5, Op<SUB, true, imagT<intBits<8>, fracBits<0>>>, cb, cq, false>("c5_sub_imagT_only", 0, out);
        case1<cm, Op<ADD, true>, cm, cm, false>("mixed_width_parts_add", 0, out);
        case1<cu, Op<SUB, false>, cu, cq, false>("unsigned_real_part_sub", 0, out);
        break;
    case 1: // complex with REAL operands: mul (both parts), add / sub (the imaginary part is carried over or negated)
        case1<cw, Op<MUL, true>, s22, cd, true>("cmul_real_scalar_default", 1, out);
        case1<cw, Op<MUL, true, realT<intBits<22>, fracBits<6>>, imagT<r104>>, s22, cd, true>("cmul_real_scalar_realT_imagT", 0, out);
        case1<cw, Op<MUL, false>, r32, cd, false>("cmul_real_tensor_efirst", 1, out);
        case1<cw, Op<MUL, true, FullPrec>, r32, cm, false>("cmul_real_tensor_fullprec", 0, out);
        case1<c5, Op<MUL, true, intBits<8>, fracBits<3>>, s22, cq, true>("c5_mul_real_scalar_loose", 0, out);
        case1<cw, Op<ADD, true>, r54, cd, false>("cadd_real_tensor", 0, out);
        case1<cw, Op<ADD, false>, r54, cd, true>("real_scalar_plus_complex", 0, out);
        case1<cw, Op<ADD, true, r104>, r54, cd, false>("cadd_real_tensor_fulltag", 1, out);
        case1<cw, Op<SUB, true>, r54, cd, false>("csub_real_tensor", 0, out);
        case1<cw, Op<SUB, true, intBits<12>, fracBits<2>, QuMode<RND::NEG_INF>>, r54, cq, true>("csub_real_scalar_tags", 0, out);
        case1<cw, Op<SUB, false>, r54, cd, false>("real_tensor_minus_complex", 0, out);
        case1<cw, Op<SUB, false, r104>, r54, cq, true>("real_scalar_minus_complex_fulltag", 1, out);
        case1<c5, Op<SUB, false>, r32, c5, false>("c5_real_minus_complex", 0, out);
        case1<cm, Op<SUB, false>, r54, cm, true>("mixed_width_real_minus_complex", 0, out);
        break;
    case 2: // chains and pure conversions
        case2<cw, Op<MUL, true, realT<intBits<20>, fracBits<6>>, imagT<intBits<20>, fracBits<6>>>, s22, cw, true, Op<ADD, true>, cb, cd, false>("scale_then_cbias", 1, out);
        case2<cw, Op<ADD, true>, cb, cw, false, Op<SUB, false>, r54, cq, true>("cbias_then_real_minus", 0, out);
        case2<c5, Op<ADD, true, FullPrec>, c5, Qcomplex<Qu<intBits<7>, fracBits<3>>, Qu<intBits<7>, fracBits<-3>>>, false, Op<MUL, true>, s22, c5, true>("c5_add_fullprec_then_scale", 0, out);
        case3<cw, Op<MUL, true>, s22, cw, true, Op<ADD, true, realT<r104>>, cb, cd, false, Op<SUB, true>, r32, cq, false>("scale_cbias_real_sub", 1, out);
        case3<cw, Op<ADD, true>, r54, cw, true, Op<SUB, false>, cb, cd, false, Op<MUL, false, imagT<r91s>>, r32, cq, true>("real_add_csub_efirst_scale", 0, out);
        case0<cw, cd>("cconvert_only_narrow", 0, out);
        case0<cw, cw>("cconvert_only_identity", 0, out);
        case0<cm, cq>("cconvert_only_mixed_width", 0, out);
        break;
    default:
        return 2;
    }
    return 0;
}
