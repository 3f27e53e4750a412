// ref_cases_scalar.cpp — TEST INFRASTRUCTURE: scalar truth tables produced by the reference header
// itself (converting constructor, Qmul, Qadd/Qsub, Qreduce), swept exhaustively over small source
// formats.  The reference's own tests cover only 40 rounding cases (test/TRN, test/RND) and no
// overflow mode, Qmul, Qadd or Qreduce at all, so these tables are what pins
// oracle/qoracle.c for those primitives.
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

template <class From, class To>
static void cvt_table(FILE* out, int64_t lo, int64_t hi, int64_t step = 1)
{
    std::fprintf(out, "{\"kind\":\"cvt\",\"from\":%s,\"to\":%s,\"lo\":%lld,\"hi\":%lld,\"step\":%lld,\"y\":[", fmt_json<From>().c_str(), fmt_json<To>().c_str(),
                 (long long)lo, (long long)hi, (long long)step);
    bool first = true;
    for (int64_t x = lo; x <= hi; x += step) {
        From f;
        f.data.data = x;
        To t = f;
        std::fprintf(out, "%s%lld", first ? "" : ",", (long long)t.data.data);
        first = false;
    }
    std::fprintf(out, "]}\n");
}

template <class From, class Q, class O>
static void cvt_targets(FILE* out, int64_t lo, int64_t hi, int64_t step = 1)
{
    cvt_table<From, Qu<intBits<3>, fracBits<2>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<3>, fracBits<2>, isSigned<false>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<1>, fracBits<4>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<5>, fracBits<0>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<6>, fracBits<-2>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<2>, fracBits<7>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
    cvt_table<From, Qu<intBits<0>, fracBits<3>, isSigned<false>, QuMode<Q>, OfMode<O>>>(out, lo, hi, step);
}

template <class From, class O>
static void cvt_modes(FILE* out, int64_t lo, int64_t hi, int64_t step = 1)
{
    cvt_targets<From, RND::POS_INF, O>(out, lo, hi, step);
    cvt_targets<From, RND::NEG_INF, O>(out, lo, hi, step);
    cvt_targets<From, RND::ZERO, O>(out, lo, hi, step);
    cvt_targets<From, RND::INF, O>(out, lo, hi, step);
    cvt_targets<From, RND::CONV, O>(out, lo, hi, step);
    cvt_targets<From, TRN::TCPL, O>(out, lo, hi, step);
    cvt_targets<From, TRN::SMGN, O>(out, lo, hi, step);
}

template <class From>
static void cvt_all(FILE* out, int64_t lo, int64_t hi, int64_t step = 1)
{
    cvt_modes<From, SAT::TCPL>(out, lo, hi, step);
    cvt_modes<From, SAT::ZERO>(out, lo, hi, step);
    cvt_modes<From, SAT::SMGN>(out, lo, hi, step);
    cvt_modes<From, WRP::TCPL>(out, lo, hi, step);
}

// Qmul / Qadd / Qsub with a tag list: prints the RESOLVED result format and a product table
template <class TA, class TB, class... Tags>
static void mul_table(FILE* out, const char* tagname)
{
    using R = decltype(Qmul<Tags...>(std::declval<TA>(), std::declval<TB>()));
    constexpr int Wa = TA::intB + TA::fracB, Wb = TB::intB + TB::fracB;
    int64_t alo = TA::isS ? -(1ll << Wa) : 0, ahi = (1ll << Wa) - 1;
    int64_t blo = TB::isS ? -(1ll << Wb) : 0, bhi = (1ll << Wb) - 1;
    std::fprintf(out, "{\"kind\":\"mul\",\"tags\":\"%s\",\"fa\":%s,\"fb\":%s,\"fr\":%s,\"y\":[", tagname, fmt_json<TA>().c_str(), fmt_json<TB>().c_str(), fmt_json<R>().c_str());
    bool first = true;
    for (int64_t a = alo; a <= ahi; ++a)
        for (int64_t b = blo; b <= bhi; ++b) {
            TA x; TB y;
            x.data.data = a; y.data.data = b;
            auto r = Qmul<Tags...>(x, y);
            std::fprintf(out, "%s%lld", first ? "" : ",", (long long)r.data.data);
            first = false;
        }
    std::fprintf(out, "]}\n");
}

template <bool SUB, class TA, class TB, class... Tags>
static void add_table(FILE* out, const char* tagname)
{
    using R = std::conditional_t<SUB, decltype(Qsub<Tags...>(std::declval<TA>(), std::declval<TB>())), decltype(Qadd<Tags...>(std::declval<TA>(), std::declval<TB>()))>;
    constexpr int Wa = TA::intB + TA::fracB, Wb = TB::intB + TB::fracB;
    int64_t alo = TA::isS ? -(1ll << Wa) : 0, ahi = (1ll << Wa) - 1;
    int64_t blo = TB::isS ? -(1ll << Wb) : 0, bhi = (1ll << Wb) - 1;
    std::fprintf(out, "{\"kind\":\"%s\",\"tags\":\"%s\",\"fa\":%s,\"fb\":%s,\"fr\":%s,\"y\":[", SUB ? "sub" : "add", tagname, fmt_json<TA>().c_str(), fmt_json<TB>().c_str(),
                 fmt_json<R>().c_str());
    bool first = true;
    for (int64_t a = alo; a <= ahi; ++a)
        for (int64_t b = blo; b <= bhi; ++b) {
            TA x; TB y;
            x.data.data = a; y.data.data = b;
            int64_t r;
            if constexpr (SUB) r = Qsub<Tags...>(x, y).data.data;
            else r = Qadd<Tags...>(x, y).data.data;
            std::fprintf(out, "%s%lld", first ? "" : ",", (long long)r);
            first = false;
        }
    std::fprintf(out, "]}\n");
}

// Qreduce over a vector of length LEN filled from the synthetic generator, several seeds
template <class T, size_t LEN, class... Levels>
static void reduce_table(FILE* out, const char* name)
{
    std::string lv = "[";
    ((lv += (lv.size() > 1 ? "," : "") + fmt_json<Levels>()), ...);
    lv += "]";
    std::fprintf(out, "{\"kind\":\"reduce\",\"name\":\"%s\",\"fin\":%s,\"levels\":%s,\"len\":%zu,\"dist\":0,\"seeds\":[1,2,3,4,5,6,7,8],\"y\":[", name, fmt_json<T>().c_str(),
                 lv.c_str(), LEN);
    for (uint64_t seed = 1; seed <= 8; ++seed) {
        Qu_s<dim<LEN>, T> v;
        for (size_t i = 0; i < LEN; ++i) v[i].data.data = synth<T>(seed, 0, i, 0);
        auto r = Qreduce<Levels...>(v);
        using R = decltype(r);
        std::fprintf(out, "%s[%lld,%s]", seed > 1 ? "," : "", (long long)r.data.data, fmt_json<R>().c_str());
    }
    std::fprintf(out, "]}\n");
}

// the same with the format's raw minimum -2^W forced into two elements: a signed SAT::SMGN element type holds it (fill() can
// produce it) although symmetric-saturation arithmetic never does; Qreduce adds it as it is
template <class T, size_t LEN, class... Levels>
static void reduce_min_table(FILE* out, const char* name)
{
    std::string lv = "[";
    ((lv += (lv.size() > 1 ? "," : "") + fmt_json<Levels>()), ...);
    lv += "]";
    constexpr int64_t rmin = T::isS ? -(int64_t(1) << (T::intB + T::fracB)) : 0;
    std::fprintf(out, "{\"kind\":\"reduce\",\"name\":\"%s\",\"fin\":%s,\"levels\":%s,\"len\":%zu,\"dist\":0,\"seeds\":[1,2,3,4,5,6,7,8],\"min_at\":[0,%zu],\"y\":[", name,
                 fmt_json<T>().c_str(), lv.c_str(), LEN, LEN / 2);
    for (uint64_t seed = 1; seed <= 8; ++seed) {
        Qu_s<dim<LEN>, T> v;
        for (size_t i = 0; i < LEN; ++i) v[i].data.data = synth<T>(seed, 0, i, 0);
        v[0].data.data = rmin;
        v[LEN / 2].data.data = rmin;
        auto r = Qreduce<Levels...>(v);
        using R = decltype(r);
        std::fprintf(out, "%s[%lld,%s]", seed > 1 ? "," : "", (long long)r.data.data, fmt_json<R>().c_str());
    }
    std::fprintf(out, "]}\n");
}

// ... with the raw minimum as the LAST element: for odd LEN it is the odd leftover of level 0, which the reference copies into
// the level-0 buffer (a same-type copy when level 0's type is the element type: the raw minimum survives it)
template <class T, size_t LEN, class... Levels>
static void reduce_min_last_table(FILE* out, const char* name)
{
    std::string lv = "[";
    ((lv += (lv.size() > 1 ? "," : "") + fmt_json<Levels>()), ...);
    lv += "]";
    constexpr int64_t rmin = T::isS ? -(int64_t(1) << (T::intB + T::fracB)) : 0;
    std::fprintf(out, "{\"kind\":\"reduce\",\"name\":\"%s\",\"fin\":%s,\"levels\":%s,\"len\":%zu,\"dist\":1,\"seeds\":[1,2,3,4,5,6,7,8],\"min_at\":[%zu],\"y\":[", name,
                 fmt_json<T>().c_str(), lv.c_str(), LEN, LEN - 1);
    for (uint64_t seed = 1; seed <= 8; ++seed) {
        Qu_s<dim<LEN>, T> v;
        for (size_t i = 0; i < LEN; ++i) v[i].data.data = synth<T>(seed, 1, i, 0);   // small values: the sums stay in range
        v[LEN - 1].data.data = rmin;
        auto r = Qreduce<Levels...>(v);
        using R = decltype(r);
        std::fprintf(out, "%s[%lld,%s]", seed > 1 ? "," : "", (long long)r.data.data, fmt_json<R>().c_str());
    }
    std::fprintf(out, "]}\n");
}

// Qu_s(double): loadFromDouble into a 2400-bit buffer, then the type's own fracConvert / intConvert
// (QuBLAS.h:2387-2393, :663-749).  Doubles are printed as hex-float so the table is exact.
template <class T>
static void from_double_table(FILE* out)
{
    static const double vals[] = {0.0, 1.0, -1.0, 1.25, -1.25, 1.5, -1.5, 1.75, -1.75, 0.5, -0.5, 0.125, -0.125, 0.0625, -0.0625, 0.1875,
                                  -0.1875, 0.3125, -0.3125, 3.999, -3.999, 7.5, -7.5, 7.75, -8.0, -8.25, 15.999, 16.0, -16.0, 20.0, -20.0,
                                  63.5, 64.0, -64.0, -64.5, 100.0, -100.0, 255.0, 256.0, 1e6, -1e6, 1e-9, -1e-9, 0.1, -0.1, 0.3, -0.3, 2.71828182845904,
                                  -3.14159265358979, 0.4999999999999999, -0.5000000000000001, 12345.6789, -54321.125, 1.0000000000000002,
                                  5.0e-324, -5.0e-324, 1.7e308, -1.7e308, 0.9999999999999999, 127.99609375, -128.00390625};
    std::fprintf(out, "{\"kind\":\"from_double\",\"to\":%s,\"x\":[", fmt_json<T>().c_str());
    bool first = true;
    for (double v : vals) { std::fprintf(out, "%s\"%a\"", first ? "" : ",", v); first = false; }
    std::fprintf(out, "],\"y\":[");
    first = true;
    for (double v : vals) { T t = v; std::fprintf(out, "%s%lld", first ? "" : ",", (long long)t.data.data); first = false; }
    std::fprintf(out, "]}\n");
}

template <class Q, class O>
static void from_double_targets(FILE* out)
{
    from_double_table<Qu<intBits<3>, fracBits<2>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out);
    from_double_table<Qu<intBits<3>, fracBits<2>, isSigned<false>, QuMode<Q>, OfMode<O>>>(out);
    from_double_table<Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out);
    from_double_table<Qu<intBits<6>, fracBits<-3>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out);
    from_double_table<Qu<intBits<4>, fracBits<3>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out);
    from_double_table<Qu<intBits<20>, fracBits<30>, isSigned<true>, QuMode<Q>, OfMode<O>>>(out);
}

template <class O>
static void from_double_modes(FILE* out)
{
    from_double_targets<RND::POS_INF, O>(out);
    from_double_targets<RND::NEG_INF, O>(out);
    from_double_targets<RND::ZERO, O>(out);
    from_double_targets<RND::INF, O>(out);
    from_double_targets<RND::CONV, O>(out);
    from_double_targets<TRN::TCPL, O>(out);
    from_double_targets<TRN::SMGN, O>(out);
}

int main(int argc, char** argv)
{
    int part = argc > 1 ? std::atoi(argv[1]) : 0;
    FILE* out = stdout;
    using s35 = Qu<intBits<3>, fracBits<5>>;                   // 9-bit signed source, exhaustive
    using u26 = Qu<intBits<2>, fracBits<6>, isSigned<false>>;  // 8-bit unsigned source, exhaustive
    using s1010 = Qu<intBits<10>, fracBits<10>>;               // the reference tests' Low_t width
    using s3030 = Qu<intBits<30>, fracBits<30>>;               // the reference tests' Mid_t width
    switch (part) {
    case 0:
        cvt_all<s35>(out, -256, 255);
        break;
    case 1:
        cvt_all<u26>(out, 0, 255);
        cvt_all<s1010>(out, -(1 << 20), (1 << 20) - 1, 4099); // strided sweep of the 21-bit source
        break;
    case 2:
        cvt_all<s3030>(out, -(1ll << 60), (1ll << 60) - 1, (1ll << 52) + 12345678901ll); // strided sweep of the 61-bit source
        break;
    case 3: {
        using a = Qu<intBits<2>, fracBits<3>>;                                             // 6 bits signed
        using b = Qu<intBits<3>, fracBits<1>, isSigned<false>>;                            // 4 bits unsigned
        using c = Qu<intBits<1>, fracBits<4>, QuMode<RND::CONV>, OfMode<SAT::ZERO>>;       // other modes
        using d = Qu<intBits<4>, fracBits<-1>>;                                            // negative frac
        mul_table<a, a>(out, "default");
        mul_table<a, b>(out, "default");
        mul_table<a, c>(out, "default");     // modes differ -> defaults
        mul_table<c, c>(out, "default");     // modes agree -> kept
        mul_table<a, d>(out, "default");
        mul_table<a, a, FullPrec>(out, "FullPrec"); // min*min overflows by one LSB
        mul_table<a, b, FullPrec>(out, "FullPrec");
        mul_table<a, a, intBits<5>, fracBits<6>>(out, "int5_frac6");
        mul_table<a, a, fracBits<1>, QuMode<RND::POS_INF>>(out, "frac1_POS_INF");
        mul_table<a, a, fracBits<1>, QuMode<RND::NEG_INF>>(out, "frac1_NEG_INF");
        mul_table<a, a, fracBits<2>, QuMode<RND::ZERO>, OfMode<SAT::SMGN>>(out, "frac2_ZERO_SMGN");
        mul_table<a, a, fracBits<2>, QuMode<RND::INF>, OfMode<WRP::TCPL>>(out, "frac2_INF_WRP");
        mul_table<a, a, fracBits<3>, QuMode<RND::CONV>, OfMode<SAT::ZERO>>(out, "frac3_CONV_ZERO");
        mul_table<a, a, fracBits<2>, QuMode<TRN::SMGN>, isSigned<false>>(out, "frac2_SMGN_unsigned");
        mul_table<a, b, c>(out, "fulltype_c");
        mul_table<b, b, OfMode<WRP::TCPL>, intBits<2>>(out, "unsigned_wrap_int2");
        add_table<false, a, a>(out, "default");
        add_table<false, a, b>(out, "default");
        add_table<false, a, c>(out, "default");
        add_table<false, a, d>(out, "default");
        add_table<false, a, a, FullPrec>(out, "FullPrec");
        add_table<false, a, b, c>(out, "fulltype_c");
        add_table<false, a, a, fracBits<1>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>(out, "frac1_CONV_SMGN");
        add_table<false, a, d, fracBits<1>, QuMode<TRN::SMGN>, OfMode<WRP::TCPL>, intBits<3>>(out, "frac1_SMGN_WRP_int3");
        add_table<true, a, a>(out, "default");
        add_table<true, a, b>(out, "default");
        add_table<true, b, a, isSigned<false>>(out, "unsigned");
        add_table<true, a, d, fracBits<0>, QuMode<RND::ZERO>, OfMode<SAT::ZERO>>(out, "frac0_ZERO_ZERO");
        add_table<true, b, b, OfMode<WRP::TCPL>>(out, "unsigned_wrap");
        break;
    }
    case 4: {
        using e = Qu<intBits<4>, fracBits<3>>;
        using z = Qu<intBits<8>, fracBits<8>, isSigned<true>, QuMode<TRN::TCPL>, OfMode<SAT::ZERO>>;
        using nar = Qu<intBits<5>, fracBits<2>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using wide = Qu<intBits<12>, fracBits<5>, QuMode<RND::ZERO>>;
        reduce_table<e, 1>(out, "len1_default");
        reduce_table<e, 1, nar>(out, "len1_nar");
        reduce_table<e, 2, nar>(out, "len2_nar");
        reduce_table<e, 3, nar, wide>(out, "len3_nar_wide");
        reduce_table<e, 4, nar, wide>(out, "len4_nar_wide");
        reduce_table<e, 5, nar, wide>(out, "len5_nar_wide");
        reduce_table<e, 6, nar, wide>(out, "len6_nar_wide");
        reduce_table<e, 7, wide, nar>(out, "len7_wide_nar");
        reduce_table<e, 8>(out, "len8_default");
        reduce_table<e, 16, wide>(out, "len16_wide");
        reduce_table<z, 16>(out, "len16_z_default");
        reduce_table<z, 512>(out, "len512_z_default");
        reduce_table<e, 512, nar, wide>(out, "len512_nar_wide");
        reduce_table<e, 1000, wide, nar>(out, "len1000_wide_nar");
        reduce_table<e, 1000>(out, "len1000_default");
        break;
    }
    case 5:
        from_double_modes<SAT::TCPL>(out);
        from_double_modes<SAT::ZERO>(out);
        from_double_modes<SAT::SMGN>(out);
        from_double_modes<WRP::TCPL>(out);
        break;
    case 6: {
        // 32-bit edges of WRP::TCPL: the unsigned mask of exactly 32 value bits is ArbiInt<32>::allOnes() = -1
        // in the reference, so nothing is masked; the signed 32-storage-bit target wraps through its int32 storage.
        using s400 = Qu<intBits<40>, fracBits<0>>;
        using u32w = Qu<intBits<32>, fracBits<0>, isSigned<false>, OfMode<WRP::TCPL>>;
        using u1616w = Qu<intBits<16>, fracBits<16>, isSigned<false>, OfMode<WRP::TCPL>>;
        using u31w = Qu<intBits<31>, fracBits<0>, isSigned<false>, OfMode<WRP::TCPL>>;
        using u33w = Qu<intBits<33>, fracBits<0>, isSigned<false>, OfMode<WRP::TCPL>>;
        using s31w = Qu<intBits<31>, fracBits<0>, isSigned<true>, OfMode<WRP::TCPL>>;
        using s30w = Qu<intBits<30>, fracBits<0>, isSigned<true>, OfMode<WRP::TCPL>>;
        const int64_t lo = -(1ll << 36), hi = (1ll << 36), step = (1ll << 27) + 12345;
        cvt_table<s400, u32w>(out, lo, hi, step);
        cvt_table<s400, u32w>(out, -16, 16);
        cvt_table<s400, u1616w>(out, lo, hi, step);
        cvt_table<s400, u31w>(out, lo, hi, step);
        cvt_table<s400, u33w>(out, lo, hi, step);
        cvt_table<s400, s31w>(out, lo, hi, step);
        cvt_table<s400, s30w>(out, lo, hi, step);
        break;
    }
    case 7: {
        // Qreduce on a signed SAT::SMGN element type whose raw minimum is present (and, for comparison, the same on TCPL / ZERO types)
        using sm = Qu<intBits<3>, fracBits<4>, OfMode<SAT::SMGN>>;
        using st = Qu<intBits<3>, fracBits<4>>;
        using sz = Qu<intBits<3>, fracBits<4>, OfMode<SAT::ZERO>>;
        using lz = Qu<intBits<4>, fracBits<6>, OfMode<SAT::ZERO>>;
        using lw = Qu<intBits<9>, fracBits<4>>;
        using lsm = Qu<intBits<5>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::SMGN>>;
        reduce_min_table<sm, 2>(out, "smgn_min_len2_default");
        reduce_min_table<sm, 5>(out, "smgn_min_len5_default");
        reduce_min_table<sm, 16>(out, "smgn_min_len16_default");
        reduce_min_table<sm, 64>(out, "smgn_min_len64_default");
        reduce_min_table<sm, 16, lz>(out, "smgn_min_len16_lz");
        reduce_min_table<sm, 64, lw>(out, "smgn_min_len64_lw");
        reduce_min_table<sm, 64, lsm, lw>(out, "smgn_min_len64_lsm_lw");
        reduce_min_table<sm, 512, lw>(out, "smgn_min_len512_lw");
        reduce_min_table<sm, 1000, lsm, lw>(out, "smgn_min_len1000_lsm_lw");
        reduce_min_table<st, 64>(out, "tcpl_min_len64_default");
        reduce_min_table<sz, 64, lz>(out, "zero_min_len64_lz");
        break;
    }
    case 8: {
        // the raw minimum of a signed SAT::SMGN element type as the odd leftover of level 0
        using sm = Qu<intBits<3>, fracBits<4>, OfMode<SAT::SMGN>>;
        using st = Qu<intBits<3>, fracBits<4>>;
        using lw = Qu<intBits<9>, fracBits<4>>;
        using lsm = Qu<intBits<5>, fracBits<3>, QuMode<RND::POS_INF>, OfMode<SAT::SMGN>>;
        using lsw = Qu<intBits<6>, fracBits<4>, OfMode<SAT::SMGN>>;
        reduce_min_last_table<sm, 1>(out, "smgn_last_len1_default");
        reduce_min_last_table<sm, 3>(out, "smgn_last_len3_default");
        reduce_min_last_table<sm, 5>(out, "smgn_last_len5_default");
        reduce_min_last_table<sm, 7>(out, "smgn_last_len7_default");
        reduce_min_last_table<sm, 9>(out, "smgn_last_len9_default");
        reduce_min_last_table<sm, 17>(out, "smgn_last_len17_default");
        reduce_min_last_table<sm, 33>(out, "smgn_last_len33_default");
        reduce_min_last_table<sm, 65>(out, "smgn_last_len65_default");
        reduce_min_last_table<sm, 999>(out, "smgn_last_len999_default");
        reduce_min_last_table<sm, 5, sm>(out, "smgn_last_len5_sm");
        reduce_min_last_table<sm, 9, sm, lw>(out, "smgn_last_len9_sm_lw");
        reduce_min_last_table<sm, 5, lw>(out, "smgn_last_len5_lw");
        reduce_min_last_table<sm, 5, lsm, lw>(out, "smgn_last_len5_lsm_lw");
        reduce_min_last_table<sm, 17, lsw>(out, "smgn_last_len17_lsw");
        reduce_min_last_table<sm, 6>(out, "smgn_last_len6_default");
        reduce_min_last_table<sm, 10>(out, "smgn_last_len10_default");
        reduce_min_last_table<st, 5>(out, "tcpl_last_len5_default");
        break;
    }
    case 9: {
        // WRP::TCPL_SAT<N>: intConvert returns its input (QuBLAS.h:2336-2344), the assignment into the target's storage then narrows
        // it to the storage word (ArbiInt<M <= 32> keeps an int32_t, <= 64 an int64_t, never masked to M bits: :353, :431-441)
        using s400 = Qu<intBits<40>, fracBits<0>>;
        using s123 = Qu<intBits<12>, fracBits<3>>;
        using t43 = Qu<intBits<4>, fracBits<3>, OfMode<WRP::TCPL_SAT<2>>>;
        using t43r = Qu<intBits<4>, fracBits<1>, QuMode<RND::POS_INF>, OfMode<WRP::TCPL_SAT<1>>>;
        using t400 = Qu<intBits<40>, fracBits<2>, OfMode<WRP::TCPL_SAT<3>>>;
        using u200 = Qu<intBits<20>, fracBits<0>, isSigned<false>, OfMode<WRP::TCPL_SAT<1>>>;
        const int64_t lo = -(1ll << 36), hi = (1ll << 36), step = (1ll << 27) + 12345;
        cvt_table<s400, t43>(out, lo, hi, step);
        cvt_table<s400, t43>(out, -300, 300, 7);
        cvt_table<s400, t43r>(out, lo, hi, step);
        cvt_table<s400, u200>(out, lo, hi, step);
        cvt_table<s123, t43>(out, -32768, 32767, 97);
        cvt_table<s123, t43r>(out, -32768, 32767, 97);
        cvt_table<s123, t400>(out, -32768, 32767, 97);
        break;
    }
    case 10: {
        // the VARIADIC overload, Qreduce<L...>(q1, q2, ...) (readme.md:62; QuBLAS.h:4924-4951): any number of scalars of any types;
        // an odd leftover is added AFTER the recursion over the pair sums, at the current level's type
        using t1 = Qu<intBits<4>, fracBits<3>>;
        using t2 = Qu<intBits<6>, fracBits<1>, QuMode<RND::POS_INF>, OfMode<SAT::SMGN>>;
        using nar = Qu<intBits<5>, fracBits<2>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;
        using wide = Qu<intBits<12>, fracBits<5>, QuMode<RND::ZERO>>;
        auto emit = [&](const char* name, auto r, std::initializer_list<int64_t> raws, std::initializer_list<std::string> fmts, const char* levels) {
            using R = decltype(r);
            std::fprintf(out, "{\"kind\":\"reduce_variadic\",\"name\":\"%s\",\"levels\":%s,\"x\":[", name, levels);
            size_t i = 0;
            for (int64_t v : raws) std::fprintf(out, "%s%lld", i++ ? "," : "", (long long)v);
            std::fprintf(out, "],\"fx\":[");
            i = 0;
            for (const auto& f : fmts) std::fprintf(out, "%s%s", i++ ? "," : "", f.c_str());
            std::fprintf(out, "],\"y\":%lld,\"fr\":%s}\n", (long long)r.data.data, fmt_json<R>().c_str());
        };
        const std::string f1 = fmt_json<t1>(), f2 = fmt_json<t2>(), lv1 = "[" + fmt_json<nar>() + "]", lv2 = "[" + fmt_json<nar>() + "," + fmt_json<wide>() + "]";
        for (int seed = 0; seed < 6; ++seed) {
            t1 a, c, e, g; t2 b, d, f;
            int64_t va = synth<t1>(seed + 1, 0, 0, 0), vb = synth<t2>(seed + 1, 0, 1, 0), vc = synth<t1>(seed + 1, 0, 2, 0), vd = synth<t2>(seed + 1, 0, 3, 0),
                    ve = synth<t1>(seed + 1, 0, 4, 0), vf = synth<t2>(seed + 1, 0, 5, 0), vg = synth<t1>(seed + 1, 0, 6, 0);
            a.data.data = va; b.data.data = vb; c.data.data = vc; d.data.data = vd; e.data.data = ve; f.data.data = vf; g.data.data = vg;
            emit("var4_readme", Qreduce<t1>(a, b, a, b), {va, vb, va, vb}, {f1, f2, f1, f2}, ("[" + f1 + "]").c_str());
            emit("var3_default", Qreduce<>(a, b, c), {va, vb, vc}, {f1, f2, f1}, "[]");
            emit("var3_nar_wide", Qreduce<nar, wide>(a, b, c), {va, vb, vc}, {f1, f2, f1}, lv2.c_str());
            emit("var5_nar_wide", Qreduce<nar, wide>(a, b, c, d, e), {va, vb, vc, vd, ve}, {f1, f2, f1, f2, f1}, lv2.c_str());
            emit("var6_nar_wide", Qreduce<nar, wide>(a, b, c, d, e, f), {va, vb, vc, vd, ve, vf}, {f1, f2, f1, f2, f1, f2}, lv2.c_str());
            emit("var7_nar_wide", Qreduce<nar, wide>(a, b, c, d, e, f, g), {va, vb, vc, vd, ve, vf, vg}, {f1, f2, f1, f2, f1, f2, f1}, lv2.c_str());
            emit("var7_nar", Qreduce<nar>(a, b, c, d, e, f, g), {va, vb, vc, vd, ve, vf, vg}, {f1, f2, f1, f2, f1, f2, f1}, lv1.c_str());
            emit("var5_list", Qreduce<TypeList<wide, nar>>(a, b, c, d, e), {va, vb, vc, vd, ve}, {f1, f2, f1, f2, f1}, ("[" + fmt_json<wide>() + "," + fmt_json<nar>() + "]").c_str());
        }
        break;
    }
    default:
        return 2;
    }
    return 0;
}
