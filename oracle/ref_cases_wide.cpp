// ref_cases_wide.cpp — TEST INFRASTRUCTURE: what the reference header computes once intermediates leave 64 bits
// (ArbiInt<N > 64>, a little-endian std::array<uint64_t, n>, QuBLAS.h:566-912; wide + - * :914-1363; shifts :1485-1701).
// Round 2 refused every descriptor with an intermediate beyond 62 bits; these tables pin the 128-bit arithmetic of
// oracle/qoracle.c and of the engine's wide kernels: converting constructor across the 64-bit boundary, Qmul / Qadd / Qsub
// with wide results, Qreduce with wide levels, and Qgemul compositions whose products / sums / C are wide.
// Values are printed as decimal integers of any size (the fixtures' readers take JSON big integers as they are).
// Usage: ref_cases_wide <part> > out.jsonl
#include "ref_driver.hpp"

#include <cstdlib>

using namespace refdrv;

// a deterministic value of format T's full range, from two generator words (own code; the fixtures carry the values themselves)
template <class T>
static raw_t wide_value(uint64_t seed, uint64_t i, int kind)
{
    constexpr int W = T::intB + T::fracB;
    const raw_t hi = (raw_t(1) << W) - 1, lo = T::isS ? -(raw_t(1) << W) : 0;
    if (kind == 0) return hi;
    if (kind == 1) return lo;
    if (kind == 2) return 0;
    if (kind == 3) return T::isS ? raw_t(-1) : raw_t(1);
    const unsigned __int128 span = (unsigned __int128)(hi - lo) + 1;
    unsigned __int128 u = ((unsigned __int128)qrand(seed, 2 * i) << 64) | qrand(seed, 2 * i + 1);
    if (kind == 4) u >>= (127 - W / 2 > 0 ? 127 - W / 2 : 0);   // small magnitudes
    return lo + raw_t(span ? u % span : u);
}

template <class From, class To>
static void cvt_table_w(FILE* out, uint64_t seed, int n = 24)
{
    std::fprintf(out, "{\"kind\":\"cvt\",\"from\":%s,\"to\":%s,\"x\":[", fmt_json<From>().c_str(), fmt_json<To>().c_str());
    std::vector<raw_t> xs;
    for (int k = 0; k < 4; ++k) xs.push_back(wide_value<From>(seed, 0, k));
    for (int i = 0; i < n; ++i) xs.push_back(wide_value<From>(seed, i, i % 3 == 2 ? 4 : 5));
    // ties and near-ties of the rounding step, where there is one
    if constexpr (From::fracB > To::fracB) {
        constexpr int d = From::fracB - To::fracB;
        const raw_t half = raw_t(1) << (d - 1);
        for (raw_t h : {raw_t(0), raw_t(1), raw_t(2), raw_t(-1), raw_t(-2), raw_t(5), raw_t(-5)})
            for (raw_t e : {raw_t(-1), raw_t(0), raw_t(1)}) xs.push_back(h * (half * 2) + half + e);
    }
    {
        // values whose ROUNDED image lies around +-2^63 .. +-2^64: where the reference's comparison of a multi-word value with a
        // one-word bound (operator<=>, QuBLAS.h:1781-1793) looks at the low word as a signed number
        constexpr int d = From::fracB > To::fracB ? From::fracB - To::fracB : 0;
        constexpr int ls = From::fracB < To::fracB ? To::fracB - From::fracB : 0;
        constexpr int W = From::intB + From::fracB;
        const raw_t hi = (raw_t(1) << W) - 1, lo = From::isS ? -(raw_t(1) << W) : 0;
        const raw_t b63 = raw_t(1) << 63, b64 = raw_t(1) << 64;
        for (raw_t h : {b63 - 1, b63, b63 + 5, b64 - 1, b64, b64 + 3, 3 * b63, -b63, -b63 - 1, -b63 - 7, -b64 + 1, -b64, -b64 - 1, -3 * b63})
            for (int e = 0; e < (d ? 3 : 1); ++e) {
                if (d + 66 >= 127) continue;
                raw_t x = d ? h * (raw_t(1) << d) + (e == 0 ? 0 : e == 1 ? (raw_t(1) << (d - 1)) : (raw_t(1) << (d - 1)) - 1) : (ls ? h / (raw_t(1) << (ls < 60 ? ls : 60)) : h);
                if (x >= lo && x <= hi) xs.push_back(x);
            }
    }
    for (size_t i = 0; i < xs.size(); ++i) std::fprintf(out, "%s%s", i ? "," : "", dec(xs[i]).c_str());
    std::fprintf(out, "],\"y\":[");
    for (size_t i = 0; i < xs.size(); ++i) {
        From f;
        put_words(f.data, xs[i]);
        To t = f;
        std::fprintf(out, "%s%s", i ? "," : "", dec(get_words(t.data)).c_str());
    }
    std::fprintf(out, "]}\n");
}

template <int OP, class TA, class TB, class... Tags>   // OP 0 mul, 1 add, 2 sub
static void op_table_w(FILE* out, const char* tagname, uint64_t seed, int n = 40)
{
    auto apply = [](const TA& x, const TB& y) {
        if constexpr (OP == 0) return Qmul<Tags...>(x, y);
        else if constexpr (OP == 1) return Qadd<Tags...>(x, y);
        else return Qsub<Tags...>(x, y);
    };
    using R = decltype(apply(std::declval<TA>(), std::declval<TB>()));
    std::fprintf(out, "{\"kind\":\"%s\",\"tags\":\"%s\",\"fa\":%s,\"fb\":%s,\"fr\":%s,\"xy\":[", OP == 0 ? "mul" : OP == 1 ? "add" : "sub", tagname,
                 fmt_json<TA>().c_str(), fmt_json<TB>().c_str(), fmt_json<R>().c_str());
    std::vector<std::pair<raw_t, raw_t>> xs;
    for (int ka = 0; ka < 4; ++ka)
        for (int kb = 0; kb < 4; ++kb) xs.push_back({wide_value<TA>(seed, 0, ka), wide_value<TB>(seed, 0, kb)});
    for (int i = 0; i < n; ++i) xs.push_back({wide_value<TA>(seed, i, i % 4 == 3 ? 4 : 5), wide_value<TB>(seed + 77, i, i % 5 == 4 ? 4 : 5)});
    for (size_t i = 0; i < xs.size(); ++i) std::fprintf(out, "%s[%s,%s]", i ? "," : "", dec(xs[i].first).c_str(), dec(xs[i].second).c_str());
    std::fprintf(out, "],\"y\":[");
    for (size_t i = 0; i < xs.size(); ++i) {
        TA x; TB y;
        put_words(x.data, xs[i].first);
        put_words(y.data, xs[i].second);
        R r = apply(x, y);
        std::fprintf(out, "%s%s", i ? "," : "", dec(get_words(r.data)).c_str());
    }
    std::fprintf(out, "]}\n");
}

// Qreduce of full-range elements through wide level types
template <class T, size_t LEN, class... Levels>
static void reduce_table_w(FILE* out, const char* name)
{
    std::string lv = "[";
    ((lv += (lv.size() > 1 ? "," : "") + fmt_json<Levels>()), ...);
    lv += "]";
    std::fprintf(out, "{\"kind\":\"reduce\",\"name\":\"%s\",\"fin\":%s,\"levels\":%s,\"len\":%zu,\"dist\":0,\"seeds\":[1,2,3,4],\"y\":[", name, fmt_json<T>().c_str(),
                 lv.c_str(), LEN);
    for (uint64_t seed = 1; seed <= 4; ++seed) {
        Qu_s<dim<LEN>, T> v;
        for (size_t i = 0; i < LEN; ++i) put_words(v[i].data, raw_t(synth<T>(seed, 0, i, 0)));
        auto r = Qreduce<Levels...>(v);
        using R = decltype(r);
        std::fprintf(out, "%s[%s,%s]", seed > 1 ? "," : "", dec(get_words(r.data)).c_str(), fmt_json<R>().c_str());
    }
    std::fprintf(out, "]}\n");
}

static Inputs syn(int dist, uint64_t sa = 1, uint64_t sb = 2)
{
    Inputs in;
    in.dist = dist; in.seedA = sa; in.seedB = sb;
    return in;
}

// ---- which tables exist ----
// The reference's multi-word code does not COMPILE for every combination of widths and modes (e.g. operator| of two wide
// integers of different sizes is ill-formed, QuBLAS.h:1946-1950, and with it conversions that mask one wide value with
// another).  A combination the reference cannot compile is not something a user of the reference can run, so it is not
// something to reproduce: oracle/probe_wide.py compiles every table below on its own (-DPROBE=<id> -fsyntax-only) and writes
// ref_cases_wide_enabled.inc — one flag per table id — which this file includes; a table whose flag is 0 is a discarded
// statement of the templates below and is never instantiated (it prints a "not_compiled" record instead).
enum { N_IDS = 480 };
#ifdef PROBE
static constexpr bool enabled(int id) { return id == PROBE; }
#else
static constexpr bool EN[N_IDS] = {
#include "ref_cases_wide_enabled.inc"
};
static constexpr bool enabled(int id) { return id >= 0 && id < N_IDS && EN[id]; }
#endif
static void not_compiled(FILE* out, int id, const char* what)
{
    std::string w(what);
    for (char& c : w) if (c == '"') c = '\'';
    std::fprintf(out, "{\"kind\":\"not_compiled\",\"id\":%d,\"what\":\"%s\"}\n", id, w.c_str());
}
#define CASE(ID, ...)                                                                                               \
    do {                                                                                                            \
        if constexpr (D == 0 && enabled(ID)) { __VA_ARGS__; }                                                       \
        else not_compiled(out, int(ID), #__VA_ARGS__);                                                              \
    } while (0)

using q1516 = Qu<intBits<15>, fracBits<16>>;                        // signed 32-bit words
using q3132 = Qu<intBits<31>, fracBits<32>>;                        // their exact product, 64 storage bits
using q3232 = Qu<intBits<32>, fracBits<32>>;                        // 65 storage bits: the first two-word type
using q4332 = Qu<intBits<43>, fracBits<32>>;                        // 76: exact sums of 4096 such products
using q4040 = Qu<intBits<40>, fracBits<40>>;                        // 81
using q6059 = Qu<intBits<60>, fracBits<59>>;                        // 120
using u5050 = Qu<intBits<50>, fracBits<50>, isSigned<false>>;       // unsigned, 101 storage bits
using q6300 = Qu<intBits<63>, fracBits<0>>;                         // exactly 64 storage bits
using q6400 = Qu<intBits<64>, fracBits<0>>;                         // 65
using q2030 = Qu<intBits<20>, fracBits<30>>;

// all 7 QuModes x 4 OfModes of one (source, target geometry): table ids BASE .. BASE + 27
template <int D, int BASE, class From, class I, class F, class S>
static void cvt_all_w(FILE* out, uint64_t seed)
{
#define CVT_O(n, Q, O) CASE(BASE + n, cvt_table_w<From, Qu<I, F, S, QuMode<Q>, OfMode<O>>>(out, seed))
#define CVT_Q(b, O) CVT_O(b, RND::POS_INF, O); CVT_O(b + 1, RND::NEG_INF, O); CVT_O(b + 2, RND::ZERO, O); CVT_O(b + 3, RND::INF, O); CVT_O(b + 4, RND::CONV, O); CVT_O(b + 5, TRN::TCPL, O); CVT_O(b + 6, TRN::SMGN, O)
    CVT_Q(0, SAT::TCPL); CVT_Q(7, SAT::ZERO); CVT_Q(14, SAT::SMGN); CVT_Q(21, WRP::TCPL);
#undef CVT_Q
#undef CVT_O
}

template <int D>
static int run_part(int part, FILE* out)
{
    switch (part) {
    case 0:   // converting constructor: wide -> narrow, wide -> wide, narrow -> wide, across the word boundary
        cvt_all_w<D, 0, q4040, intBits<10>, fracBits<5>, isSigned<true>>(out, 11);
        cvt_all_w<D, 28, q4040, intBits<30>, fracBits<20>, isSigned<true>>(out, 12);
        cvt_all_w<D, 56, q4040, intBits<38>, fracBits<33>, isSigned<true>>(out, 13);      // wide target, rounding by 7
        cvt_all_w<D, 84, q4040, intBits<20>, fracBits<8>, isSigned<false>>(out, 14);
        cvt_all_w<D, 112, q6059, intBits<40>, fracBits<30>, isSigned<true>>(out, 15);
        cvt_all_w<D, 140, q6059, intBits<12>, fracBits<3>, isSigned<true>>(out, 16);      // rounding by 56
        cvt_all_w<D, 168, u5050, intBits<45>, fracBits<40>, isSigned<false>>(out, 17);
        break;
    case 1:
        cvt_all_w<D, 196, q4332, intBits<15>, fracBits<16>, isSigned<true>>(out, 18);     // the Q15.16 accumulate-and-store epilogue
        cvt_all_w<D, 224, q3232, intBits<31>, fracBits<32>, isSigned<true>>(out, 19);     // 65 -> 64 storage bits
        cvt_all_w<D, 252, q3132, intBits<32>, fracBits<32>, isSigned<true>>(out, 20);     // 64 -> 65
        cvt_all_w<D, 280, q3132, intBits<40>, fracBits<40>, isSigned<true>>(out, 21);     // exact left shift into a wide type
        cvt_all_w<D, 308, q6400, intBits<63>, fracBits<0>, isSigned<true>>(out, 22);
        cvt_all_w<D, 336, q6300, intBits<64>, fracBits<0>, isSigned<true>>(out, 23);
        cvt_all_w<D, 364, q2030, intBits<60>, fracBits<59>, isSigned<true>>(out, 24);     // left shift by 29 into 120 bits
        break;
    case 2:   // Qmul / Qadd / Qsub whose full-precision result or target is wide
        CASE(400, op_table_w<0, q1516, q1516, intBits<31>, fracBits<32>>(out, "int31_frac32", 31));
        CASE(401, op_table_w<0, q1516, q1516, FullPrec>(out, "FullPrec", 32));
        CASE(402, op_table_w<0, q1516, q1516>(out, "default", 33));
        CASE(403, op_table_w<0, q3132, q1516, intBits<47>, fracBits<48>>(out, "int47_frac48", 34));
        CASE(404, op_table_w<0, q3132, q3132, intBits<40>, fracBits<30>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>(out, "int40_frac30_CONV_SMGN", 35));
        CASE(405, op_table_w<0, q2030, q2030, FullPrec>(out, "FullPrec", 36));
        CASE(406, op_table_w<0, q4040, q1516, intBits<20>, fracBits<20>, QuMode<RND::ZERO>, OfMode<SAT::ZERO>>(out, "int20_frac20_ZERO_ZERO", 37));
        CASE(407, op_table_w<0, u5050, q1516, intBits<60>, fracBits<40>, QuMode<TRN::SMGN>, OfMode<WRP::TCPL>>(out, "int60_frac40_SMGN_WRP", 38));
        CASE(408, op_table_w<1, q4332, q4332>(out, "default", 41));
        CASE(409, op_table_w<1, q4332, q4332, FullPrec>(out, "FullPrec", 42));
        CASE(410, op_table_w<1, q3132, q3132, q4332>(out, "q4332", 43));
        CASE(411, op_table_w<1, q4040, q2030, intBits<41>, fracBits<35>, QuMode<RND::INF>>(out, "int41_frac35_INF", 44));
        CASE(412, op_table_w<1, q6059, q4040, intBits<50>, fracBits<50>, QuMode<RND::NEG_INF>, OfMode<SAT::ZERO>>(out, "int50_frac50_NEGINF_ZERO", 45));
        CASE(413, op_table_w<2, q4332, q3132>(out, "default", 46));
        CASE(414, op_table_w<2, q4040, q4040, intBits<30>, fracBits<45>, OfMode<WRP::TCPL>>(out, "int30_frac45_WRP", 47));
        CASE(415, op_table_w<2, u5050, u5050, isSigned<true>, intBits<51>>(out, "signed_int51", 48));
        CASE(416, op_table_w<1, q6300, q6300, FullPrec>(out, "FullPrec", 49));
        CASE(417, op_table_w<2, q6400, q6300>(out, "default", 50));
        CASE(418, op_table_w<0, q1516, q1516, intBits<31>, fracBits<30>, QuMode<RND::POS_INF>>(out, "int31_frac30_POSINF", 51));
        CASE(419, op_table_w<1, q3132, q3132, intBits<35>, fracBits<30>, QuMode<RND::POS_INF>>(out, "int35_frac30_POSINF", 52));
        break;
    case 3:   // Qreduce with wide level types
        CASE(430, reduce_table_w<q3132, 16, q4332>(out, "q3132_len16_q4332"));
        CASE(431, reduce_table_w<q3132, 100, q4332>(out, "q3132_len100_q4332"));
        CASE(432, reduce_table_w<q3132, 512, q4332>(out, "q3132_len512_q4332"));
        CASE(433, reduce_table_w<q3132, 7, q3232, q4332>(out, "q3132_len7_q3232_q4332"));
        CASE(434, reduce_table_w<q3132, 64, Qu<intBits<33>, fracBits<30>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, q4332>(out, "q3132_len64_conv_q4332"));
        CASE(435, reduce_table_w<q2030, 33, q4040>(out, "q2030_len33_q4040"));
        CASE(436, reduce_table_w<q3132, 1000, Qu<intBits<45>, fracBits<32>>>(out, "q3132_len1000_q4532"));
        break;
    case 4: {   // Qgemul on Q15.16 words with an exact product and exact sums (the linear class beyond 64 bits)
        using mul = TypeList<intBits<31>, fracBits<32>>;
        using add = TypeList<q4332>;
        CASE(440, run_case<q1516, q1516, q4332, mul, add, false, 8, 6, 64>("w_q1516_L_8x6x64_wideC", syn(0), out));
        CASE(441, run_case<q1516, q1516, q1516, mul, add, true, 8, 6, 64>("w_q1516_L_tn_8x6x64_q1516C", syn(0), out));
        CASE(442, run_case<q1516, q1516, Qu<intBits<20>, fracBits<20>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>, mul, add, false, 5, 7, 100>("w_q1516_L_5x7x100_convC", syn(0), out));
        CASE(443, run_case<q1516, q1516, q4332, mul, add, false, 4, 4, 4096>("w_q1516_L_4x4x4096_wideC", syn(0), out));
        CASE(444, run_case<q1516, q1516, q1516, mul, add, false, 4, 4, 4096>("w_q1516_L_4x4x4096_q1516C", syn(0), out));
        CASE(445, run_case<q1516, q1516, Qu<intBits<40>, fracBits<40>>, mul, add, false, 6, 5, 512>("w_q1516_L_6x5x512_q4040C", syn(1), out));
        CASE(446, run_case<q1516, q1516, Qu<intBits<15>, fracBits<16>, QuMode<RND::POS_INF>>, mul, add, false, 6, 5, 512>("w_q1516_L_6x5x512_q1516posinfC", syn(1), out));
        break;
    }
    case 5: {   // tree class with wide intermediates: quantising wide products / levels
        using mulr = TypeList<intBits<31>, fracBits<32>>;
        using lv1 = Qu<intBits<35>, fracBits<30>, QuMode<RND::POS_INF>, OfMode<SAT::TCPL>>;     // 66 storage bits, rounds by 2
        using lv2 = Qu<intBits<33>, fracBits<34>, QuMode<RND::CONV>, OfMode<SAT::SMGN>>;        // saturates: 33 < 31 + levels
        using lv3 = Qu<intBits<36>, fracBits<32>, OfMode<WRP::TCPL>>;
        using lv4 = Qu<intBits<33>, fracBits<32>>;                                              // default modes, saturates
        CASE(450, run_case<q1516, q1516, q4332, mulr, TypeList<lv1>, false, 6, 5, 64>("w_q1516_T_6x5x64_lv1", syn(0), out));
        CASE(451, run_case<q1516, q1516, q4332, mulr, TypeList<lv2, lv1>, false, 6, 5, 100>("w_q1516_T_6x5x100_lv2_lv1", syn(0), out));
        CASE(452, run_case<q1516, q1516, q1516, mulr, TypeList<lv3>, true, 4, 6, 512>("w_q1516_T_tn_4x6x512_lv3", syn(0), out));
        CASE(453, run_case<q1516, q1516, Qu<intBits<36>, fracBits<32>>, mulr, TypeList<lv3>, false, 3, 3, 1024>("w_q1516_T_3x3x1024_lv3", syn(0), out));
        CASE(454, run_case<q3132, q1516, Qu<intBits<50>, fracBits<40>>, TypeList<intBits<47>, fracBits<40>, QuMode<RND::ZERO>>, TypeList<Qu<intBits<55>, fracBits<40>>>, false, 4, 4, 64>("w_q3132_q1516_T_4x4x64", syn(0), out));
        CASE(455, run_case<q1516, q1516, q4332, TypeList<FullPrec>, TypeList<q4332>, false, 5, 4, 37>("w_q1516_T_fullprec_5x4x37", syn(0), out));
        CASE(456, run_case<q1516, q1516, q1516, mulr, TypeList<lv4>, false, 6, 5, 64>("w_q1516_T_6x5x64_lv4", syn(0), out));
        CASE(457, run_case<q1516, q1516, lv4, mulr, TypeList<lv4>, false, 4, 4, 2048>("w_q1516_T_4x4x2048_lv4", syn(0), out));
        break;
    }
    default:
        return 2;
    }
    return 0;
}

int main(int argc, char** argv)
{
    return run_part<0>(argc > 1 ? std::atoi(argv[1]) : 0, stdout);
}
