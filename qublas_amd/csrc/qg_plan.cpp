// qg_plan.cpp — descriptor analysis (host only).  See qg_plan.h.
#include "qg_plan.h"

#include <stdio.h>
#include <string.h>
#include <vector>

typedef __int128 I128;

namespace {

struct Rng {
    I128 lo, hi;
};

int bits_of(I128 v)
{
    // signed width of v
    unsigned __int128 u = v < 0 ? (unsigned __int128)(~v) : (unsigned __int128)v;
    int n = 0;
    while (u) { ++n; u >>= 1; }
    return n + 1;
}

struct Ctx {
    QAnalysis* out;
    bool exact = true;      // every step so far was the identity
    int max_bits = 1;
    int max_bits_np = 1;    // the same, not counting full-precision products before their rounding
    int max_fmt_bits = 1;   // widest storage (1 + W) of any format on the path
    bool raw_product = false;
    bool band = false;      // a multi-word value in [2^63, 2^64) / [-2^64, -2^63) can reach a one-word target (through())
    bool final_step = false; // the conversion into C is being analysed (nothing follows it)
    bool raw_c = false;     // C's WRP::TCPL_SAT lets the root through unclamped: host-word containers, general kernels
    void fail(int st, const char* why)
    {
        if (out->status == QG_OK) {
            out->status = st;
            snprintf(out->reason, sizeof out->reason, "%s", why);
        }
    }
    void note(Rng r)
    {
        int b = bits_of(r.lo), c = bits_of(r.hi);
        if (c > b) b = c;
        if (b > max_bits) max_bits = b;
        if (!raw_product && b > max_bits_np) max_bits_np = b;
    }
};

bool same(qfmt a, qfmt b) { return a.I == b.I && a.F == b.F && a.S == b.S && a.Q == b.Q && a.O == b.O; }

bool fmt_ok(Ctx& c, qfmt f)
{
    int W = (int)f.I + (int)f.F;
    if (W < 0) { c.fail(QG_EINVAL, "intBits + fracBits < 0"); return false; }
    if (f.Q > QG_TRN_SMGN) { c.fail(QG_EINVAL, "unknown QuMode code"); return false; }
    if (f.O > QG_WRP_TCPL_SAT) { c.fail(QG_EINVAL, "unknown OfMode code"); return false; }   // (WRP::TCPL_SAT: see through())
    if (W > 119) { c.fail(QG_EUNSUPPORTED, "format wider than 120 storage bits"); return false; }
    // ArbiInt<65>::maximum() is -1 (oneBits = 65 % 64 - 1 = 0 selects ~0 for the top word, QuBLAS.h:594-603): every saturating
    // conversion INTO a format of exactly 65 storage bits is an artefact in the reference (tests/golden/ref_wide_1: Qu<32,32>, Qu<64,0>)
    if (W == 64) { c.fail(QG_EUNSUPPORTED, "a format of exactly 65 storage bits: ArbiInt<65>::maximum() artefact of the reference"); return false; }
    return true;
}

Rng fmt_range(qfmt f)
{
    int W = (int)f.I + (int)f.F;
    Rng r;
    r.hi = ((I128)1 << W) - 1;
    r.lo = f.S ? -((I128)1 << W) : 0;
    return r;
}

thread_local int g_fmt_bits_seen = 0; // widest target format make_step has seen since qg_analyze reset it

QStep make_step(int fromF, qfmt to, bool identity)
{
    if (1 + (int)to.I + (int)to.F > g_fmt_bits_seen) g_fmt_bits_seen = 1 + (int)to.I + (int)to.F;
    QStep s;
    memset(&s, 0, sizeof s);
    s.d = fromF - (int)to.F;
    s.Q = to.Q;
    s.O = to.O;
    s.W = (int)to.I + (int)to.F;
    s.S = to.S;
    s.identity = identity ? 1 : 0;
    if (s.W <= 62) {   // (wide steps derive their bounds from W and S on the device: qg_overflow_w)
        s.hi = (int64_t)(((I128)1 << s.W) - 1);
        s.lo = to.S ? -(int64_t)((I128)1 << s.W) : 0;
    }
    return s;
}

// width of the ArbiInt that fracConvert returns for an n-bit input: shifts change the width by the shift (QuBLAS.h:1485-1701),
// the RND modes end in `Xh + ArbiInt<1>` — one bit more (:2032) —, TRN::TCPL / TRN::SMGN keep n - d (:2166, :2175-2178)
int round_width(int n, int d, int Q)
{
    if (d <= 0) return n - d;
    const int w = n - d < 1 ? 1 : n - d;
    return (Q == QG_TRN_TCPL || Q == QG_TRN_SMGN) ? w : w + 1;
}

// propagate an exact interval (raw values at frac fromF) through round + overflow into `to`.  nin: width of the ArbiInt TYPE
// the value has in the reference before this step (0: not tracked — element-wise chains, which stay within 62 bits); st: the
// step record, whose refcmp this sets.
Rng through(Ctx& c, Rng in, int fromF, qfmt to, bool identity, int nin = 0, QStep* st = nullptr)
{
    c.note(in);
    c.raw_product = false;
    if (identity) return in;
    // Converting INTO an unsigned WRP::TCPL format of exactly 32 value bits: the reference's mask is ArbiInt<32>::allOnes(),
    // whose data is -1 (QuBLAS.h:361-377), so `val & mask` (:2328-2331) masks nothing and the value is stored unwrapped in
    // the 33-bit storage (-7 -> -7, 2^33 + 5 -> 2^33 + 5; pinned by tests/golden ref_scalar cvt tables).  An artefact of
    // the same family as the d = 32 RND case below: rejected, not imitated.  (Signed, 32 storage bits: the reference's
    // int32 storage wraps by itself and agrees with the arithmetic definition.)  Checked BELOW, only where a value can actually
    // leave the range: an in-range value is untouched by the reference's `val & allOnes` too (uint32-style formats run).
    int d = fromF - (int)to.F;
    Rng r = in;
    if (d <= 0) {
        if (-d > 100 || bits_of(in.lo) - d > 126 || bits_of(in.hi) - d > 126) { c.fail(QG_EUNSUPPORTED, "left shift beyond 127 bits"); return in; }
        r.lo = in.lo * ((I128)1 << -d);
        r.hi = in.hi * ((I128)1 << -d);
        c.note(r);
    } else {
        if (d > 119) { c.fail(QG_EUNSUPPORTED, "rounding shift beyond 119 bits"); return in; }
        if ((d == 32 || d == 64) && to.Q <= QG_RND_CONV)
            c.fail(QG_EUNSUPPORTED, "RND over a 32/64-bit shift: reference result is an ArbiInt<32>::allOnes artefact");
        // RND::CONV of a multi-word value: the reference's floor / ceil construction (QuBLAS.h:2137-2156) returns artefacts (every
        // RND::CONV table of tests/golden/ref_wide_* with a source wider than 64 bits differs from round-half-to-even)
        if (to.Q == QG_RND_CONV && nin > 64)
            c.fail(QG_EUNSUPPORTED, "RND::CONV of a value wider than 64 bits: multi-word artefact of the reference");
        c.exact = false;
        r.lo = in.lo >> d;
        r.hi = (in.hi >> d) + 1;
    }
    const int nr = nin ? round_width(nin, d, to.Q) : 0;   // type width of what reaches the overflow handling
    const int mbits = 1 + (int)to.I + (int)to.F;
    // signed WRP::TCPL from one multi-word type into another of a different width: operator| of two wide integers of different
    // sizes is ill-formed (QuBLAS.h:1946-1950) — the reference does not compile such a conversion (oracle/ref_cases_wide_probe.log)
    if (to.O == QG_WRP_TCPL && to.S && mbits > 64 && nr > 64 && nr != mbits)
        c.fail(QG_EUNSUPPORTED, "signed WRP::TCPL between multi-word types of different widths: the reference does not compile it");
    Rng R = fmt_range(to);
    Rng Re = R;
    if (to.O == QG_SAT_SMGN) Re.lo = to.S ? -R.hi : 0;
    if (to.O == QG_WRP_TCPL_SAT) {
        // WRP::TCPL_SAT<N> is a stub in the reference: intConvert returns its input (QuBLAS.h:2336-2344) and the assignment into the
        // target's storage narrows it to the storage WORD (int32_t / int64_t, never masked to the format's bits: :353, :431-441;
        // tests/golden/ref_scalar_9, ref_gemm_real_8).  In range that is the identity; out of range the value leaves its format,
        // and what the reference's next operation would make of it (int32_t arithmetic on a word that holds more bits than its
        // type says) is not something to build on: accepted only where nothing follows — the conversion into C.
        if (r.lo >= R.lo && r.hi <= R.hi) { if (st) st->O = QG_SAT_TCPL; return r; }   // (any in-range overflow mode: a clamp that never fires)
        if (!c.final_step) { c.fail(QG_EUNSUPPORTED, "WRP::TCPL_SAT (a stub in the reference) whose value can leave its format before the last conversion"); return r; }
        c.exact = false;
        c.raw_c = true;
        const I128 full = mbits <= 32 ? ((I128)1 << 31) : mbits <= 64 ? ((I128)1 << 63) : ((I128)1 << 126);
        if (r.lo < -full) r.lo = -full;
        if (r.hi > full - 1) r.hi = full - 1;
        return r;
    }
    if (nr > 64 && mbits <= 64 && to.O <= QG_SAT_SMGN) {
        // a multi-word value compared with one-word bounds: the reference reads the low word as a signed number (operator<=>,
        // QuBLAS.h:1781-1793) and narrows by keeping the low word (:436-441).  Identical to the arithmetic definition unless the
        // value lies in [2^63, 2^64) or [-2^64, -2^63); the kernels reproduce it (QStep::refcmp, qg_overflow_w).
        if (st) st->refcmp = 1;
        const I128 b63 = (I128)1 << 63, b64 = (I128)1 << 64;
        if ((r.hi >= b63 && r.lo < b64) || (r.lo < -b63 && r.hi >= -b64)) {
            // ... reachable: such a value comes out as its low word, whatever the format says (int32_t / int64_t storage)
            c.exact = false;
            c.band = true;
            const I128 full = mbits <= 32 ? ((I128)1 << 31) : ((I128)1 << 63);
            Rng q = r;
            if (q.lo < Re.lo) q.lo = Re.lo;
            if (q.hi > Re.hi) q.hi = Re.hi;
            if (to.O == QG_SAT_ZERO) { if (q.lo > 0) q.lo = 0; if (q.hi < 0) q.hi = 0; }
            if (q.lo > -full) q.lo = -full;
            if (q.hi < full - 1) q.hi = full - 1;
            return q;
        }
    }
    if (r.lo >= Re.lo && r.hi <= Re.hi) return r; // overflow handling is the identity
    c.exact = false;
    if (to.O == QG_WRP_TCPL && !to.S && (int)to.I + (int)to.F == 32)
        c.fail(QG_EUNSUPPORTED, "unsigned WRP::TCPL into exactly 32 bits: reference result is an ArbiInt<32>::allOnes artefact");
    switch (to.O) {
    case QG_SAT_TCPL:
    case QG_SAT_SMGN:
        if (r.lo < Re.lo) r.lo = Re.lo;
        if (r.hi > Re.hi) r.hi = Re.hi;
        if (r.lo > Re.hi) r.lo = Re.hi;
        if (r.hi < Re.lo) r.hi = Re.lo;
        return r;
    case QG_SAT_ZERO: {
        Rng q = r;
        if (q.lo < Re.lo) q.lo = Re.lo;
        if (q.hi > Re.hi) q.hi = Re.hi;
        if (q.lo > 0) q.lo = 0;
        if (q.hi < 0) q.hi = 0;
        if (q.lo > q.hi) { q.lo = 0; q.hi = 0; }
        return q;
    }
    default:
        return R; // wrap: anything representable
    }
}

Rng mul_rng(Rng a, Rng b)
{
    I128 p[4] = {a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi};
    Rng r = {p[0], p[0]};
    for (int i = 1; i < 4; ++i) {
        if (p[i] < r.lo) r.lo = p[i];
        if (p[i] > r.hi) r.hi = p[i];
    }
    return r;
}

struct Val {
    Rng r;
    qfmt f;
};
int sbits(qfmt f) { return 1 + (int)f.I + (int)f.F; }   // storage bits = width of the value's ArbiInt type (QuBLAS.h:2384-2385)
int rng_bits(Rng r) { const int a = bits_of(r.lo), b = bits_of(r.hi); return a > b ? a : b; }

Val do_mul(Ctx& c, Val a, Val b, qfmt res, QNode* node)
{
    node->sa = node->sb = 0;
    node->q = make_step((int)a.f.F + (int)b.f.F, res, false);
    Val v;
    v.f = res;
    v.r = fmt_range(res);
    if (rng_bits(a.r) + rng_bits(b.r) > 127) { c.fail(QG_EUNSUPPORTED, "a product needs more than 127 bits"); return v; }
    c.raw_product = true; // the unrounded product is noted for max_bits only
    // operator*: an N-bit by an M-bit integer gives N + M bits (QuBLAS.h:1186-1207)
    v.r = through(c, mul_rng(a.r, b.r), (int)a.f.F + (int)b.f.F, res, false, sbits(a.f) + sbits(b.f), &node->q);
    return v;
}

Val do_addsub(Ctx& c, Val a, Val b, qfmt res, bool sub, QNode* node)
{
    int fm = a.f.F > b.f.F ? a.f.F : b.f.F;
    node->sa = fm - a.f.F;
    node->sb = fm - b.f.F;
    node->q = make_step(fm, res, false);
    Val v;
    v.f = res;
    v.r = fmt_range(res);
    if (rng_bits(a.r) + node->sa > 125 || rng_bits(b.r) + node->sb > 125) { c.fail(QG_EUNSUPPORTED, "an aligned operand needs more than 126 bits"); return v; }
    Rng x = {a.r.lo * ((I128)1 << node->sa), a.r.hi * ((I128)1 << node->sa)};
    Rng y = {b.r.lo * ((I128)1 << node->sb), b.r.hi * ((I128)1 << node->sb)};
    c.note(x);
    c.note(y);
    // the aligned operands are N + shift bits wide, their sum / difference one bit more than the wider one (QuBLAS.h:914-1010)
    const int na = sbits(a.f) + node->sa, nb = sbits(b.f) + node->sb;
    // operator-(multi-word, one-word) loses the borrow / sign extension of the one-word operand (tests/golden/ref_wide_2:
    // Qsub(Qu<43,32>, Qu<31,32>) is off by multiples of 2^63): an artefact, rejected
    if (sub && na > 64 && nb <= 64) c.fail(QG_EUNSUPPORTED, "Qsub of a one-word value from a multi-word one: artefact of the reference's operator-");
    Rng s = sub ? Rng{x.lo - y.hi, x.hi - y.lo} : Rng{x.lo + y.lo, x.hi + y.hi};
    v.r = through(c, s, fm, res, false, (na > nb ? na : nb) + 1, &node->q);
    return v;
}

Val do_cvt(Ctx& c, Val a, qfmt to, QStep* st)
{
    bool id = same(a.f, to);
    *st = make_step((int)a.f.F, to, id);
    Val v;
    v.f = to;
    v.r = through(c, a.r, (int)a.f.F, to, id, sbits(a.f), st);
    return v;
}

} // namespace

int qg_analyze_ep(qfmt cfmt, const qgemul_epilogue* ep, QEpTable* out, int* max_bits, char* reason, size_t reason_len)
{
    memset(out, 0, sizeof *out);
    g_fmt_bits_seen = 0;
    QAnalysis* an = new QAnalysis;
    memset(an, 0, sizeof *an);
    an->status = QG_OK;
    Ctx c;
    c.out = an;
    auto pow2_bytes = [](qfmt f) {
        int b = (1 + (int)f.I + (int)f.F + 7) / 8, r = 1;
        while (r < b) r *= 2;
        return r;
    };
    do {
        if (!ep || ep->n_stages > QG_MAX_EW) { c.fail(QG_EINVAL, "null epilogue or too many stages"); break; }
        if (!fmt_ok(c, cfmt) || !fmt_ok(c, ep->d)) break;
        Val x;
        x.f = cfmt;
        x.r = fmt_range(cfmt);
        bool ok = true;
        for (uint32_t k = 0; k < ep->n_stages && ok; ++k) {
            const qgemul_ew_stage& s = ep->stage[k];
            if (s.op < QG_EW_ADD || s.op > QG_EW_PASS) { c.fail(QG_EINVAL, "unknown element-wise op"); ok = false; break; }
            if (s.op == QG_EW_PASS) {
                // the part is carried over in its own format (the imaginary part under a real operand, QuBLAS.h:3654/3670/3701);
                // only the assignment to the stage's tensor touches it
                QEpStage& t = out->st[k];
                memset(&t, 0, sizeof t);
                t.op = QG_EW_PASS;
                t.scalar = 1;                       // no operand is read
                t.ebytes = 1;
                t.node.q.identity = 1;
                t.cvt.identity = 1;
                if (k + 1 < ep->n_stages) {
                    if (!fmt_ok(c, s.t)) { ok = false; break; }
                    x = do_cvt(c, x, s.t, &t.cvt);
                }
                continue;
            }
            if (!fmt_ok(c, s.e) || !fmt_ok(c, s.r)) { ok = false; break; }
            Val e;
            e.f = s.e;
            e.r = fmt_range(s.e);
            QEpStage& t = out->st[k];
            t.op = s.op;
            t.x_first = s.x_first ? 1 : 0;
            t.scalar = s.e_scalar ? 1 : 0;
            t.ebytes = pow2_bytes(s.e);
            const Val& first = t.x_first ? x : e;
            const Val& second = t.x_first ? e : x;
            x = s.op == QG_EW_MUL ? do_mul(c, first, second, s.r, &t.node) : do_addsub(c, first, second, s.r, s.op == QG_EW_SUB, &t.node);
            memset(&t.cvt, 0, sizeof t.cvt);
            t.cvt.identity = 1;
            if (k + 1 < ep->n_stages) {
                if (!fmt_ok(c, s.t)) { ok = false; break; }
                x = do_cvt(c, x, s.t, &t.cvt);
            }
        }
        if (!ok) break;
        do_cvt(c, x, ep->d, &out->to_d);
        out->n = (int)ep->n_stages;
        out->dbytes = pow2_bytes(ep->d);
        out->max_bits = c.max_bits;
        {
            // 32-bit arithmetic: every intermediate (raw products and aligned operands included) within 32 bits, every
            // format within 32 storage bits, every rounding shift below 31
            bool ok32 = c.max_bits <= 32 && g_fmt_bits_seen <= 32 && 1 + (int)cfmt.I + (int)cfmt.F <= 32;
            auto shift_ok = [](const QStep& q) { return q.identity || q.d <= 30; };
            for (int k = 0; k < out->n; ++k) ok32 = ok32 && shift_ok(out->st[k].node.q) && shift_ok(out->st[k].cvt) && out->st[k].ebytes <= 4;
            out->bits32 = (ok32 && shift_ok(out->to_d)) ? 1 : 0;
        }
        if (c.max_bits > 62) c.fail(QG_EUNSUPPORTED, "epilogue intermediate wider than 62 bits");
    } while (0);
    const int st = an->status;
    if (reason && reason_len) snprintf(reason, reason_len, "%s", an->reason);
    if (max_bits) *max_bits = c.max_bits;
    delete an;
    return st;
}

QHostElem qg_host_elem(const qfmt f[2], int is_complex)
{
    QHostElem L;
    // int32_t / int64_t (ArbiInt<N <= 64>, QuBLAS.h:353) or two little-endian uint64_t words (ArbiInt<65..128>, :572-573; 8-byte aligned)
    auto sb = [](qfmt q) { const int b = 1 + (int)q.I + (int)q.F; return b <= 32 ? 4 : b <= 64 ? 8 : 16; };
    L.sb[0] = sb(f[0]);
    L.off[0] = 0;
    if (!is_complex) {
        L.sb[1] = 0;
        L.off[1] = 0;
        L.size = L.sb[0];
        return L;
    }
    L.sb[1] = sb(f[1]);
    const int a0 = L.sb[0] > 8 ? 8 : L.sb[0], a1 = L.sb[1] > 8 ? 8 : L.sb[1];
    const int al = a0 > a1 ? a0 : a1;
    L.off[1] = (L.sb[0] + a1 - 1) / a1 * a1;
    L.size = (L.off[1] + L.sb[1] + al - 1) / al * al;
    return L;
}

int qg_limbs_for(qfmt f)
{
    // balanced base-256 digits d in [-128,127]; the remainder after peeling n-1 digits is
    // floor((x + 128) / 256) applied n-1 times, monotone in x, so the extremes decide
    Rng r = fmt_range(f);
    for (int n = 1; n <= 8; ++n) {
        I128 lo = r.lo, hi = r.hi;
        for (int i = 1; i < n; ++i) {
            lo = (lo + 128) >> 8;
            hi = (hi + 128) >> 8;
        }
        if (lo >= -128 && hi <= 127) return n;
    }
    return 9;
}

int qg_limbs_centred(qfmt f, int64_t* centre)
{
    const Rng r = fmt_range(f);
    *centre = 0;
    I128 S = 0;
    for (int n = 1; n <= 7; ++n) {
        S = S * 256 + 1;                                  // (256^n - 1) / 255
        if (r.hi - r.lo <= 255 * S) {
            const I128 c = r.lo + 128 * S;                // lo - c = -128 S, hi - c <= 127 S
            if (c < -((I128)1 << 62) || c > ((I128)1 << 62)) return 9;
            *centre = (int64_t)c;
            return n;
        }
    }
    return 9;
}

void qg_analyze(const qgemul_desc* d, QAnalysis* out)
{
    memset(out, 0, sizeof *out);
    out->status = QG_OK;
    g_fmt_bits_seen = 0;
    Ctx c;
    c.out = out;
    if (!d || d->abi != QGEMUL_ABI_VERSION) { c.fail(QG_EINVAL, "null descriptor or ABI mismatch"); return; }
    // QG_DESC_REFERENCE_ARTEFACTS: C of an unsigned WRP::TCPL format with exactly 32 value bits is stored unwrapped by the reference
    // (include/qgemul.h) — which is what its WRP::TCPL_SAT stub does: analysed, and executed, as that
    qgemul_desc with_artefacts;
    if (d->flags & QG_DESC_REFERENCE_ARTEFACTS) {
        with_artefacts = *d;
        for (int p = 0; p < (d->is_complex ? 2 : 1); ++p) {
            qfmt& f = with_artefacts.c[p];
            if (f.O == QG_WRP_TCPL && !f.S && (int)f.I + (int)f.F == 32) f.O = QG_WRP_TCPL_SAT;
        }
        d = &with_artefacts;
    }
    if (d->M < 0 || d->N < 0 || d->K < 1) { c.fail(QG_EINVAL, "bad M/N/K"); return; }
    if (d->M > (1ll << 31) || d->N > (1ll << 31) || d->K > (1ll << 31)) { c.fail(QG_EUNSUPPORTED, "dimension beyond 2^31"); return; }
    {
        int64_t len = d->K;
        uint32_t n = 0;
        while (len > 1) { len = (len + 1) / 2; ++n; }
        if (n != d->n_levels || n > QG_MAX_LEVELS) { c.fail(QG_EINVAL, "n_levels != ceil(log2 K)"); return; }
    }
    const int cx = d->is_complex ? 1 : 0;
    const int parts = cx ? 2 : 1;
    if (cx && d->cmul != QG_CMUL_BASIC && d->cmul != QG_CMUL_TF) { c.fail(QG_EINVAL, "complex descriptor without cmul"); return; }
    if (!cx && d->cmul != QG_CMUL_NONE) { c.fail(QG_EINVAL, "real descriptor with cmul"); return; }

    QTreeTable& T = out->tree;
    T.is_complex = cx;
    T.cmul = d->cmul;
    T.n_levels = (int)d->n_levels;
    T.n_levels_k = d->n_levels < 5 ? 5 : (int)d->n_levels;
    for (int p = 0; p < 2; ++p)
        for (int l = (int)d->n_levels; l < T.n_levels_k && l < QG_MAX_LEVELS; ++l) {   // identity levels behind a short tree
            memset(&T.level_add[p][l], 0, sizeof T.level_add[p][l]);
            T.level_add[p][l].q.identity = 1;
            memset(&T.level_cvt[p][l], 0, sizeof T.level_cvt[p][l]);
            T.level_cvt[p][l].identity = 1;
            memset(&T.leftover[p][l], 0, sizeof T.leftover[p][l]);
            T.leftover[p][l].identity = 1;
        }
    T.parts = parts;

    for (int p = 0; p < parts; ++p) {
        if (!fmt_ok(c, d->a[p]) || !fmt_ok(c, d->b[p]) || !fmt_ok(c, d->c[p])) return;
        // operand ELEMENTS are one-word values (int32_t / int64_t host elements); products, sums, levels and C may be multi-word
        if (sbits(d->a[p]) > 64 || sbits(d->b[p]) > 64) { c.fail(QG_EUNSUPPORTED, "operand element wider than 64 storage bits"); return; }
    }
    const int nslots = !cx ? 1 : (d->cmul == QG_CMUL_TF ? 8 : 6);
    for (int i = 0; i < nslots; ++i)
        if (!fmt_ok(c, d->mul[i])) return;
    for (int p = 0; p < parts; ++p)
        for (uint32_t l = 0; l < d->n_levels; ++l)
            if (!fmt_ok(c, d->level_add[p][l]) || !fmt_ok(c, d->level[p][l])) return;

    // ---- the product ----
    Val prod[2];
    const qfmt* m = d->mul;
    if (!cx) {
        Val a = {fmt_range(d->a[0]), d->a[0]}, b = {fmt_range(d->b[0]), d->b[0]};
        prod[0] = do_mul(c, a, b, m[QG_MUL_REAL], &T.mul[0]);
        prod[1] = prod[0];
        // the exact dot product, for the linear class
        Rng pr = mul_rng(a.r, b.r);
        Rng dot = {pr.lo * d->K, pr.hi * d->K};
        out->dot_bits = bits_of(dot.lo) > bits_of(dot.hi) ? bits_of(dot.lo) : bits_of(dot.hi);
    } else {
        Val a = {fmt_range(d->a[0]), d->a[0]}, b = {fmt_range(d->a[1]), d->a[1]};
        Val cc = {fmt_range(d->b[0]), d->b[0]}, dd = {fmt_range(d->b[1]), d->b[1]};
        if (d->cmul == QG_CMUL_TF) {
            Val ab = do_addsub(c, a, b, m[QG_T_AB], false, &T.mul[QG_T_AB]);
            Val cd = do_addsub(c, cc, dd, m[QG_T_CD], false, &T.mul[QG_T_CD]);
            Val ba = do_addsub(c, b, a, m[QG_T_BA], true, &T.mul[QG_T_BA]);
            Val A = do_mul(c, ab, cc, m[QG_T_A], &T.mul[QG_T_A]);
            Val B = do_mul(c, cd, b, m[QG_T_B], &T.mul[QG_T_B]);
            Val C = do_mul(c, ba, dd, m[QG_T_C], &T.mul[QG_T_C]);
            prod[0] = do_addsub(c, A, B, m[QG_T_RE], true, &T.mul[QG_T_RE]);
            prod[1] = do_addsub(c, B, C, m[QG_T_IM], true, &T.mul[QG_T_IM]);
        } else {
            Val ac = do_mul(c, a, cc, m[QG_B_AC], &T.mul[QG_B_AC]);
            Val bd = do_mul(c, b, dd, m[QG_B_BD], &T.mul[QG_B_BD]);
            Val ad = do_mul(c, a, dd, m[QG_B_AD], &T.mul[QG_B_AD]);
            Val bc = do_mul(c, b, cc, m[QG_B_BC], &T.mul[QG_B_BC]);
            prod[0] = do_addsub(c, ac, bd, m[QG_B_RE], true, &T.mul[QG_B_RE]);
            prod[1] = do_addsub(c, ad, bc, m[QG_B_IM], false, &T.mul[QG_B_IM]);
        }
    }

    // ---- the tree ----
    for (int p = 0; p < parts; ++p) {
        Val cur = prod[p];
        int64_t len = d->K;
        for (uint32_t l = 0; l < d->n_levels; ++l) {
            Val s = do_addsub(c, cur, cur, d->level_add[p][l], false, &T.level_add[p][l]);
            Val st = do_cvt(c, s, d->level[p][l], &T.level_cvt[p][l]);
            // odd leftover: the converting copy is on the path only when this level has odd length
            bool was_exact = c.exact;
            Val lf = do_cvt(c, cur, d->level[p][l], &T.leftover[p][l]);
            if (l == 0 && (d->flags & QG_DESC_LEFTOVER0_COPY)) {
                // the level-0 buffer has the leaf's own type in the reference: a same-type copy (QuBLAS.h:4977-4980, :2401-2404)
                T.leftover[p][l].identity = 1;
                lf.r = cur.r;
                if (len & 1) c.exact = was_exact;
            }
            if (!(len & 1)) c.exact = was_exact;
            else {
                if (lf.r.lo < st.r.lo) st.r.lo = lf.r.lo;
                if (lf.r.hi > st.r.hi) st.r.hi = lf.r.hi;
            }
            cur = st;
            len = (len + 1) / 2;
        }
        bool tree_exact = c.exact;
        c.final_step = true;
        do_cvt(c, cur, d->c[p], &T.c_cvt[p]);
        c.final_step = false;
        c.exact = tree_exact; // the epilogue is allowed (and expected) to quantise
        if (!cx) {
            // linear-class epilogue: D = sum a*b at frac Fa+Fb, one round+overflow into C.
            // Equal to cvt_C(root) because root = D << (F_root - Fa - Fb) exactly and the root
            // lies inside the root format's range.
            int fp = (int)d->a[0].F + (int)d->b[0].F;
            out->lin.to_c[0] = make_step(fp, d->c[0], false);
            out->lin.to_c[0].refcmp = T.c_cvt[0].refcmp;   // (the root's type width decides how the reference compares with C's bounds)
            if (d->c[0].O == QG_WRP_TCPL_SAT && !T.c_cvt[0].identity) out->lin.to_c[0].O = T.c_cvt[0].O;   // (in range: a clamp that never fires)
            out->lin.to_c[1] = out->lin.to_c[0];
            if (out->lin.to_c[0].d > 100 || out->lin.to_c[0].d < -100) c.exact = false;
        }
    }
    if (cx) {
        // Complex linear class.  When every sub-operation and tree node is exact, both multipliers reduce to
        //   re = sum_k (a c - b d),  im = sum_k (a d + b c)
        // (TF: A - B = (a+b)c - (c+d)b, B - C = (c+d)b - (b-a)d), each term carrying the power of two of its
        // operands' fracBits.  Evaluate the two parts at frac Fx = max(Fa+Fc, Fb+Fd) and Fy = max(Fa+Fd, Fb+Fc).
        const int Fa = d->a[0].F, Fb = d->a[1].F, Fc = d->b[0].F, Fd = d->b[1].F;
        const int Fx = (Fa + Fc > Fb + Fd) ? Fa + Fc : Fb + Fd;
        const int Fy = (Fa + Fd > Fb + Fc) ? Fa + Fd : Fb + Fc;
        out->lin.sh[0] = Fx - (Fa + Fc);
        out->lin.sh[1] = Fx - (Fb + Fd);
        out->lin.sh[2] = Fy - (Fa + Fd);
        out->lin.sh[3] = Fy - (Fb + Fc);
        out->lin.to_c[0] = make_step(Fx, d->c[0], false);
        out->lin.to_c[1] = make_step(Fy, d->c[1], false);
        for (int p = 0; p < 2; ++p)
            if (out->lin.to_c[p].d > 61 || out->lin.to_c[p].d < -61) c.exact = false;
        // width of the combined value: K terms of |a c| 2^sh + |b d| 2^sh
        Rng ra = fmt_range(d->a[0]), rb = fmt_range(d->a[1]), rc = fmt_range(d->b[0]), rd = fmt_range(d->b[1]);
        I128 m = 0;
        auto mag = [](Rng r) { I128 a = r.lo < 0 ? -r.lo : r.lo; return a > r.hi ? a : r.hi; };
        I128 t1 = mag(ra) * mag(rc) * ((I128)1 << out->lin.sh[0]) + mag(rb) * mag(rd) * ((I128)1 << out->lin.sh[1]);
        I128 t2 = mag(ra) * mag(rd) * ((I128)1 << out->lin.sh[2]) + mag(rb) * mag(rc) * ((I128)1 << out->lin.sh[3]);
        m = (t1 > t2 ? t1 : t2) * d->K;
        out->dot_bits = bits_of(m);
        if (out->dot_bits > 62 || out->lin.sh[0] > 40 || out->lin.sh[1] > 40 || out->lin.sh[2] > 40 || out->lin.sh[3] > 40) c.exact = false;
    }
    if (out->status != QG_OK) return;

    out->max_bits = c.max_bits;
    out->max_bits_np = c.max_bits_np;
    c.max_fmt_bits = g_fmt_bits_seen; // the int32 kernels keep every format's bounds in 32-bit registers
    // 64-bit kernels: every value within 62 bits, so that sums, alignment shifts and rounding addends stay inside int64 — except
    // the unrounded product of two operands, which is formed exactly by one 64-bit multiply and goes straight into its rounding
    // shift: it may use all of int64 (two signed 32-bit words: |a * b| <= 2^62).  32-bit fixed-point words therefore run.
    // Beyond that ("wide" plans): 128-bit values — the reference's multi-word ArbiInt<N > 64> (QuBLAS.h:566-912) — on the
    // general tree kernel's 128-bit instantiation, or, for the linear class, the composite MFMA plan with a 128-bit combine pass.
    if (c.max_bits_np > 120 || c.max_bits > 127) { c.fail(QG_EUNSUPPORTED, "an intermediate needs more than 120 bits"); return; }
    out->wide = (c.max_bits_np > 62 || c.max_bits > 64 || g_fmt_bits_seen > 62) ? 1 : 0;
    out->band = (c.band || c.raw_c) ? 1 : 0;
    out->generic_only = c.raw_c ? 1 : 0;
    if ((out->wide || out->generic_only) && cx) c.exact = false;   // (complex linear class: its own 62-bit combine only)
    if (!out->wide && (out->lin.to_c[0].d > 61 || out->lin.to_c[0].d < -61)) c.exact = false;
    out->linear_ok = c.exact ? 1 : 0;
    out->cls = out->linear_ok ? QG_CLASS_LINEAR : QG_CLASS_TREE;
    // 32-bit tree kernel (qg_tree_fast.hip): real, K a power of two >= 32, every value except the
    // unrounded product fits 31 bits, and the product is either directly 32-bit or splittable at
    // its rounding shift
    out->tree_fast_ok = 0;
    // (any K with 5..16 levels: the packed operands are zero-padded to 2^n_levels leaves — a node whose right child is a
    // zero is the reference's converting copy of an odd leftover, QuBLAS.h:4977-4980)
    if (!cx && !out->wide && !out->generic_only && d->n_levels <= 16 && c.max_bits_np <= 31 && c.max_fmt_bits <= 31) {
        const int bitsA = 1 + (int)d->a[0].I + (int)d->a[0].F, bitsB = 1 + (int)d->b[0].I + (int)d->b[0].F;
        const int sh = T.mul[0].q.d;
        int bh = bitsB - sh;
        if (bh < 1) bh = 1;
        if (bitsA + bitsB <= 31) {
            out->tree_fast_ok = 1;
            out->split_s = 0;
            out->mul24_ok = bitsA <= 24 && bitsB <= 24;
        } else if (sh >= 1 && sh <= 23 && bitsA + bh <= 31 && bitsA + sh <= 31) {
            out->tree_fast_ok = 1;
            out->split_s = sh;
            out->mul24_ok = bitsA <= 24 && bh <= 24;
        }
    }
    // fixed-mode variants of the 32-bit tree kernel: product and all levels in one format, truncating
    // product, SAT::ZERO or SAT::TCPL overflow (the shapes default tags produce)
    out->fast_mode = 0;
    if (out->tree_fast_ok && out->mul24_ok) {
        const qfmt pf = d->mul[0];
        bool one = (pf.Q == QG_TRN_TCPL || T.mul[0].q.d <= 0) && T.mul[0].q.d >= 0 && (pf.O == QG_SAT_ZERO || pf.O == QG_SAT_TCPL);
        for (uint32_t l = 0; l < d->n_levels && one; ++l) one = same(d->level[0][l], pf) && same(d->level_add[0][l], pf);
        const int W = (int)pf.I + (int)pf.F;
        if (one && W + 1 + T.mul[0].q.d <= 30) out->fast_mode = pf.O == QG_SAT_ZERO ? 1 : 2;
    }
    // one-column kernel (qg_gemv.hip): products are formed in 64 bits, everything else in 32
    out->gemv_ok = (!cx && !out->wide && !out->generic_only && d->N == 1 && d->n_levels <= 30 && c.max_bits_np <= 31 && c.max_fmt_bits <= 31) ? 1 : 0;
    // ... with 64-bit tree values when only the ELEMENTS fit 32 storage bits (sums of 32-bit words, wide level types)
    out->gemv_wide_ok = (!cx && !out->wide && !out->generic_only && !out->gemv_ok && d->N == 1 && d->n_levels <= 30 &&
                         1 + (int)d->a[0].I + (int)d->a[0].F <= 32 && 1 + (int)d->b[0].I + (int)d->b[0].F <= 32) ? 1 : 0;
    // fast_mode 3: per-level formats, but every step "add a constant, shift right, clamp" (QFix, qg_plan.h): TRN::TCPL /
    // RND::POS_INF / RND::NEG_INF rounding; SAT::TCPL / SAT::SMGN (one clamp), SAT::ZERO (range test + select) or WRP::TCPL
    // (sign extension / mask) overflow — e.g. default modes with a wider level type in QgemulAddArgs, which used to take the
    // run-time-mode form (4.3x slower at 2048^3)
    int rec_form = 0;   // 3 / 5: the unbiased records are valid (every level clamps / overflow kinds), for the one-column kernel
    if (((out->tree_fast_ok && out->mul24_ok && out->fast_mode == 0) || out->gemv_ok)) {
        const bool for_tree = out->tree_fast_ok && out->mul24_ok && out->fast_mode == 0 && !out->gemv_ok;
        auto okq = [](const QStep& q) {   // (every QuMode: the value-dependent ones as a rounding kind of the unbiased form)
            return q.identity || (q.d >= -22 && q.d <= 29 && (q.O == QG_SAT_TCPL || q.O == QG_SAT_SMGN || q.O == QG_SAT_ZERO || (q.O == QG_WRP_TCPL && q.W >= 1)));
        };
        auto rk_of = [](const QStep& q) {
            if (q.identity || q.d <= 0) return 0;
            return q.Q == QG_RND_ZERO ? 1 : q.Q == QG_RND_INF ? 2 : q.Q == QG_RND_CONV ? 3 : q.Q == QG_TRN_SMGN ? 4 : 0;
        };
        auto fix_of = [](const QStep& q, QFix* f) {
            memset(f, 0, sizeof *f);
            f->ka = 1;
            if (q.identity) { f->lo = INT32_MIN; f->hi = INT32_MAX; return; }
            f->kb = q.O == QG_SAT_ZERO ? 1 : q.O == QG_WRP_TCPL ? (q.S ? 2 : 3) : 0;   // overflow kind (qg_plan.h)
            f->lo = q.O == QG_SAT_SMGN ? (q.S ? -(int32_t)q.hi : 0) : (int32_t)q.lo;
            f->hi = (int32_t)q.hi;
            if (q.d < 0) { f->ls = -q.d; return; }
            f->d = q.d;
            f->t = q.d == 0 ? 0 : q.Q == QG_RND_POS_INF ? (1 << (q.d - 1)) : q.Q == QG_RND_NEG_INF ? (1 << (q.d - 1)) - 1 : 0;
            if (q.d > 0) {   // value-dependent roundings: kind + constant (qg_fix.h)
                const int32_t half = 1 << (q.d - 1);
                switch (q.Q) {
                case QG_RND_ZERO: f->skip = 1; f->ka = half - 1; break;
                case QG_RND_INF: f->skip = 2; f->ka = half; break;
                case QG_RND_CONV: f->skip = 3; f->ka = half - 1; break;
                case QG_TRN_SMGN: f->skip = 4; f->ka = (int32_t)(((int64_t)1 << q.d) - 1); break;
                default: break;
                }
            }
        };
        const QStep& pq = T.mul[0].q;
        // (the split product is rounded inside its low half: its shift is the split, never a left shift)
        // (the one-column kernel forms its products itself: only the levels' records matter there)
        bool ok3 = out->gemv_ok || (okq(pq) && (out->split_s == 0 || (!pq.identity && pq.d == out->split_s)));
        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && ok3; ++l)
            ok3 = okq(T.level_add[0][l].q) && T.level_cvt[0][l].identity && T.level_add[0][l].sa == 0 && T.level_add[0][l].sb == 0;
        if (ok3) {
            if (okq(pq)) fix_of(pq, &T.fmul[0]);
            bool clamps = out->gemv_ok || (T.fmul[0].kb == 0 && rk_of(pq) == 0);
            bool plain_rounding = out->gemv_ok || rk_of(pq) == 0;   // no value-dependent rounding anywhere: the biased form applies
            for (uint32_t l = 0; l < (uint32_t)T.n_levels_k; ++l) {
                fix_of(T.level_add[0][l].q, &T.fadd[0][l]);
                clamps = clamps && T.fadd[0][l].kb == 0 && rk_of(T.level_add[0][l].q) == 0;
                plain_rounding = plain_rounding && rk_of(T.level_add[0][l].q) == 0;
            }
            rec_form = clamps ? 3 : 5;
            if (for_tree) out->fast_mode = rec_form;   // 3: every step clamps (one v_med3 per value, no branch on the overflow kind);
                                                       // 5: the records' overflow kinds on unbiased values (any format the kernel admits)
            if (!clamps && for_tree && plain_rounding) {
                QFix keep_mul = T.fmul[0];
                std::vector<QFix> keep_add(T.fadd[0], T.fadd[0] + (uint32_t)T.n_levels_k);
                // 4: some step tests the range (SAT::ZERO) or wraps.  The running value is then kept BIASED by -lo of its own
                // format, u = v - lo >= 0 (as the one-format SAT::ZERO form does): the range test is ONE unsigned compare
                // against span = hi - lo with the biased zero as the select's other operand, a clamp is med3(u, 0, span) with
                // no register for a bound, a wrap is u & span; the change of bias from level to level, the rounding addend and
                // the new bias (scaled by 2^d: subtracting a multiple of 2^d before a right shift by d is exact) are ONE
                // constant of the node's v_add3.  Record fields (QFix): t = that constant, d, ls, kb = overflow kind
                // (0 clamp, 1 zero, 2 wrap, 4 none), hi = span, lo = the bias B = -lo (also the biased zero).
                struct L { int kind; int64_t B, span; };
                auto lv = [](const QStep& q) {
                    L r{4, 0, 0};
                    if (q.identity) return r;
                    const int64_t lo = q.O == QG_SAT_SMGN ? (q.S ? -q.hi : 0) : q.lo;
                    r.kind = q.O == QG_SAT_ZERO ? 1 : q.O == QG_WRP_TCPL ? 2 : 0;
                    r.B = -lo;
                    r.span = q.hi - lo;
                    return r;
                };
                bool fits = c.max_bits_np <= 30;   // (the raw product is never formed: split, or bitsA + bitsB <= 31)
                auto put = [&](const QStep& q, const L& me, int64_t bias_in2, QFix* f) {   // bias_in2: the bias the inputs carry, summed
                    const QFix u = *f;                                                    // (the unbiased record: t, d, ls)
                    memset(f, 0, sizeof *f);
                    f->ka = 1;
                    f->d = u.d;
                    f->ls = u.ls;
                    f->kb = me.kind;
                    f->hi = (int32_t)me.span;
                    f->lo = (int32_t)me.B;
                    const int64_t cst = u.ls ? -bias_in2 : (int64_t)u.t - bias_in2 + (me.B << u.d);   // (left shift: B is added after it)
                    fits = fits && me.span < (1ll << 30) && (me.B << u.d) < (1ll << 29) && cst > -(1ll << 30) && cst < (1ll << 30) &&
                           (me.B << u.ls) < (1ll << 30);
                    f->t = (int32_t)cst;
                    (void)q;
                };
                L cur = lv(pq);
                put(pq, cur, 0, &T.fmul[0]);
                for (uint32_t l = 0; l < (uint32_t)T.n_levels_k; ++l) {
                    const L me = lv(T.level_add[0][l].q);
                    put(T.level_add[0][l].q, me, 2 * cur.B, &T.fadd[0][l]);
                    cur = me;
                }
                T.fmul[0].ka = (int32_t)cur.B;   // the root's bias, removed once per output
                if (fits) {
                    out->fast_mode = 4;
                } else {   // too wide for the bias arithmetic: the unbiased records stand
                    T.fmul[0] = keep_mul;
                    for (uint32_t l = 0; l < (uint32_t)T.n_levels_k; ++l) T.fadd[0][l] = keep_add[l];
                }
            }
        }
    }
    // fast_mode 6, left-justified values (qg_fix.h): the product and every level clamp into ONE signed SAT::TCPL format of
    // Wt = W + 1 bits, the product rounds by "add a constant, shift right by d" (TRN::TCPL, RND::POS_INF, RND::NEG_INF) and the
    // nodes do not shift.  Values are held as x * 2^s, s = 32 - Wt: the operands are staged with factors 2^ea, 2^eb,
    // ea + eb = s - d, so that a * b * 2^(s-d) + t * 2^(s-d), saturated by the multiply-add itself and its low s bits cleared,
    // is the quantised product; a node is one saturating add.  fast_mode_base keeps the form such a descriptor had before.
    out->fast_mode_base = out->fast_mode;
    if (out->tree_fast_ok && (out->fast_mode == 2 || out->fast_mode == 3)) {
        const QStep& pq = T.mul[0].q;
        const int bitsA = 1 + (int)d->a[0].I + (int)d->a[0].F, bitsB = 1 + (int)d->b[0].I + (int)d->b[0].F;
        // (unsigned: every operand, the product and the levels unsigned — nothing ever goes below 0 — and [0, 2^W - 1] is the uint32
        //  range of x * 2^(32 - W): the unsigned multiply-add / add with the clamp bit, QAnalysis::lj_unsigned)
        const bool uns = !pq.S && !d->a[0].S && !d->b[0].S && pq.lo == 0;
        bool lj = !pq.identity && pq.O == QG_SAT_TCPL && (uns || (pq.S && pq.lo == -pq.hi - 1)) && pq.d >= 0 && ((pq.hi + 1) & pq.hi) == 0 && pq.hi > 0 &&
                  (pq.d == 0 || pq.Q == QG_TRN_TCPL || pq.Q == QG_RND_POS_INF || pq.Q == QG_RND_NEG_INF);
        int Wt = uns ? 0 : 1;
        while (lj && ((int64_t)1 << (Wt - (uns ? 0 : 1))) <= pq.hi) ++Wt;   // hi = 2^(Wt-1) - 1 (signed), 2^Wt - 1 (unsigned)
        const int sj = 32 - Wt;
        lj = lj && sj >= 1 && sj - pq.d >= 0;
        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && lj; ++l) {
            const QStep& q = T.level_add[0][l].q;
            const bool pad = (int)l >= T.n_levels && q.identity;   // (x + 0 behind a short tree)
            lj = T.level_cvt[0][l].identity && T.level_add[0][l].sa == 0 && T.level_add[0][l].sb == 0 &&
                 (pad || (!q.identity && q.O == QG_SAT_TCPL && q.S == pq.S && q.d == 0 && q.lo == pq.lo && q.hi == pq.hi));
        }
        if (lj) {
            int ea = sj - pq.d;
            if (ea > 24 - bitsA) ea = 24 - bitsA;
            if (ea < 0) ea = 0;
            const int eb = sj - pq.d - ea;
            lj = bitsA + ea <= 24 && bitsB + eb <= 24 && eb >= 0;
            if (lj) {
                memset(&T.lj, 0, sizeof T.lj);
                const int64_t t = pq.d == 0 ? 0 : pq.Q == QG_RND_POS_INF ? ((int64_t)1 << (pq.d - 1)) : pq.Q == QG_RND_NEG_INF ? ((int64_t)1 << (pq.d - 1)) - 1 : 0;
                T.lj.t[0] = (int32_t)(t << (sj - pq.d));   // (< 2^(s-1))
                T.lj.s = sj;
                T.lj.e[0] = ea;
                T.lj.e[1] = eb;
                out->fast_mode = 6;
                out->lj_unsigned = uns ? 1 : 0;
                // 7: ... in 16-bit halves, two outputs per register (k_tree_pk16): the format has at most 16 bits and the justified
                // operands fit int16 (QTreeTable::lj16).
                const int s16 = 16 - Wt;
                if (s16 >= 1 && s16 - pq.d >= 0) {
                    int ea16 = s16 - pq.d;
                    if (ea16 > 16 - bitsA) ea16 = 16 - bitsA;
                    if (ea16 < 0) ea16 = 0;
                    const int eb16 = s16 - pq.d - ea16;
                    if (bitsA + ea16 <= 16 && bitsB + eb16 <= 16 && eb16 >= 0) {
                        memset(&T.lj16, 0, sizeof T.lj16);
                        T.lj16.s = s16;
                        T.lj16.e[0] = ea16;
                        T.lj16.e[1] = eb16;
                        T.lj16.t[0] = (int32_t)(t << (s16 - pq.d));
                        out->fast_mode = 7;
                    }
                }
                // 8: the format has at most 16 bits but its product does not fit the halves (bits + shift > 16): MODE 6's 32-bit justified
                // product, whose high half is the value justified in 16 bits, and the tree on packed halves (k_tree_pk16<., true>)
                if (out->fast_mode == 6 && Wt >= 2 && Wt <= 16) out->fast_mode = Wt == 16 ? 9 : 8;   // (9: no bits below the unit in a half)
            }
        }
    }
    // fast_mode 10, 32-BIT WORDS (Q15.16 with default tags, the 32-bit fixed-point format of most code): the product and every level
    // clamp into ONE signed SAT::TCPL format of exactly 32 bits — whose range is the int32 range as it stands, no justification and
    // no bits below the unit — the nodes do not shift, and the product rounds by "add a constant, shift right by d" out of the
    // exact 64-bit product of two elements of at most 32 bits.  Then a node is ONE v_add_i32 ... clamp, where the 64-bit tree
    // kernel spends a 64-bit add and a 64-bit clamp; the product is v_mul_hi / v_mul_lo, the shift and a saturation to the word.
    // Runs on the 32-bit tree kernel's frame (k_tree_fast<., 17>), which has no run-time-mode form for such a format: the plan
    // flag QG_OPT_RUNTIME_MODES sends the descriptor to the 64-bit kernel (qg_api.hip).
    if (!out->tree_fast_ok && !cx && !out->wide && !out->generic_only && d->n_levels <= 16 && T.n_levels_k <= 16) {
        const QStep& pq = T.mul[0].q;
        const int bitsA = 1 + (int)d->a[0].I + (int)d->a[0].F, bitsB = 1 + (int)d->b[0].I + (int)d->b[0].F;
        // ... and JUSTIFIED WORDS: a signed SAT::TCPL format of Wt < 32 bits (Q11.12: 24-bit words) held as x * 2^sj, sj = 32 - Wt, as
        // fast_mode 6 holds it, when the product needs a net RIGHT shift dn = d - sj >= 1 (fast_mode 6 needs a left one, and 24-bit
        // factors): the word is floor((a b + t) / 2^dn) of the same exact 64-bit product with its low sj bits cleared, the nodes
        // are fast_mode 6's (T.lj.e[0] = sj; k_tree_fast<., 19 / 20>).  Such descriptors ran on the 64-bit tree kernel.
        int Wt = 1;
        while (Wt < 33 && pq.S && ((int64_t)1 << (Wt - 1)) <= pq.hi) ++Wt;      // hi = 2^(Wt-1) - 1
        const int sj = 32 - Wt, dn = pq.d - sj;
        bool w32 = !pq.identity && pq.O == QG_SAT_TCPL && pq.S && Wt >= 2 && Wt <= 32 && pq.hi == ((int64_t)1 << (Wt - 1)) - 1 && pq.lo == -pq.hi - 1 &&
                   dn >= 1 && dn <= 31 && pq.d <= 31 &&   // (dn = 0: the kernel's range test of the high half has no form; d <= 31: the rounding addend is an int32)
                   (pq.Q == QG_TRN_TCPL || pq.Q == QG_RND_POS_INF || pq.Q == QG_RND_NEG_INF) && bitsA <= 32 && bitsB <= 32 &&
                   (d->a[0].S || bitsA <= 31) && (d->b[0].S || bitsB <= 31);   // (elements are int32 words in the packed operands)
        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && w32; ++l) {
            const QStep& q = T.level_add[0][l].q;
            const bool pad = (int)l >= T.n_levels && q.identity;
            w32 = T.level_cvt[0][l].identity && T.level_add[0][l].sa == 0 && T.level_add[0][l].sb == 0 &&
                  (pad || (!q.identity && q.O == QG_SAT_TCPL && q.S && q.d == 0 && q.lo == pq.lo && q.hi == pq.hi));
        }
        // the root is converted into C by the kernel's 32-bit step: C's bounds must be words too, and no left shift
        const QStep& cq = T.c_cvt[0];
        w32 = w32 && (cq.identity || (cq.d >= 0 && 1 + (int)d->c[0].I + (int)d->c[0].F <= 32));
        // ... and WRAPPING words: the same frame for a signed WRP::TCPL format of exactly 32 bits (T.lj.e[1] = 1; k_tree_fast<., 21>) — the
        // product's word is the low 32 bits of floor((a b + t) / 2^d), 0 <= d <= 31, a node a plain 32-bit add
        bool wrap32 = !w32 && !pq.identity && pq.O == QG_WRP_TCPL && pq.S && pq.W == 31 && pq.d >= 0 && pq.d <= 31 &&
                      (pq.d == 0 || pq.Q == QG_TRN_TCPL || pq.Q == QG_RND_POS_INF || pq.Q == QG_RND_NEG_INF) && bitsA <= 32 && bitsB <= 32 &&
                      (d->a[0].S || bitsA <= 31) && (d->b[0].S || bitsB <= 31);
        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && wrap32; ++l) {
            const QStep& q = T.level_add[0][l].q;
            const bool pad = (int)l >= T.n_levels && q.identity;
            wrap32 = T.level_cvt[0][l].identity && T.level_add[0][l].sa == 0 && T.level_add[0][l].sb == 0 &&
                     (pad || (!q.identity && q.O == QG_WRP_TCPL && q.S && q.W == 31 && q.d == 0));
        }
        wrap32 = wrap32 && (cq.identity || (cq.d >= 0 && 1 + (int)d->c[0].I + (int)d->c[0].F <= 32));
        if (wrap32) {
            memset(&T.lj, 0, sizeof T.lj);
            T.lj.s = pq.d;
            T.lj.e[1] = 1;
            T.lj.t[0] = pq.d == 0 ? 0 : pq.Q == QG_RND_POS_INF ? (int32_t)((int64_t)1 << (pq.d - 1)) : pq.Q == QG_RND_NEG_INF ? (int32_t)(((int64_t)1 << (pq.d - 1)) - 1) : 0;
            out->tree_fast_ok = 1;
            out->split_s = 0;
            out->mul24_ok = 0;
            out->fast_mode = out->fast_mode_base = 10;
        }
        if (w32) {
            memset(&T.lj, 0, sizeof T.lj);
            T.lj.s = dn;                                                                                       // the product's shift (to the justified word)
            T.lj.e[0] = sj;                                                                                    // bits below the unit in a word (0: 32-bit formats)
            T.lj.t[0] = pq.d == 0 ? 0 : pq.Q == QG_RND_POS_INF ? (int32_t)((int64_t)1 << (pq.d - 1)) : pq.Q == QG_RND_NEG_INF ? (int32_t)(((int64_t)1 << (pq.d - 1)) - 1) : 0;
            out->tree_fast_ok = 1;
            out->split_s = 0;
            out->mul24_ok = 0;
            out->fast_mode = out->fast_mode_base = 10;
        }
    }
    out->tree64_ok = (!cx && !out->wide && !out->generic_only && d->n_levels <= 16) ? 1 : 0;   // (64-bit values: not a wide plan)
    // QG_DESC_LEFTOVER0_COPY with an odd K: the zero-padded kernels would form the leftover as x + 0 in level 0's type — a
    // conversion, where the reference copies — so only the general kernel, which has the leftover step itself, may run it
    const bool copy0 = (d->flags & QG_DESC_LEFTOVER0_COPY) && (d->K & 1) && d->K > 1;
    if (copy0) out->tree_fast_ok = out->fast_mode = out->tree64_ok = out->gemv_ok = out->gemv_wide_ok = 0;
    // (the Qreduce lowering: a * 1 into a's own format.  That is the identity for every raw value EXCEPT -2^W of a signed
    // SAT::SMGN format, which the conversion clamps to -(2^W - 1); the lowerings therefore name a's format with SAT::TCPL
    // as the leaf format of such element types, and only that form takes the shortcut)
    {
        qfmt leaf = d->a[0];
        const bool smgn = leaf.S && leaf.O == QG_SAT_SMGN;
        if (smgn) leaf.O = QG_SAT_TCPL;
        out->gemv_b_bit = ((out->gemv_ok || out->gemv_wide_ok) && d->b[0].I == 1 && d->b[0].F == 0 && !d->b[0].S && same(d->mul[0], leaf)) ? 1 : 0;
    }
    // 32-bit words in the one-column kernel (a Qreduce / GEMV of Q15.16 with default levels): the values fit the element
    // registers and a node is one v_add_i32 ... clamp; the 64-bit-value form of the same kernel spends a 64-bit add and a
    // run-time step per node (65 536 x 4096: 7.6 ms where the bytes take 0.2).  The product is formed in 64 bits by the
    // general step (or not at all: gemv_b_bit), so any rounding / overflow pair into a format of at most 32 bits will do.
    out->gemv_w32 = 0;
    if (out->gemv_wide_ok && d->n_levels >= 1) {
        const int bitsM = (d->mul[0].S ? 1 : 0) + (int)d->mul[0].I + (int)d->mul[0].F;
        bool w32 = bitsM <= (d->mul[0].S ? 32 : 31) && (int)d->mul[0].I + (int)d->mul[0].F >= 0;
        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && w32; ++l) {
            const QStep& q = T.level_add[0][l].q;
            const bool pad = (int)l >= T.n_levels && q.identity;
            w32 = T.level_cvt[0][l].identity && T.level_add[0][l].sa == 0 && T.level_add[0][l].sb == 0 &&
                  (pad || (!q.identity && q.O == QG_SAT_TCPL && q.S && q.d == 0 && q.lo == -((int64_t)1 << 31) && q.hi == ((int64_t)1 << 31) - 1));
        }
        out->gemv_w32 = w32 ? 1 : 0;
    }
    out->gemv_fixed = out->gemv_w32 ? 6 : 0;   // (6: only with gemv_wide_ok, i.e. never together with the forms below)
    if (out->gemv_w32 && !out->gemv_b_bit) {    // 7: ... and the product itself is fast_mode 10's "add a constant, shift right, saturate to the word"
        const QStep& pq = T.mul[0].q;
        if (!pq.identity && pq.O == QG_SAT_TCPL && pq.S && pq.lo == -((int64_t)1 << 31) && pq.hi == ((int64_t)1 << 31) - 1 && pq.d >= 1 && pq.d <= 31 &&
            (pq.Q == QG_TRN_TCPL || pq.Q == QG_RND_POS_INF || pq.Q == QG_RND_NEG_INF))
            out->gemv_fixed = 7;
    }
    if (out->gemv_ok) {
        // all levels one format (the product's), exact alignment (d == 0), SAT::ZERO or SAT::TCPL
        const qfmt lf = d->level_add[0][0];
        bool one = (lf.O == QG_SAT_ZERO || lf.O == QG_SAT_TCPL) && same(lf, d->mul[0]) && (int)lf.I + (int)lf.F <= 29;
        for (uint32_t l = 0; l < d->n_levels && one; ++l)
            one = same(d->level_add[0][l], lf) && same(d->level[0][l], lf) && T.level_add[0][l].q.d == 0;
        if (one) out->gemv_fixed = lf.O == QG_SAT_ZERO ? 1 : 2;
        else if (rec_form) out->gemv_fixed = rec_form;   // per-level formats in compact records (3: every level clamps, 5: kinds)
    }
    out->cplx_fast_ok = (cx && !out->wide && !out->generic_only && !copy0 && d->n_levels <= 16 && c.max_bits <= 31 && c.max_fmt_bits <= 31) ? 1 : 0;
    // fixed-mode variant of the complex kernel (BASELINE configuration 5's "RND + SAT"): every sub-operation and every tree
    // step either the identity, or an exact left shift / a rounding shift with RND::POS_INF, followed by SAT::TCPL, so a
    // step is (v + 2^(d-1)) >> d (or v << -d) and one clamp
    out->cplx_fixed_ok = 0;
    if (out->cplx_fast_ok) {
        auto ok = [](const QStep& q) {
            return q.identity || (q.d >= -29 && q.d <= 29 && q.O == QG_SAT_TCPL && (q.d <= 0 || q.Q == QG_RND_POS_INF));
        };
        bool all = true;
        const int ns = d->cmul == QG_CMUL_TF ? 8 : 6;
        for (int i = 0; i < ns; ++i) all = all && ok(T.mul[i].q);
        for (int p = 0; p < 2; ++p)
            for (uint32_t l = 0; l < d->n_levels; ++l) all = all && ok(T.level_add[p][l].q) && ok(T.level_cvt[p][l]);
        out->cplx_fixed_ok = all ? 1 : 0;
        // ... and 2 when every step fits the branch-free forms of QFix (qg_plan.h): a rounding that is "add a constant, shift
        // right" — TRN::TCPL (+0; the reference's default QuMode), RND::POS_INF (+2^(d-1)), RND::NEG_INF (+2^(d-1) - 1) — and an
        // overflow that is one clamp — SAT::TCPL (the reference's default OfMode) or SAT::SMGN ([-hi, hi]); the values that
        // enter a multiplication or an alignment fit 24 bits (v_mad_i32_i24), alignment and exact left shifts are folded
        // into power-of-two factors of at most 2^22
        auto ok2 = [](const QStep& q) {   // (the other modes as rounding / overflow kinds: cplx_fixed_ok = 3, 8 + features)
            return q.identity || (q.d >= -22 && q.d <= 29 && (q.O == QG_SAT_TCPL || q.O == QG_SAT_SMGN || q.O == QG_SAT_ZERO || (q.O == QG_WRP_TCPL && q.W >= 1)));
        };
        bool kinds = false;
        bool all2 = true;
        for (int i = 0; i < ns; ++i) all2 = all2 && ok2(T.mul[i].q);
        for (int p = 0; p < 2; ++p)
            for (uint32_t l = 0; l < d->n_levels; ++l) all2 = all2 && ok2(T.level_add[p][l].q) && ok2(T.level_cvt[p][l]);
        if (all2) {
            auto b24 = [](qfmt f, int extra) { return 1 + (int)f.I + (int)f.F + extra <= 24; };
            const bool tf = d->cmul == QG_CMUL_TF;
            // slot -> (is a multiplication, format of its first operand x, of its second operand y)
            struct Slot { int idx; bool mul; qfmt x, y; };
            const qfmt a = d->a[0], b = d->a[1], cc = d->b[0], dd = d->b[1];
            std::vector<Slot> slots;
            if (tf) {
                slots = {{QG_T_AB, false, a, b}, {QG_T_CD, false, cc, dd}, {QG_T_BA, false, b, a},
                         {QG_T_A, true, d->mul[QG_T_AB], cc}, {QG_T_B, true, d->mul[QG_T_CD], b}, {QG_T_C, true, d->mul[QG_T_BA], dd},
                         {QG_T_RE, false, d->mul[QG_T_A], d->mul[QG_T_B]}, {QG_T_IM, false, d->mul[QG_T_B], d->mul[QG_T_C]}};
            } else {
                slots = {{QG_B_AC, true, a, cc}, {QG_B_BD, true, b, dd}, {QG_B_AD, true, a, dd}, {QG_B_BC, true, b, cc},
                         {QG_B_RE, false, d->mul[QG_B_AC], d->mul[QG_B_BD]}, {QG_B_IM, false, d->mul[QG_B_AD], d->mul[QG_B_BC]}};
            }
            // kinds: any rounding / overflow kind met; feat: the branch-free feature bits they need (qg_fix.h, fx_finish_feat:
            // 1 R, 2 Z, 4 W), -1 once a kind only the branching form covers was met (unsigned WRP::TCPL, TRN::SMGN by > 23 bits)
            int feat = 0;
            bool bf = false;   // second pass: pack for the branch-free form
            // the rounding / overflow part; returns the rounding factor k of the branch-free form
            auto fix_of = [&kinds, &feat, &bf](const QStep& q, QFix* f) -> int32_t {
                memset(f, 0, sizeof *f);
                f->ka = f->kb = 1;
                if (q.identity) { f->skip = bf ? (1 | 31 << 16 | 1 << 24) : 1; f->lo = INT32_MIN; f->hi = INT32_MAX; return 0; }
                f->lo = q.O == QG_SAT_SMGN ? (q.S ? -(int32_t)q.hi : 0) : (int32_t)q.lo;
                f->hi = (int32_t)q.hi;
                const int ok = q.O == QG_SAT_ZERO ? 1 : q.O == QG_WRP_TCPL ? (q.S ? 2 : 3) : 0;
                int rk = 0;
                if (q.d > 0) rk = q.Q == QG_RND_ZERO ? 1 : q.Q == QG_RND_INF ? 2 : q.Q == QG_RND_CONV ? 3 : q.Q == QG_TRN_SMGN ? 4 : 0;
                kinds = kinds || rk || ok;
                if (ok == 3 || (rk == 4 && q.d > 23)) feat = -1;
                else if (feat >= 0) feat |= (rk ? 1 : 0) | (ok == 1 ? 2 : 0) | (ok == 2 || rk == 2 ? 4 : 0);
                int32_t k = 0;
                if (bf) {
                    const int off = rk == 3 ? q.d : 31;
                    int wd = ok == 2 ? q.W + 1 : 31;
                    // RND::INF adds the INVERTED sign bit: the step's constant carries 2^31, which inverts bit 31 before it is read; the
                    // shifted value is then right in its low 31 - d bits, which the wrap's sign-extraction (feature W) keeps
                    if (rk == 2 && wd > 31 - q.d) wd = 31 - q.d;
                    f->skip = off << 8 | wd << 16 | (ok == 1 ? 0 : 1) << 24;
                    k = rk == 4 ? ((int32_t)1 << q.d) - 1 : rk ? 1 : 0;
                } else {
                    f->skip = (rk << 8) | (ok << 16);   // (qg_fix.h, fx_finish_packed)
                }
                if (q.d < 0) { f->ls = -q.d; return 0; }
                f->d = q.d;
                const int32_t half = q.d ? (int32_t)1 << (q.d - 1) : 0;
                f->t = q.d == 0 ? 0 : q.Q == QG_RND_POS_INF ? half : q.Q == QG_RND_NEG_INF ? half - 1 : (bf && (rk == 1 || rk == 3)) ? half - 1 : (bf && rk == 2) ? INT32_MIN + (half - 1) : 0;
                return k;
            };
            bool reg = true;
            for (int pass = 0; pass < 2 && reg; ++pass) {
                bf = pass == 1;
                memset(T.fmul, 0, sizeof T.fmul);
                for (const Slot& sl : slots) {
                    const QNode& n = T.mul[sl.idx];
                    QFix& f = T.fmul[sl.idx];
                    const int32_t k = fix_of(n.q, &f);
                    const int ls = (!n.q.identity && n.q.d < 0) ? -n.q.d : 0;   // exact left shift after the operation: folded into the factors
                    if (n.q.identity) f.skip &= ~1;   // (the operation itself still runs; only its rounding / overflow is the identity)
                    f.ls = k;                         // (slots fold their shift into the factors: ls carries the rounding factor)
                    auto pow2 = [](int sh) { return sh >= 0 && sh <= 22 ? (int32_t)1 << sh : 0; };   // (0: out of the form's range, see reg)
                    if (sl.mul) {
                        f.ka = pow2(ls);
                        reg = reg && ls <= 22 && b24(sl.x, ls) && b24(sl.y, 0);
                    } else {
                        f.ka = pow2(n.sa + ls);
                        f.kb = pow2(n.sb + ls);
                        reg = reg && n.sa >= 0 && n.sb >= 0 && n.sa + ls <= 22 && n.sb + ls <= 22 && b24(sl.x, 0) && b24(sl.y, 0);
                    }
                }
                for (int p = 0; p < 2 && reg; ++p)
                    for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && reg; ++l) {
                        reg = reg && T.level_add[p][l].sa == 0 && T.level_add[p][l].sb == 0;
                        T.fadd[p][l].ka = fix_of(T.level_add[p][l].q, &T.fadd[p][l]);   // (a left shift at a node is QFix::ls; ka: the rounding factor)
                        T.fcvt[p][l].ka = fix_of(T.level_cvt[p][l], &T.fcvt[p][l]);
                        if (T.level_add[p][l].q.identity) T.fadd[p][l].skip &= ~1;
                    }
                if (!kinds || feat <= 0) break;   // plain compact records, or the branching form: the first packing stands
            }
            if (reg) out->cplx_fixed_ok = !kinds ? 2 : feat > 0 ? 8 + feat : 3;
            // 4 ("one clamp for the whole loop"): the plain compact form where, in addition, every value made in the k loop — the
            // products, their sum / differences and every tree node of both parts — is clamped into ONE range, the sums and the
            // nodes neither shift nor round (equal fraction bits throughout), and the level buffers do not convert.  That is what
            // default tags give on operands whose parts merge into one format (BASELINE configuration 5: everything is int<6,3>
            // RND::POS_INF / SAT::TCPL inside the loop).  The kernel then keeps (lo, hi) and the products' (t, d) in registers for
            // the whole launch — no record loads, no moves of bounds — and the products' exact left shifts are folded into the
            // staged operand planes: 18 vector instructions per complex MAC instead of 24.8 (TF), 21 instead of 27.4 (Basic).
            if (reg && out->cplx_fixed_ok == 2) {
                const int re = tf ? QG_T_RE : QG_B_RE, im = tf ? QG_T_IM : QG_B_IM;
                const QFix& r0 = T.fmul[re];
                bool uni = r0.lo > INT32_MIN && r0.hi < INT32_MAX;
                auto same_clamp = [&](const QFix& f) { return f.lo == r0.lo && f.hi == r0.hi; };
                std::vector<int> prods = tf ? std::vector<int>{QG_T_A, QG_T_B, QG_T_C} : std::vector<int>{QG_B_AC, QG_B_BD, QG_B_AD, QG_B_BC};
                for (int sl : prods) {
                    const QFix& f = T.fmul[sl];
                    uni = uni && same_clamp(f) && f.d >= 0 && f.ka >= 1;
                }
                for (int sl : {re, im}) {
                    const QFix& f = T.fmul[sl];
                    uni = uni && same_clamp(f) && f.ka == 1 && f.kb == 1 && f.t == 0 && f.d == 0;
                }
                for (int p = 0; p < 2 && uni; ++p)
                    for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && uni; ++l) {
                        const QFix &fa = T.fadd[p][l], &fc = T.fcvt[p][l];
                        // (levels that continue a short tree add 0 to a value of the common range: clamping it again changes nothing; an
                        //  identity node of the tree proper is a WIDER format the sum always fits — not this form)
                        const bool ident = (int)l >= T.n_levels && T.level_add[p][l].q.identity != 0;
                        uni = (fc.skip & 1) && (ident || (same_clamp(fa) && fa.t == 0 && fa.d == 0 && fa.ls == 0));
                    }
                if (uni) {
                    memset(&T.uni, 0, sizeof T.uni);
                    T.uni.lo = r0.lo;
                    T.uni.hi = r0.hi;
                    auto lg = [](int32_t k) { int n = 0; while ((1 << n) < k) ++n; return n; };
                    if (tf) {   // every product owns one plane: the plane takes the product's left shift (checked against 24 bits above: b24(x, ls))
                        for (int i = 0; i < 3; ++i) { const QFix& f = T.fmul[prods[i]]; T.uni.t[i] = f.t; T.uni.d[i] = f.d; T.uni.k[i] = f.ka; }
                        T.uni.k[3] = 1;
                    } else {
                        // Basic: a plane serves two products (a: ac, ad; b: bd, bc; c: ac, bc; d: bd, ad).  Plane shifts la, lb, lc, ld with
                        // lx + ly >= the product's left shift; what they exceed it by is shifted out again after the rounding addend (scaled
                        // alike): (x y 2^(lx+ly) + t 2^e) >> (d + e) = (x y 2^ls + t) >> d.  The smallest total that keeps planes within
                        // 24 bits and products within 31.
                        const int px[4] = {0, 1, 0, 1}, py[4] = {2, 3, 3, 2};   // ac, bd, ad, bc over planes a b c d
                        const qfmt pf[4] = {a, b, cc, dd};
                        int w[4], ls[4], best[4] = {0, 0, 0, 0}, best_sum = -1;
                        for (int i = 0; i < 4; ++i) { w[i] = 1 + (int)pf[i].I + (int)pf[i].F; ls[i] = lg(T.fmul[prods[i]].ka); }
                        for (int la = 0; la <= 12; ++la)
                            for (int lb = 0; lb <= 12; ++lb)
                                for (int lc = 0; lc <= 12; ++lc)
                                    for (int ld = 0; ld <= 12; ++ld) {
                                        const int l[4] = {la, lb, lc, ld};
                                        bool ok = true;
                                        for (int i = 0; i < 4 && ok; ++i) ok = w[i] + l[i] <= 24;
                                        for (int i = 0; i < 4 && ok; ++i) {
                                            const int e = l[px[i]] + l[py[i]] - ls[i];
                                            ok = e >= 0 && w[px[i]] + w[py[i]] + l[px[i]] + l[py[i]] <= 30 && T.fmul[prods[i]].d + e <= 30;
                                        }
                                        const int sum = la + lb + lc + ld;
                                        if (ok && (best_sum < 0 || sum < best_sum)) { best_sum = sum; memcpy(best, l, sizeof best); }
                                    }
                        uni = best_sum >= 0;
                        for (int i = 0; i < 4 && uni; ++i) {
                            const QFix& f = T.fmul[prods[i]];
                            const int e = best[px[i]] + best[py[i]] - ls[i];
                            T.uni.t[i] = (int32_t)((uint32_t)f.t << e);
                            T.uni.d[i] = f.d + e;
                            T.uni.k[i] = (int32_t)1 << best[i];
                        }
                    }
                }
                if (uni) out->cplx_fixed_ok = 4;
                // The justified forms below also take products with FEWER fraction bits than the common format but the same integer
                // bits (BasicComplexMul on Qcomplex<int<6,3>, int<6,-3>>: b d lives in int<6,-3>, RE = ac - 64 bd): left-justified, such
                // a product's range ends where the common range does, its alignment factor in RE / IM is implicit, and only its mask
                // (which floors to ITS unit) differs.  g[i] = log2 of that factor.
                int gfac[4] = {0, 0, 0, 0};
                bool ru = r0.lo > INT32_MIN && r0.hi < INT32_MAX;
                {
                    auto lg2 = [](int32_t k) { int n = 0; while (n < 31 && ((int32_t)1 << n) < k) ++n; return ((int32_t)1 << n) == k ? n : -1; };
                    const QFix &fr = T.fmul[re], &fi = T.fmul[im];
                    // the factor each product is aligned with: TF re = A - B, im = B - C; Basic re = ac - bd, im = ad + bc
                    const int32_t kf[4] = {fr.ka, fr.kb, tf ? fi.kb : fi.ka, tf ? 0 : fi.kb};
                    if (tf) ru = ru && fi.ka == fr.kb;   // (B enters both with one factor)
                    for (size_t i = 0; i < prods.size() && ru; ++i) {
                        const QFix& f = T.fmul[prods[i]];
                        gfac[i] = lg2(kf[i]);
                        ru = gfac[i] >= 0 && gfac[i] <= 16 && f.d >= 0 && f.ka >= 1 && f.lo > INT32_MIN && f.hi < INT32_MAX &&
                             (int64_t)f.lo * kf[i] == r0.lo && ((int64_t)f.hi + 1) * kf[i] == (int64_t)r0.hi + 1;
                    }
                    for (const QFix* f : {&fr, &fi}) ru = ru && same_clamp(*f) && f->t == 0 && f->d == 0;
                    for (int p = 0; p < 2 && ru; ++p)
                        for (uint32_t l = 0; l < (uint32_t)T.n_levels_k && ru; ++l) {
                            const QFix &fa = T.fadd[p][l], &fc = T.fcvt[p][l];
                            const bool ident = (int)l >= T.n_levels && T.level_add[p][l].q.identity != 0;
                            ru = (fc.skip & 1) && (ident || (same_clamp(fa) && fa.t == 0 && fa.d == 0 && fa.ls == 0));
                        }
                }
                out->cplx_fixed_base = uni ? 4 : 2;
                // 5: ... and that one range is a signed SAT::TCPL format: left-justified values (qg_fix.h).  The planes are staged with the
                // left shifts that justify each product exactly (x y 2^(s + ls - d), the addend scaled alike), so a product is one
                // saturating multiply-add + v_and, RE / IM one saturating add / subtract, a node one saturating add (+ v_and at the
                // even levels): 12.5 instead of 18 vector instructions per complex MAC (TF).
                if (ru && r0.lo == -r0.hi - 1 && ((r0.hi + 1) & r0.hi) == 0 && r0.hi > 0) {
                    int Wt = 1;
                    while (((int64_t)1 << (Wt - 1)) <= r0.hi) ++Wt;
                    auto lg = [](int32_t k) { int n = 0; while ((1 << n) < k) ++n; return n; };
                    auto wd = [](qfmt f) { return 1 + (int)f.I + (int)f.F; };
                    // word: 32 (operands of v_mad_i32_i24: 24 bits) or 16 (halves of v_pk_mad_i16: 16 bits)
                    auto justify = [&](int word, int opbits, QJustify* J) {
                        const int sj = word - Wt;
                        bool lj = sj >= 1;
                        memset(J, 0, sizeof *J);
                        J->s = sj;
                        int E[4] = {0, 0, 0, 0};
                        for (size_t i = 0; i < prods.size() && lj; ++i) {
                            const QFix& f = T.fmul[prods[i]];
                            E[i] = sj + gfac[i] + lg(f.ka) - f.d;
                            lj = E[i] >= 0 && (f.d == 0 || sj + gfac[i] - f.d >= 0);
                            if (lj) J->t[i] = f.d ? (int32_t)((uint32_t)f.t << (sj + gfac[i] - f.d)) : 0;
                            J->g[i] = gfac[i];
                        }
                        if (lj && tf) {
                            // products A = (a+b) c, B = (c+d) b, C = (b-a) d: every plane serves one product
                            const qfmt fx[3] = {d->mul[QG_T_AB], d->mul[QG_T_CD], d->mul[QG_T_BA]}, fy[3] = {cc, b, dd};
                            const int px[3] = {0, 4, 2}, py[3] = {3, 1, 5};   // planes (a+b), b, (b-a), c, (c+d), d
                            for (int i = 0; i < 3 && lj; ++i) {
                                int ex = E[i] < opbits - wd(fx[i]) ? E[i] : opbits - wd(fx[i]);
                                if (ex < 0) ex = 0;
                                const int ey = E[i] - ex;
                                lj = wd(fx[i]) + ex <= opbits && wd(fy[i]) + ey <= opbits;
                                J->e[px[i]] = ex;
                                J->e[py[i]] = ey;
                            }
                        } else if (lj) {
                            const qfmt pf[4] = {a, b, cc, dd};   // ac, bd, ad, bc over planes a b c d
                            bool found = false;
                            for (int la = 0; la <= opbits - wd(pf[0]) && !found; ++la) {
                                const int lc = E[0] - la, ld = E[2] - la, lb = E[3] - lc;
                                if (lc < 0 || ld < 0 || lb < 0 || lb + ld != E[1]) continue;
                                if (wd(pf[1]) + lb > opbits || wd(pf[2]) + lc > opbits || wd(pf[3]) + ld > opbits) continue;
                                J->e[0] = la; J->e[1] = lb; J->e[2] = lc; J->e[3] = ld;
                                found = true;
                            }
                            lj = found;
                        }
                        return lj;
                    };
                    if (justify(32, 24, &T.lj)) {
                        out->cplx_fixed_ok = 5;
                        // 6: ... in packed 16-bit halves, two outputs per register (k_tree_cplx_pk16)
                        if (justify(16, 16, &T.lj16)) out->cplx_fixed_ok = 6;
                    }
                }
            }
        }
    }
    if (!out->linear_ok)
        snprintf(out->reason, sizeof out->reason, "%s",
                 "a product or tree node may round or overflow: exact tree evaluation");
}
