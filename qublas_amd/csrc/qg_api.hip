// qg_api.hip — the C-ABI of include/qgemul.h: contexts, plans, packing, execution.
//
// There is deliberately no CPU arithmetic path in this library: without a gfx950 device every
// compute entry point returns QG_ENOGPU.  (The CPU restatement lives in oracle/ and is test
// infrastructure only.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/qgemul.h"
#include "qg_kernels.h"
#include "qg_plan.h"

static thread_local int g_last_hip = 0;

#define QG_HIP(expr)                         \
    do {                                     \
        hipError_t e_ = (expr);              \
        if (e_ != hipSuccess) {              \
            g_last_hip = (int)e_;            \
            return QG_EHIP;                  \
        }                                    \
    } while (0)

struct qgemul_ctx {
    int device;
    hipStream_t stream;
    int* flag_dev;
};

// Every entry point that launches, allocates or frees runs on ITS context's device, whatever device the calling thread has
// current, and leaves the caller's current device as it found it (one process may drive several GPUs: qgemul_run_sharded).
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
        }
    }
    ~DeviceScope() { if (changed) hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define QG_ON_DEVICE(ctxp)                 \
    DeviceScope dev_scope_((ctxp)->device); \
    QG_HIP(dev_scope_.err)

// Composite linear plan.  The MFMA kernels take operands of at most 3 int8 limbs and keep exact int32 accumulators only while
// K * min(LA, LB) < 2^17.  Beyond either bound the linear class used to fall to the 64-bit VALU tree kernel (40-300x slower); the
// reference has no such boundary (Reducer, QuBLAS.h:4960-4990; ArbiInt elements up to 64 bits, :347-564).  Now:
//   * an operand of L > 3 limbs is stored as limb GROUPS of 2-3 limbs (x = sum_g x_g * 256^limb0[g], every x_g a balanced
//     base-256 number of its own), each group a complete packed operand of the existing layout;
//   * K is cut into chunks of at most kc reduction indices (a multiple of 256), each chunk a complete packed operand as well;
//   * every (chunk, A group, B group) is ONE launch of an existing MFMA kernel that stores raw dot products (identity
//     epilogue) into a slab, and k_lin_combine (qg_pack.hip) adds the slabs by weight into an exact running sum and, after the
//     last chunk, rounds + overflow-handles it once into C — the linear class's whole epilogue (QuBLAS.h:2398-2411).
// Packed operand = the sub-operands back to back, chunk-major, each 256-byte aligned.
struct QComposite {
    int on;
    int ga, gb;            // limb groups of A, B (1..3)
    int la[3], lb[3];      // limbs per group
    int la0[3], lb0[3];    // first limb of each group
    int var[3][3];         // MFMA variant of the pair (A group, B group); one tile geometry for all pairs
    int nc;                // k-chunks
    int64_t kc;            // reduction indices per chunk (the last chunk: K - (nc - 1) * kc)
    int slab_bytes;        // 4: single-limb pairs (raw int32), 8 otherwise
    int wide;              // 128-bit sums
    int64_t chunk_bytes[2];   // bytes of one FULL chunk of packed A / B (all groups)
};

struct qgemul_plan {
    qgemul_ctx* ctx;
    qgemul_desc desc;
    uint32_t flags;
    QAnalysis an;
    qgemul_info info;
    int LA, LB, variant;
    QPackedGeom pa, pb;
    QCGeom pc;
    QHostElem ha, hb, hc;
    QTreeTable* dev_table;
    int64_t* workspace;   // complex linear class: raw dot products [2Mh x 2Nh] int64
    QMfmaCfg cfg;
    // element-wise epilogue (qgemul_epilogue): pc then describes packed D; pc_c is the kernel's own packed C, which
    // only exists in memory (cwork) for the kernels that do not fuse the chain
    int has_ep;
    qgemul_epilogue ep;
    QEpTable ept;
    // complex chain (qgemul_epilogue_cplx): ep / ept are the chain of the real parts, ep_im / ept_im of the imaginary parts
    int ep_cplx;
    qgemul_epilogue ep_im;
    QEpTable ept_im;
    uint8_t e_cplx[QG_MAX_EW];
    QCGeom pc_c;
    void* cwork;
    int32_t* wide_ws;     // single-limb MFMA with a left-shifting epilogue that leaves 32 bits: raw int32 dot products
    void* hostc_pc;       // qgemul_execute_host_c on a kernel that cannot store the reference layout: its packed C
    QComposite comp;      // composite linear plan (comp.on): limb groups x k-chunks of sub-GEMMs + an exact combine pass
    void* comp_slabs;     // comp.ga * comp.gb slabs of raw dot products, one common packed-C layout
    void* comp_acc;       // running exact sums between k-chunks (comp.nc > 1)
};

struct HostC { void* C; int64_t ld; };
// an element-wise chain as the planner sees it: a real chain (im == nullptr) or the two part chains of a complex one
struct EpView { const qgemul_epilogue* re; const qgemul_epilogue* im; const uint8_t* e_cplx; };   // execute_kernel: store the reference layout directly (kernels that can)

static int pow2_bytes(int storage_bits)
{
    int b = (storage_bits + 7) / 8;
    int c = 1;
    while (c < b) c *= 2;
    return c;
}

static bool same_fmt(const qfmt& x, const qfmt& y) { return x.I == y.I && x.F == y.F && x.S == y.S && x.Q == y.Q && x.O == y.O; }
static int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- composite linear plans: geometry helpers (see QComposite) ----
static int64_t comp_sub_bytes(int limbs, int64_t rows_p, int64_t K_p)
{
    return round_up((int64_t)limbs * rows_p * K_p + (limbs > 1 ? QG_TRAILER_BYTES : 0), 256);
}
// reduction indices of chunk c
static int64_t comp_chunk_len(const QComposite& q, int64_t K, int c) { return c + 1 < q.nc ? q.kc : K - (int64_t)(q.nc - 1) * q.kc; }
// the packed sub-operand (chunk c, group g) of operand `which` (0 A, 1 B): its geometry and its byte offset in the packed operand
static QPackedGeom comp_sub_geom(const QComposite& q, const QPackedGeom& base, int which, int64_t K, int c, int g, int64_t* off)
{
    const int* L = which ? q.lb : q.la;
    const int* L0 = which ? q.lb0 : q.la0;
    const int ng = which ? q.gb : q.ga;
    const int64_t K_p = round_up(comp_chunk_len(q, K, c), base.bk);
    QPackedGeom s = base;
    s.K_p = K_p;
    s.limbs = L[g];
    s.limb0 = L0[g];
    s.trailer = L[g] > 1 ? (int64_t)L[g] * base.rows_p * K_p : 0;
    s.digit6 = 0;
    int64_t o = (int64_t)c * q.chunk_bytes[which];
    for (int h = 0; h < g && h < ng; ++h) o += comp_sub_bytes(L[h], base.rows_p, K_p);
    if (off) *off = o;
    if (base.offs) {   // centred operand: ONE row-sum array behind all sub-operands; the groups that hold digit 0 add their k-chunk to it
        s.offs = L0[g] == 0 ? 2 : 3;
        s.rowsum_off = base.rowsum_off - o;
    }
    return s;
}
// groups of 2-3 limbs, low limbs first
static int comp_split(int L, int* sizes, int* first)
{
    static const int tab[9][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {2, 2, 0}, {3, 2, 0}, {3, 3, 0}, {3, 2, 2}, {3, 3, 2}};
    if (L < 1 || L > 8) return 0;
    int n = 0, f = 0;
    for (int i = 0; i < 3 && tab[L][i]; ++i) { sizes[n] = tab[L][i]; first[n] = f; f += tab[L][i]; ++n; }
    return n;
}
static bool comp_geometry(int LA, int LB, const qgemul_desc* d, uint32_t flags, QComposite* q, QMfmaCfg* cfg)
{
    memset(q, 0, sizeof *q);
    q->ga = comp_split(LA, q->la, q->la0);
    q->gb = comp_split(LB, q->lb, q->lb0);
    if (!q->ga || !q->gb) return false;
    int maxa = 0, maxb = 0;
    bool single = true;
    for (int i = 0; i < q->ga; ++i)
        for (int j = 0; j < q->gb; ++j) {
            const QMfmaCfg c = qg_mfma_pick(q->la[i], q->lb[j], d->M, d->N, flags);
            if (!c.variant) return false;
            if (i + j == 0) *cfg = c;
            else if (c.TM != cfg->TM || c.TN != cfg->TN || c.BK != cfg->BK) return false;   // (one packed layout for all pairs)
            q->var[i][j] = c.variant;
            if (q->la[i] > maxa) maxa = q->la[i];
            if (q->lb[j] > maxb) maxb = q->lb[j];
            single = single && q->la[i] == 1 && q->lb[j] == 1;
        }
    // int32 accumulators of a limb weight collect min(la, lb) products of magnitude <= 2^14 per reduction index
    const int mn = maxa < maxb ? maxa : maxb;
    const int64_t kmax = (((1ll << 17) - 1) / mn) / 256 * 256;
    q->nc = (int)((d->K + kmax - 1) / kmax);
    q->kc = q->nc > 1 ? kmax : d->K;
    q->slab_bytes = single ? 4 : 8;
    q->on = 1;
    return true;
}

// composite plans: run `fn` on every (k-chunk, limb group) sub-operand of A or B — its view of the host tensor (the chunk's
// reduction indices), its packed geometry, its byte offset inside the packed operand
template <class F>
static hipError_t comp_for_each_sub(const qgemul_plan* p, int operand, const QOperandGeom& g, F fn)
{
    const QComposite& q = p->comp;
    const int w = operand == QG_OPERAND_A ? 0 : 1;
    const QPackedGeom& base = w ? p->pb : p->pa;
    for (int c = 0; c < q.nc; ++c)
        for (int gi = 0; gi < (w ? q.gb : q.ga); ++gi) {
            int64_t off = 0;
            const QPackedGeom sp = comp_sub_geom(q, base, w, p->desc.K, c, gi, &off);
            QOperandGeom sg = g;
            sg.k0 = (int64_t)c * q.kc;
            sg.K = comp_chunk_len(q, p->desc.K, c);
            if (hipError_t e = fn(sg, sp, off); e != hipSuccess) return e;
        }
    return hipSuccess;
}

// composite plan on a centred operand: its one row-sum array is zeroed before the sub-operands' packs add to it
static hipError_t comp_zero_rowsums(const qgemul_plan* p, int operand, void* packed_dev)
{
    const QPackedGeom& base = operand == QG_OPERAND_A ? p->pa : p->pb;
    if (!p->comp.on || !base.offs) return hipSuccess;
    return hipMemsetAsync((char*)packed_dev + base.rowsum_off, 0, (size_t)base.rows_p * 8, p->ctx->stream);
}

// fill info + geometry for a descriptor; no GPU access
static int plan_geometry(const qgemul_desc* d, uint32_t flags, QAnalysis* an, qgemul_info* info, int* pLA, int* pLB, QMfmaCfg* pVar,
                         QPackedGeom* pa, QPackedGeom* pb, QCGeom* pc, QHostElem* ha, QHostElem* hb, QHostElem* hc,
                         const EpView* ev = nullptr, QEpTable* ept = nullptr, QCGeom* pc_c = nullptr, QEpTable* ept_im = nullptr,
                         QComposite* pcomp = nullptr)
{
    QComposite comp_local;
    QComposite& comp = pcomp ? *pcomp : comp_local;
    memset(&comp, 0, sizeof comp);
    const qgemul_epilogue* ep = ev ? ev->re : nullptr;
    qg_analyze(d, an);
    memset(info, 0, sizeof *info);
    snprintf(info->reason, sizeof info->reason, "%s", an->reason);
    if (an->status != QG_OK) {
        info->supported = 0;
        return an->status;
    }
    const int parts = d->is_complex ? 2 : 1;
    *ha = qg_host_elem(d->a, d->is_complex);
    *hb = qg_host_elem(d->b, d->is_complex);
    *hc = qg_host_elem(d->c, d->is_complex);
    info->cls = an->cls;
    info->max_bits = an->max_bits;
    info->supported = 1;
    auto sbits = [&](const qfmt* f) {
        int m = 0;
        for (int p = 0; p < parts; ++p) {
            int b = 1 + (int)f[p].I + (int)f[p].F;
            if (b > m) m = b;
        }
        return m;
    };
    info->in_bits[0] = sbits(d->a);
    info->in_bits[1] = sbits(d->b);
    info->host_elem_bytes[0] = ha->size;
    info->host_elem_bytes[1] = hb->size;
    info->host_elem_bytes[2] = hc->size;
    info->host_imag_off[0] = ha->off[1];
    info->host_imag_off[1] = hb->off[1];
    info->host_imag_off[2] = hc->off[1];
    const double mnk = (double)d->M * (double)d->N * (double)d->K;
    info->ops = !d->is_complex ? 2.0 * mnk : (d->cmul == QG_CMUL_TF ? 6.0 * mnk : 8.0 * mnk);

    int LA = 0, LB = 0, kernel = QG_KERNEL_NONE;
    int64_t centreA = 0, centreB = 0;
    bool centred = false;
    QMfmaCfg cfg = {0, 0, 0, 0};
    if (an->linear_ok && !(flags & QG_OPT_FORCE_TREE)) {
        LA = qg_limbs_for(d->a[0]);
        LB = qg_limbs_for(d->b[0]);
        if (d->is_complex) {  // parts are stacked along the row axis and share one limb count
            const int la2 = qg_limbs_for(d->a[1]), lb2 = qg_limbs_for(d->b[1]);
            if (la2 > LA) LA = la2;
            if (lb2 > LB) LB = lb2;
        }
        // CENTRED operands (qg_plan.h: qg_limbs_centred; QPackedGeom::offs): x - c in balanced limbs where that saves a limb — signed
        // formats of 16 / 24 / 32 bits (3 -> 2, 4 -> 3, 5 -> 4 limbs), unsigned formats (uint8: 2 -> 1) — the centre taken back out
        // with row sums after the dot product.  Real descriptors only (stacked complex parts have different centres).
        if (!d->is_complex && !(flags & QG_OPT_BALANCED_LIMBS)) {
            int64_t c = 0;
            int l = qg_limbs_centred(d->a[0], &c);
            if (l < LA) { LA = l; centreA = c; centred = true; }
            l = qg_limbs_centred(d->b[0], &c);
            if (l < LB) { LB = l; centreB = c; centred = true; }
            // the row sums are int64.  A 64-bit plan may let them wrap (its whole correction is arithmetic modulo 2^64 and the sum
            // itself fits); a wide plan (128-bit combine) may not: there an operand's row sums must fit — |x'| < 2^(8 L - 1) * 1.01,
            // K of them (found by the wide fuzzer: a 56-bit operand over K = 43 519 next to a centred 32-bit one)
            if (centred && an->wide) {
                int lgk = 0;
                while (((int64_t)1 << lgk) < d->K) ++lgk;
                const bool fitA = 8 * LA + lgk <= 62, fitB = 8 * LB + lgk <= 62;
                if ((centreB != 0 && !fitA) || (centreA != 0 && !fitB)) {
                    LA = qg_limbs_for(d->a[0]);
                    LB = qg_limbs_for(d->b[0]);
                    centreA = centreB = 0;
                    centred = false;
                }
            }
        }
        const int mn = LA < LB ? LA : LB;
        cfg = qg_mfma_pick(LA, LB, d->M * parts, d->N * parts, (ep ? QG_OPT_LOCKSTEP_TILES : 0u) | flags);   // (the fused / unfused element-wise chain keeps the kernel it was measured on)
        // (wide plans: the kernels' own epilogues are 64-bit; the composite plan's combine pass is not.  Centred single-limb pairs: the
        //  single-limb kernels' epilogue is 32-bit, sum a b of two uint8 operands is not: raw int32 slab + combine pass)
        // (... except where its image in C fits 31 bits and no element-wise chain follows: the single-limb kernels' epilogues then restore
        //  the sum in 64 bits themselves)
        const QStep& tc = an->lin.to_c[0];
        const bool pp_centred = cfg.variant && !ep && sbits(d->c) <= 31 && (tc.identity || (tc.d >= 0 && tc.W <= 30));   // (every single-limb kernel has the 64-bit branch)
        if (cfg.variant && (int64_t)mn * d->K <= (1ll << 17) - 1 && !an->wide && !an->generic_only && !(centred && LA == 1 && LB == 1 && !pp_centred))
            kernel = d->is_complex ? QG_KERNEL_MFMA_CPLX : ((LA == 1 && LB == 1) ? QG_KERNEL_MFMA_I8 : QG_KERNEL_MFMA_I8_LIMB);
        // (one output column whose tree the one-column kernels can walk: those stream A once at HBM rate; a composite plan would read
        // it once per limb group and write slabs — the batched Qreduce of 32-bit words with exact level types stays there)
        else if (!d->is_complex && !(d->N == 1 && (an->gemv_ok || an->gemv_wide_ok) && !(flags & QG_OPT_GENERIC_TREE)) &&
                 comp_geometry(LA, LB, d, flags, &comp, &cfg)) {
            // more than 3 limbs, or K beyond the int32 accumulators' exact range: limb groups x k-chunks on the same kernels
            kernel = (LA == 1 && LB == 1) ? QG_KERNEL_MFMA_I8 : QG_KERNEL_MFMA_I8_LIMB;
            comp.wide = an->wide;
            snprintf(info->reason, sizeof info->reason, "linear class: %d k-chunk(s) x %d x %d limb groups on the MFMA kernels, exact %s combine", comp.nc, comp.ga, comp.gb,
                     comp.wide ? "128-bit" : "64-bit");
        } else {
            LA = LB = 0;
            centred = false;
            snprintf(info->reason, sizeof info->reason, "linear class, but limbs/K outside the MFMA kernels' range: tree kernel");
        }
    }
    pc->M = d->M;
    pc->N = d->N;
    pc->parts = parts;
    pc->cbytes = pow2_bytes(sbits(d->c));
    // (a multi-word value in the band the reference mis-compares comes out as its low WORD — int32_t / int64_t — whatever C's
    // format says: such plans keep the host element's width in packed C)
    if (an->band && pc->cbytes < (hc->sb[0] > hc->sb[1] ? hc->sb[0] : hc->sb[1])) pc->cbytes = hc->sb[0] > hc->sb[1] ? hc->sb[0] : hc->sb[1];
    pc->ldc = d->M;
    pc->elem_bytes = hc->size;
    for (int p = 0; p < 2; ++p) { pc->off[p] = hc->off[p]; pc->sb[p] = hc->sb[p]; }
    if (kernel != QG_KERNEL_NONE) {
        *pa = QPackedGeom{round_up(d->M, cfg.TM), round_up(d->K, cfg.BK), 1, LA, cfg.TM, cfg.BK};
        *pb = QPackedGeom{round_up(d->N, cfg.TN), round_up(d->K, cfg.BK), 1, LB, cfg.TN, cfg.BK};
        if (comp.on) {   // pa / pb: the first sub-operand (group 0 of chunk 0); the others through comp_sub_geom
            pa->K_p = pb->K_p = round_up(comp.kc, cfg.BK);
            pa->limbs = comp.la[0];
            pb->limbs = comp.lb[0];
        }
        pc->Mp = pa->rows_p;
        pc->Np = pb->rows_p;
        pc->tm = cfg.TM;
        pc->tn = cfg.TN;
        if (d->is_complex) {
            // the MFMA kernel writes the raw 2Mh x 2Nh dot products into the plan's workspace; the combine
            // pass writes the packed complex C row-major [2][M][N]
            pc->Mp = d->M;
            pc->Np = d->N;
            pc->tm = pc->tn = 0;
        }
    } else {
        const bool fast = !(flags & QG_OPT_GENERIC_TREE);
        // 32-bit words on the one-column kernel (QAnalysis::gemv_w32): rows of at least 256 leaves; like fast_mode 10 it has no
        // run-time-mode form, QG_OPT_RUNTIME_MODES keeps the 64-bit-value form (the tests' second opinion)
        const bool gemv_w32 = an->gemv_wide_ok && an->gemv_w32 && !(flags & QG_OPT_RUNTIME_MODES) && an->tree.n_levels_k >= 8;
        kernel = an->wide ? QG_KERNEL_TREE_I128 : d->is_complex ? ((an->cplx_fast_ok && fast) ? QG_KERNEL_TREE_CPLX_I32 : QG_KERNEL_TREE_CPLX)
                               : (((an->gemv_ok || gemv_w32) && fast) ? QG_KERNEL_GEMV_I32 : (an->gemv_wide_ok && fast) ? QG_KERNEL_GEMV_I64
                                  : (an->tree_fast_ok && fast && !(an->fast_mode == 10 && (flags & QG_OPT_RUNTIME_MODES))) ? QG_KERNEL_TREE_I32 : QG_KERNEL_TREE_I64);   // (fast_mode 10, 32-bit words: no run-time-mode form on that kernel)
        // the 32-bit tree kernels walk a perfect binary tree: their operands are zero-padded along K to 2^n_levels leaves
        // (a node whose right child is a zero leaf / zero subtree is the reference's converting copy of an odd leftover)
        const bool t64 = kernel == QG_KERNEL_TREE_I64 && an->tree64_ok && fast;   // the 2x2-per-lane 64-bit kernel, not the general one
        const int64_t Kt = (kernel == QG_KERNEL_TREE_I32 || kernel == QG_KERNEL_TREE_CPLX_I32 || kernel == QG_KERNEL_GEMV_I32 || kernel == QG_KERNEL_GEMV_I64 || t64)
                               ? ((int64_t)1 << an->tree.n_levels_k) : d->K;   // (n_levels_k: at least 5 levels, qg_plan.h)
        *pa = QPackedGeom{d->M, Kt, info->in_bits[0] <= 32 ? 4 : 8, 0, 0, 0};
        *pb = QPackedGeom{d->N, Kt, info->in_bits[1] <= 32 ? 4 : 8, 0, 0, 0};
        pc->Mp = d->M;
        pc->Np = d->N;
        pc->tm = pc->tn = 0;
    }
    info->kernel = kernel;
    if (kernel == QG_KERNEL_TREE_I128)
        snprintf(info->reason, sizeof info->reason, "exact tree evaluation on 128-bit values (intermediates of %d bits)", an->max_bits);
    if (kernel == QG_KERNEL_TREE_I32) {
        const int fm = (flags & QG_OPT_RUNTIME_MODES) ? 0 : an->fast_mode;
        snprintf(info->reason, sizeof info->reason, "exact tree evaluation; tree kernel steps: %s",
                 fm == 10 ? (an->tree.lj.e[1] ? "one 32-bit format, WRP::TCPL, wrapping word adds" : an->tree.lj.e[0] > 0 ? "one format, SAT::TCPL, justified words" : "one 32-bit format, SAT::TCPL, saturating word adds") : fm >= 8 ? "one format, SAT::TCPL, left-justified, packed nodes" : fm == 7 ? "one format, SAT::TCPL, left-justified, packed 16-bit" : fm == 6 ? "one format, SAT::TCPL, left-justified" : fm == 1 ? "one format, SAT::ZERO" : fm == 2 ? "one format, SAT::TCPL" : fm == 3 ? "per-level formats, compact (clamps)" : fm == 4 ? "per-level formats, compact" : fm == 5 ? "per-level formats, compact (unbiased)" : "run-time modes");
    }
    if (kernel == QG_KERNEL_GEMV_I64)
        snprintf(info->reason, sizeof info->reason, "exact tree evaluation; one-column kernel steps: run-time modes, 64-bit values");
    if (kernel == QG_KERNEL_GEMV_I32) {
        const int fm = (flags & QG_OPT_RUNTIME_MODES) ? 0 : an->gemv_fixed;
        snprintf(info->reason, sizeof info->reason, "exact tree evaluation; one-column kernel steps: %s",
                 fm == 1 ? "one format, SAT::ZERO" : fm == 2 ? "one format, SAT::TCPL" : fm == 3 ? "per-level formats, compact (clamps)" : fm == 5 ? "per-level formats, compact" : (fm == 6 || fm == 7) ? "one 32-bit format, saturating adds" : "run-time modes");
    }
    if (kernel == QG_KERNEL_TREE_CPLX_I32) {
        // which form of the complex kernel's steps this descriptor gets (tests assert their coverage through it)
        const int fx = (flags & QG_OPT_RUNTIME_MODES) ? 0 : an->cplx_fixed_ok;
        snprintf(info->reason, sizeof info->reason, "exact tree evaluation; complex kernel steps: %s",
                 fx >= 8 ? "compact, branch-free rounding / overflow kinds" : fx == 6 ? "fixed modes, one clamp, packed 16-bit" : fx == 5 ? "fixed modes, one clamp, left-justified" : fx == 4 ? "fixed modes, one clamp for the whole loop" : fx == 3 ? "compact, rounding / overflow kinds" : fx == 2 ? "fixed modes, compact" : fx == 1 ? "fixed modes, table" : "run-time modes");
    }
    info->limbs[0] = LA;
    info->limbs[1] = LB;
    info->packed_bytes[0] = (int64_t)parts * (pa->limbs ? pa->limbs : 1) * pa->rows_p * pa->K_p * pa->cbytes;
    info->packed_bytes[1] = (int64_t)parts * (pb->limbs ? pb->limbs : 1) * pb->rows_p * pb->K_p * pb->cbytes;
    // multi-limb operands carry their plane mask in a trailer behind the planes (QPackedGeom::trailer)
    if (pa->limbs > 1) { pa->trailer = info->packed_bytes[0]; info->packed_bytes[0] += QG_TRAILER_BYTES; }
    if (pb->limbs > 1) { pb->trailer = info->packed_bytes[1]; info->packed_bytes[1] += QG_TRAILER_BYTES; }
    if (comp.on) {
        // the packed operand = every (chunk, group) sub-operand back to back
        for (int w = 0; w < 2; ++w) {
            const QPackedGeom& base = w ? *pb : *pa;
            const int ng = w ? comp.gb : comp.ga;
            const int* L = w ? comp.lb : comp.la;
            int64_t full = 0, last = 0;
            const int64_t Kl = round_up(comp_chunk_len(comp, d->K, comp.nc - 1), base.bk);
            for (int g = 0; g < ng; ++g) { full += comp_sub_bytes(L[g], base.rows_p, base.K_p); last += comp_sub_bytes(L[g], base.rows_p, Kl); }
            comp.chunk_bytes[w] = full;
            info->packed_bytes[w] = (int64_t)(comp.nc - 1) * full + last;
        }
        pa->trailer = pb->trailer = 0;   // (per sub-operand: comp_sub_geom)
        if ((int64_t)comp.ga * comp.gb * pa->rows_p * pb->rows_p * comp.slab_bytes > (64ll << 30)) {
            info->supported = 0;
            snprintf(info->reason, sizeof info->reason, "composite linear plan: the slabs of raw dot products would exceed 64 GiB");
            return QG_EUNSUPPORTED;
        }
    }
    if (centred && kernel != QG_KERNEL_NONE) {   // both operands carry row sums (an uncentred partner: bias 0), behind everything else
        pa->offs = pb->offs = 1;
        pa->bias = -centreA;
        pb->bias = -centreB;
        pa->rowsum_off = round_up(info->packed_bytes[0], 256);
        pb->rowsum_off = round_up(info->packed_bytes[1], 256);
        info->packed_bytes[0] = pa->rowsum_off + pa->rows_p * 8;
        info->packed_bytes[1] = pb->rowsum_off + pb->rows_p * 8;
    }
    // Karatsuba (qg_mfma.hip, KARA): two-limb operands whose biased values fit 12 bits are stored as two unsigned base-64
    // digits each, and the product takes 3 MFMAs per k-step instead of 4
    if (!comp.on && !centred) {
        static const bool no_kara = QG_DIAG_ENV("QG_NO_KARA");   // A/B switch
        auto ubits = [](qfmt f) { return (int)f.I + (int)f.F + (f.S ? 1 : 0); };
        // (problems small enough for the 64x64 tiles are latency-bound: measured 9.5 vs 8.9 us at 1024^3, schoolbook kept there)
        if (!no_kara && !d->is_complex && LA == 2 && LB == 2 && (kernel == QG_KERNEL_MFMA_I8_LIMB) && (cfg.variant == 3 || cfg.variant == 10) && ubits(d->a[0]) <= 12 &&
            ubits(d->b[0]) <= 12 && d->K * (int64_t)(126 * 126) < (1ll << 31)) {
            cfg.variant = 3;   // three products on the lock-step Karatsuba kernel (same tiles and k-tiles as variant 10)
            for (QPackedGeom* g : {pa, pb}) {
                const qfmt f = g == pa ? d->a[0] : d->b[0];
                g->digit6 = 1;
                g->bias = f.S ? ((int64_t)1 << ((int)f.I + (int)f.F)) : 0;
                g->rowsum_off = g->trailer + QG_TRAILER_BYTES;
            }
            info->packed_bytes[0] += pa->rows_p * 8;
            info->packed_bytes[1] += pb->rows_p * 8;
        }
    }
    info->packed_bytes[2] = (int64_t)parts * pc->Mp * pc->Np * pc->cbytes;
    *pLA = LA;
    *pLB = LB;
    *pVar = cfg;
    if (pc_c) *pc_c = *pc;
    if (ep && an->band) {
        info->supported = 0;
        snprintf(info->reason, sizeof info->reason, "element-wise chain after a plan whose C can leave its format (multi-word comparison artefact of the reference)");
        return QG_EUNSUPPORTED;
    }
    if (ep) {
        // D replaces C as the stored result: same index space, D's container and host element
        if ((d->is_complex != 0) != (ev->im != nullptr)) {
            info->supported = 0;
            snprintf(info->reason, sizeof info->reason, d->is_complex ? "complex GEMM: the chain is a qgemul_epilogue_cplx" : "real GEMM: the chain is a qgemul_epilogue");
            return QG_EINVAL;
        }
        QEpTable local[2];
        QEpTable* t[2] = {ept ? ept : &local[0], ept_im ? ept_im : &local[1]};
        qfmt df[2] = {ep->d, ep->d};
        for (int part = 0; part < parts; ++part) {
            const qgemul_epilogue* e = part ? ev->im : ep;
            int ep_bits = 0;
            char why[96];
            const int st = qg_analyze_ep(d->c[part], e, t[part], &ep_bits, why, sizeof why);
            if (st != QG_OK) {
                info->supported = 0;
                snprintf(info->reason, sizeof info->reason, "%s", why);
                return st;
            }
            if (ep_bits > info->max_bits) info->max_bits = ep_bits;
            df[part] = e->d;
            if (!d->is_complex)
                for (uint32_t k = 0; k < e->n_stages; ++k)
                    if (e->stage[k].op == QG_EW_PASS) {
                        info->supported = 0;
                        snprintf(info->reason, sizeof info->reason, "QG_EW_PASS: complex chains only");
                        return QG_EINVAL;
                    }
        }
        if (d->is_complex) {
            // the two chains describe the same operators: same length; a tensor operand is one packed buffer in one container;
            // a real operand has no imaginary half to read
            if (ev->im->n_stages != ep->n_stages) {
                info->supported = 0;
                snprintf(info->reason, sizeof info->reason, "complex chain: the part chains differ in length");
                return QG_EINVAL;
            }
            for (int k = 0; k < t[0]->n; ++k) {
                QEpStage &a = t[0]->st[k], &b = t[1]->st[k];
                const bool ta = a.op != QG_EW_PASS && !a.scalar, tb = b.op != QG_EW_PASS && !b.scalar;
                if (ev->e_cplx[k] && ta != tb) {
                    info->supported = 0;
                    snprintf(info->reason, sizeof info->reason, "complex chain: a complex tensor operand feeds both parts");
                    return QG_EINVAL;
                }
                if (!ev->e_cplx[k] && ta && tb && !same_fmt(ep->stage[k].e, ev->im->stage[k].e)) {
                    info->supported = 0;
                    snprintf(info->reason, sizeof info->reason, "complex chain: a real tensor operand has one format");
                    return QG_EINVAL;
                }
                if (ta && tb) a.ebytes = b.ebytes = a.ebytes > b.ebytes ? a.ebytes : b.ebytes;
            }
            t[0]->dbytes = t[1]->dbytes = t[0]->dbytes > t[1]->dbytes ? t[0]->dbytes : t[1]->dbytes;
        }
        *hc = qg_host_elem(df, d->is_complex);
        pc->cbytes = t[0]->dbytes;
        pc->elem_bytes = hc->size;
        for (int part = 0; part < 2; ++part) { pc->off[part] = hc->off[part]; pc->sb[part] = hc->sb[part]; }
        info->host_elem_bytes[2] = hc->size;
        info->host_imag_off[2] = hc->off[1];
        info->packed_bytes[2] = (int64_t)parts * pc->Mp * pc->Np * pc->cbytes;
    }
    return QG_OK;
}

// Where the chain runs (measurements: profiles/r01v_eltwise.jsonl, chain = scale + bias into C's own type):
//   3x3-limb kernel, 4096^3: plain 0.458 ms, chain fused into the kernel's epilogue 0.473, chain as a pass 0.490
//   single-limb 256^2-tile kernel, 8192^2 x 4096: plain 0.280, fused 0.503, pass 0.436  (the fused epilogue spills:
//   128 accumulator registers stay live; and with one workgroup per CU the matrix cores idle meanwhile)
// so the default fuses on the limb kernel only, and only chains the planner has bounded by 32-bit arithmetic (a 64-bit
// chain inside the kernel was measured slower than the pass on every kernel).  Everything else runs as ONE linear,
// HBM-bound pass over the stored C (all stages and the final conversion in that pass, 5-6 TB/s).
// The single-limb MFMA kernels keep the dot product in int32 and run their epilogue in 32 bits.  An exact LEFT shift into a C
// with finer fracBits can leave 32 bits before the overflow handling sees the value (found by tests/extended_fuzz.py:
// int<10,-3> operands into Qu<5,7>, shift by 13).  Those descriptors store the raw dot products and convert them in the
// 64-bit linear pass instead, which keeps the hot kernels' epilogue as it is.
static bool wide_epilogue(const qgemul_plan* p)
{
    if (p->comp.on) return false;   // (the combine pass converts in 64-bit arithmetic anyway)
    const QStep& q = p->an.lin.to_c[0];
    // ... and a C format beyond 31 value bits does not fit the 32-bit epilogue's clamp bounds at all (second find of the
    // extended fuzz runs: int<7,-2> x int<7,-1> into Qu<24,9>)
    return p->info.kernel == QG_KERNEL_MFMA_I8 && !q.identity && (q.W > 30 || (q.d < 0 && p->an.dot_bits - q.d > 31));
}

// centred operands (QPackedGeom::offs): the limb kernels' epilogues take the centres back out before the one round + overflow
static void centre_args(const qgemul_plan* p, QMfmaArgs& a, const void* packedA, const void* packedB)
{
    if (!p->pa.offs || p->comp.on) return;
    a.rsA = (const int64_t*)((const char*)packedA + p->pa.rowsum_off);
    a.rsB = (const int64_t*)((const char*)packedB + p->pb.rowsum_off);
    a.biasA = p->pa.bias;
    a.biasB = p->pb.bias;
    a.corr = (int64_t)((uint64_t)p->desc.K * (uint64_t)p->pa.bias * (uint64_t)p->pb.bias);
}

#ifdef QG_DIAG
static uint32_t* g_diag_stamps = nullptr;   // device buffer for in-kernel clock stamps (diagnostic build only)
extern "C" void qgemul_diag_set_stamps(void* dev) { g_diag_stamps = (uint32_t*)dev; }
#endif

static bool fuses_epilogue(const qgemul_plan* p)
{
    if (wide_epilogue(p) || p->comp.on) return false;
    if (p->flags & QG_OPT_UNFUSED_EPILOGUE) return false;
    if (!p->ept.bits32) return false;
    if (p->info.kernel == QG_KERNEL_MFMA_I8_LIMB && p->LA == 3 && p->LB == 3) return true;
    return p->info.kernel == QG_KERNEL_MFMA_I8 && (p->flags & QG_OPT_FUSED_EPILOGUE);
}

extern "C" {

uint32_t qgemul_abi_version(void) { return QGEMUL_ABI_VERSION; }
int qgemul_last_hip_error(void) { return g_last_hip; }

const char* qgemul_strerror(int st)
{
    switch (st) {
    case QG_OK: return "ok";
    case QG_EINVAL: return "invalid descriptor or argument";
    case QG_EUNSUPPORTED: return "descriptor outside the engine's supported range";
    case QG_EHIP: return "HIP runtime error";
    case QG_ERCCL: return "RCCL error";
    case QG_ERANGE: return "input raw value outside its declared format";
    case QG_ENOGPU: return "no gfx950 device: the engine has no CPU fallback";
    default: return "unknown status";
    }
}

int qgemul_classify(const qgemul_desc* d, uint32_t opt_flags, qgemul_info* out) { return qgemul_classify_ep(d, nullptr, opt_flags, out); }

static int classify_view(const qgemul_desc* d, const EpView* ev, uint32_t opt_flags, qgemul_info* out)
{
    if (!d || !out) return QG_EINVAL;
    QAnalysis* an = new (std::nothrow) QAnalysis;
    if (!an) return QG_EINVAL;
    int LA, LB;
    QMfmaCfg variant;
    QPackedGeom pa, pb;
    QCGeom pc;
    QHostElem ha, hb, hc;
    int st = plan_geometry(d, opt_flags, an, out, &LA, &LB, &variant, &pa, &pb, &pc, &ha, &hb, &hc, ev);
    delete an;
    return st;
}

int qgemul_classify_ep(const qgemul_desc* d, const qgemul_epilogue* ep, uint32_t opt_flags, qgemul_info* out)
{
    const EpView v = {ep, nullptr, nullptr};
    return classify_view(d, ep ? &v : nullptr, opt_flags, out);
}

int qgemul_classify_epc(const qgemul_desc* d, const qgemul_epilogue_cplx* ep, uint32_t opt_flags, qgemul_info* out)
{
    if (!ep) return QG_EINVAL;
    const EpView v = {&ep->part[0], &ep->part[1], ep->e_complex};
    return classify_view(d, &v, opt_flags, out);
}

int qgemul_ctx_create(int device, qgemul_ctx** out)
{
    if (!out) return QG_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return QG_ENOGPU;
    if (device < 0) QG_HIP(hipGetDevice(&device));
    if (device >= n) return QG_EINVAL;
    DeviceScope scope(device);   // the caller's current device is restored on return
    QG_HIP(scope.err);
    hipDeviceProp_t prop;
    QG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return QG_ENOGPU; // kernels exist for gfx950 only
    qgemul_ctx* c = (qgemul_ctx*)calloc(1, sizeof *c);
    if (!c) return QG_EINVAL;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { free(c); return QG_EHIP; }
    if (hipMalloc((void**)&c->flag_dev, 64) != hipSuccess) { hipStreamDestroy(c->stream); free(c); return QG_EHIP; }
    hipMemsetAsync(c->flag_dev, 0, 64, c->stream);
    *out = c;
    return QG_OK;
}

void qgemul_ctx_destroy(qgemul_ctx* c)
{
    if (!c) return;
    DeviceScope scope(c->device);
    hipStreamSynchronize(c->stream);
    hipFree(c->flag_dev);
    hipStreamDestroy(c->stream);
    free(c);
}

int qgemul_ctx_sync(qgemul_ctx* c)
{
    if (!c) return QG_EINVAL;
    QG_ON_DEVICE(c);
    QG_HIP(hipStreamSynchronize(c->stream));
    return QG_OK;
}

void* qgemul_ctx_stream(qgemul_ctx* c) { return c ? (void*)c->stream : nullptr; }
int qgemul_ctx_device(const qgemul_ctx* c) { return c ? c->device : -1; }

int qgemul_dev_alloc(qgemul_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return QG_EINVAL;
    QG_ON_DEVICE(c);
    QG_HIP(hipMalloc(out, bytes ? bytes : 16));
    return QG_OK;
}
int qgemul_dev_free(qgemul_ctx* c, void* p)
{
    if (!c) return QG_EINVAL;
    QG_ON_DEVICE(c);
    QG_HIP(hipStreamSynchronize(c->stream));
    QG_HIP(hipFree(p));
    return QG_OK;
}
int qgemul_memcpy_h2d(qgemul_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return QG_EINVAL;
    if (!bytes) return QG_OK;
    QG_ON_DEVICE(c);
    QG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    QG_HIP(hipStreamSynchronize(c->stream));
    return QG_OK;
}
int qgemul_memcpy_d2h(qgemul_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return QG_EINVAL;
    if (!bytes) return QG_OK;
    QG_ON_DEVICE(c);
    QG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    QG_HIP(hipStreamSynchronize(c->stream));
    return QG_OK;
}

int qgemul_plan_create(qgemul_ctx* c, const qgemul_desc* d, uint32_t opt_flags, qgemul_plan** out)
{
    return qgemul_plan_create_ep(c, d, nullptr, opt_flags, out);
}

static int plan_create_view(qgemul_ctx* c, const qgemul_desc* d, const EpView* ev, uint32_t opt_flags, qgemul_plan** out);

int qgemul_plan_create_ep(qgemul_ctx* c, const qgemul_desc* d, const qgemul_epilogue* ep, uint32_t opt_flags, qgemul_plan** out)
{
    const EpView v = {ep, nullptr, nullptr};
    return plan_create_view(c, d, ep ? &v : nullptr, opt_flags, out);
}

int qgemul_plan_create_epc(qgemul_ctx* c, const qgemul_desc* d, const qgemul_epilogue_cplx* ep, uint32_t opt_flags, qgemul_plan** out)
{
    if (!ep) return QG_EINVAL;
    const EpView v = {&ep->part[0], &ep->part[1], ep->e_complex};
    return plan_create_view(c, d, &v, opt_flags, out);
}

static int plan_create_view(qgemul_ctx* c, const qgemul_desc* d, const EpView* ev, uint32_t opt_flags, qgemul_plan** out)
{
    if (!c || !d || !out) return QG_EINVAL;
    qgemul_plan* p = new (std::nothrow) qgemul_plan;
    if (!p) return QG_EINVAL;
    memset(p, 0, sizeof *p);
    p->ctx = c;
    p->desc = *d;
    p->flags = opt_flags;
    if (ev) {
        p->has_ep = 1;
        p->ep = *ev->re;
        if (ev->im) {
            p->ep_cplx = 1;
            p->ep_im = *ev->im;
            memcpy(p->e_cplx, ev->e_cplx, sizeof p->e_cplx);
        }
    }
    int st = plan_geometry(d, opt_flags, &p->an, &p->info, &p->LA, &p->LB, &p->cfg, &p->pa, &p->pb, &p->pc, &p->ha, &p->hb, &p->hc,
                           ev, &p->ept, &p->pc_c, &p->ept_im, &p->comp);
    if (st != QG_OK) { delete p; return st; }
    p->variant = p->cfg.variant;
    DeviceScope scope(c->device);
    if (scope.err != hipSuccess || hipMalloc((void**)&p->dev_table, sizeof(QTreeTable)) != hipSuccess) {
        delete p;
        return QG_EHIP;
    }
    if (hipMemcpyAsync(p->dev_table, &p->an.tree, sizeof(QTreeTable), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        hipFree(p->dev_table);
        delete p;
        return QG_EHIP;
    }
    if (p->info.kernel == QG_KERNEL_MFMA_CPLX) {
        const size_t ws = (size_t)(2 * p->pa.rows_p) * (size_t)(2 * p->pb.rows_p) * sizeof(int64_t);
        if (hipMalloc((void**)&p->workspace, ws) != hipSuccess) {
            hipFree(p->dev_table);
            delete p;
            return QG_EHIP;
        }
    }
    if (wide_epilogue(p) && hipMalloc((void**)&p->wide_ws, (size_t)(p->pc_c.Mp * p->pc_c.Np) * sizeof(int32_t)) != hipSuccess) {
        hipFree(p->dev_table);
        hipFree(p->workspace);
        delete p;
        return QG_EHIP;
    }
    if (p->comp.on) {
        const size_t n = (size_t)p->pa.rows_p * (size_t)p->pb.rows_p;
        const size_t sl = n * (size_t)(p->comp.ga * p->comp.gb) * (size_t)p->comp.slab_bytes;
        const size_t ac = p->comp.nc > 1 ? n * (p->comp.wide ? 16 : 8) : 0;
        if (hipMalloc(&p->comp_slabs, sl ? sl : 16) != hipSuccess || (ac && hipMalloc(&p->comp_acc, ac) != hipSuccess)) {
            hipFree(p->dev_table);
            hipFree(p->comp_slabs);
            delete p;
            return QG_EHIP;
        }
    }
    if (p->has_ep && !fuses_epilogue(p)) {
        // the tree kernels store C; the chain then runs as a pass over it
        const size_t cb = (size_t)(p->pc_c.parts * p->pc_c.Mp * p->pc_c.Np) * (size_t)p->pc_c.cbytes;
        if (hipMalloc(&p->cwork, cb ? cb : 16) != hipSuccess) {
            hipFree(p->dev_table);
            hipFree(p->workspace);
            hipFree(p->wide_ws);
            hipFree(p->comp_slabs);
            hipFree(p->comp_acc);
            delete p;
            return QG_EHIP;
        }
    }
    *out = p;
    return QG_OK;
}

void qgemul_plan_destroy(qgemul_plan* p)
{
    if (!p) return;
    DeviceScope scope(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->dev_table);
    hipFree(p->workspace);
    hipFree(p->cwork);
    hipFree(p->wide_ws);
    hipFree(p->hostc_pc);
    hipFree(p->comp_slabs);
    hipFree(p->comp_acc);
    delete p;
}

int qgemul_plan_info(const qgemul_plan* p, qgemul_info* out)
{
    if (!p || !out) return QG_EINVAL;
    *out = p->info;
    return QG_OK;
}

static QOperandGeom operand_geom(const qgemul_plan* p, int operand, int64_t ld)
{
    const qgemul_desc& d = p->desc;
    QOperandGeom g;
    memset(&g, 0, sizeof g);
    const QHostElem& h = operand == QG_OPERAND_A ? p->ha : p->hb;
    const qfmt* f = operand == QG_OPERAND_A ? d.a : d.b;
    g.K = d.K;
    g.parts = d.is_complex ? 2 : 1;
    g.elem_bytes = h.size;
    for (int q = 0; q < 2; ++q) {
        g.off[q] = h.off[q];
        g.sb[q] = h.sb[q];
        g.W[q] = (int)f[q].I + (int)f[q].F;
        g.S[q] = f[q].S;
        g.F[q] = f[q].F;
        g.Q[q] = f[q].Q;
        g.O[q] = f[q].O;
    }
    if (operand == QG_OPERAND_A) {
        g.rows = d.M;
        if (d.transA) { g.rs = ld ? ld : d.K; g.ks = 1; }   // A declared dim<K,M>: (i,k) at k + i*ld
        else { g.rs = 1; g.ks = ld ? ld : d.M; }            // A declared dim<M,K>: (i,k) at i + k*ld
    } else {
        g.rows = d.N;
        g.rs = ld ? ld : d.K;                                // B declared dim<K,N>: (k,j) at k + j*ld
        g.ks = 1;
    }
    return g;
}

int qgemul_pack(qgemul_plan* p, int operand, const void* src_dev, int64_t ld, void* packed_dev)
{
    if (!p || !src_dev || !packed_dev || (operand != QG_OPERAND_A && operand != QG_OPERAND_B)) return QG_EINVAL;
    QG_ON_DEVICE(p->ctx);
    QOperandGeom g = operand_geom(p, operand, ld);
    const QPackedGeom& pg = operand == QG_OPERAND_A ? p->pa : p->pb;
    const int check = (p->flags & QG_OPT_CHECK_RANGE) ? 1 : 0;
    if (check) QG_HIP(hipMemsetAsync(p->ctx->flag_dev, 0, 4, p->ctx->stream));
    if (p->comp.on) {
        QG_HIP(comp_zero_rowsums(p, operand, packed_dev));
        QG_HIP(comp_for_each_sub(p, operand, g, [&](const QOperandGeom& sg, const QPackedGeom& sp, int64_t off) {
            return qg_launch_pack(sg, sp, src_dev, (char*)packed_dev + off, check, p->ctx->flag_dev, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0);
        }));
    } else
    QG_HIP(qg_launch_pack(g, pg, src_dev, packed_dev, check, p->ctx->flag_dev, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0));
    if (check) {
        int flag = 0;
        QG_HIP(hipMemcpyAsync(&flag, p->ctx->flag_dev, 4, hipMemcpyDeviceToHost, p->ctx->stream));
        QG_HIP(hipStreamSynchronize(p->ctx->stream));
        if (flag) return QG_ERANGE;
    }
    return QG_OK;
}

int qgemul_pack_f64(qgemul_plan* p, int operand, const double* src_dev, int64_t ld, void* packed_dev)
{
    if (!p || !src_dev || !packed_dev || (operand != QG_OPERAND_A && operand != QG_OPERAND_B)) return QG_EINVAL;
    if (!(p->flags & QG_OPT_ARITHMETIC_CONV)) {
        // Qu_s(double) with QuMode<RND::CONV>: the reference's result is an artefact of its multi-word CONV branch
        // (QuBLAS.h:2137-2156 on ArbiInt<2400>); silently returning the arithmetic answer would break bit-exactness
        const qfmt* f = operand == QG_OPERAND_A ? p->desc.a : p->desc.b;
        for (int q = 0; q < (p->desc.is_complex ? 2 : 1); ++q)
            if (f[q].Q == QG_RND_CONV) return QG_EUNSUPPORTED;
    }
    QG_ON_DEVICE(p->ctx);
    QOperandGeom g = operand_geom(p, operand, ld);
    const QPackedGeom& pg = operand == QG_OPERAND_A ? p->pa : p->pb;
    if (p->comp.on) {
        QG_HIP(comp_zero_rowsums(p, operand, packed_dev));
        QG_HIP(comp_for_each_sub(p, operand, g, [&](const QOperandGeom& sg, const QPackedGeom& sp, int64_t off) {
            return qg_launch_pack_f64(sg, sp, src_dev, (char*)packed_dev + off, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0);
        }));
        return QG_OK;
    }
    QG_HIP(qg_launch_pack_f64(g, pg, src_dev, packed_dev, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0));
    return QG_OK;
}

int qgemul_fill_packed(qgemul_plan* p, int operand, uint64_t seed, int dist, void* packed_dev)
{
    if (!p || !packed_dev || (operand != QG_OPERAND_A && operand != QG_OPERAND_B)) return QG_EINVAL;
    QG_ON_DEVICE(p->ctx);
    QOperandGeom g = operand_geom(p, operand, 0);
    const QPackedGeom& pg = operand == QG_OPERAND_A ? p->pa : p->pb;
    if (p->comp.on) {
        QG_HIP(comp_zero_rowsums(p, operand, packed_dev));
        QG_HIP(comp_for_each_sub(p, operand, g, [&](const QOperandGeom& sg, const QPackedGeom& sp, int64_t off) {
            return qg_launch_fill(sg, sp, seed, dist, (char*)packed_dev + off, p->ctx->stream);
        }));
        return QG_OK;
    }
    QG_HIP(qg_launch_fill(g, pg, seed, dist, packed_dev, p->ctx->stream));
    return QG_OK;
}

int qgemul_unpack_c(qgemul_plan* p, const void* packed_dev, void* dst_dev, int64_t ld)
{
    if (!p || !packed_dev || !dst_dev) return QG_EINVAL;
    QG_ON_DEVICE(p->ctx);
    QCGeom c = p->pc;
    c.ldc = ld ? ld : p->desc.M;
    QG_HIP(qg_launch_unpack_c(c, packed_dev, dst_dev, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0));
    return QG_OK;
}

static int execute_kernel(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB, const qgemul_ep_args* epa, const HostC* hostc = nullptr);

int qgemul_execute(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB)
{
    if (!p || !packedC || !packedA || !packedB) return QG_EINVAL;
    if (p->has_ep) return QG_EINVAL;  // a plan with an epilogue runs through qgemul_execute_ep
    if (p->desc.M == 0 || p->desc.N == 0) return QG_OK;
    QG_ON_DEVICE(p->ctx);
    return execute_kernel(p, packedC, packedA, packedB, nullptr);
}

// the kernels whose epilogue can store the reference layout: k_mfma_pp (variant 9) / k_mfma_ppl (variant 10), real, 4- or 8-byte
// container equal to the host element, no raw-dot-product detour
static bool stores_host_c(const qgemul_plan* p)
{
    if (p->has_ep || p->desc.is_complex || wide_epilogue(p) || p->comp.on) return false;
    if (p->info.kernel != QG_KERNEL_MFMA_I8 && p->info.kernel != QG_KERNEL_MFMA_I8_LIMB) return false;
    if (p->variant != 9 && p->variant != 10) return false;
    if (p->variant == 10 && (p->pa.rows_p / p->cfg.TM) * (p->pb.rows_p / p->cfg.TN) < 256) return false;   // (falls back to the lock-step kernel)
    return (p->pc.cbytes == 4 || p->pc.cbytes == 8) && p->pc.cbytes == p->hc.size;
}

int qgemul_plan_stores_host_c(const qgemul_plan* p) { return p && stores_host_c(p) ? 1 : 0; }

int qgemul_execute_host_c(qgemul_plan* p, void* C_dev, int64_t ldc, const void* packedA, const void* packedB)
{
    if (!p || !C_dev || !packedA || !packedB || p->has_ep) return QG_EINVAL;
    if (ldc && ldc < p->desc.M) return QG_EINVAL;
    if (p->desc.M == 0 || p->desc.N == 0) return QG_OK;
    QG_ON_DEVICE(p->ctx);
    if (stores_host_c(p)) {
        const HostC h{C_dev, ldc ? ldc : p->desc.M};
        return execute_kernel(p, C_dev, packedA, packedB, nullptr, &h);
    }
    if (!p->hostc_pc) QG_HIP(hipMalloc(&p->hostc_pc, (size_t)p->info.packed_bytes[2] ? (size_t)p->info.packed_bytes[2] : 16));
    const int st = execute_kernel(p, p->hostc_pc, packedA, packedB, nullptr);
    if (st != QG_OK) return st;
    QCGeom c = p->pc;
    c.ldc = ldc ? ldc : p->desc.M;
    QG_HIP(qg_launch_unpack_c(c, p->hostc_pc, C_dev, p->ctx->stream, (p->flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0));
    return QG_OK;
}

int qgemul_execute_ep(qgemul_plan* p, void* packedD, const void* packedA, const void* packedB, const qgemul_ep_args* args)
{
    if (!p || !packedD || !packedA || !packedB) return QG_EINVAL;
    if (!p->has_ep) return args ? QG_EINVAL : qgemul_execute(p, packedD, packedA, packedB);
    if (!args && p->ept.n > 0) return QG_EINVAL;
    for (int k = 0; k < p->ept.n; ++k)
        if ((!p->ept.st[k].scalar || (p->ep_cplx && !p->ept_im.st[k].scalar)) && !args->e_packed[k]) return QG_EINVAL;
    if (p->desc.M == 0 || p->desc.N == 0) return QG_OK;
    QG_ON_DEVICE(p->ctx);
    QEpArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < p->ept.n; ++k) {
        a.e[k] = (const char*)args->e_packed[k];
        a.scalar[k] = args->e_scalar[k];
    }
    hipStream_t st = p->ctx->stream;
    if (fuses_epilogue(p)) {
        // fused: the MFMA kernel's epilogue runs the chain on the value it has just converted into C's type
        QMfmaArgs m;
        memset(&m, 0, sizeof m);
        m.A = (const int8_t*)packedA;
        m.B = (const int8_t*)packedB;
        m.C = packedD;
        m.Mp = p->pa.rows_p;
        m.Np = p->pb.rows_p;
        m.Kp = p->pa.K_p;
        m.cbytes = p->pc_c.cbytes;
        m.variant = p->variant;
        m.to_c = p->an.lin.to_c[0];
        m.maskA = p->pa.trailer ? (const uint32_t*)((const char*)packedA + p->pa.trailer) : nullptr;
        m.maskB = p->pb.trailer ? (const uint32_t*)((const char*)packedB + p->pb.trailer) : nullptr;
        m.has_ep = 1;
        m.ep = p->ept;
        m.epa = a;
        centre_args(p, m, packedA, packedB);
        QG_HIP(qg_launch_mfma(p->LA, p->LB, m, st));
        return QG_OK;
    }
    // not fused: the kernel stores C into the plan's buffer, one linear pass turns it into D
    const int s = execute_kernel(p, p->cwork, packedA, packedB, nullptr);
    if (s != QG_OK) return s;
    QEltwiseArgs g;
    memset(&g, 0, sizeof g);
    g.C = (const char*)p->cwork;
    g.D = (char*)packedD;
    g.n = p->pc_c.Mp * p->pc_c.Np;
    g.cbytes = p->pc_c.cbytes;
    g.t = p->ept;
    g.a = a;
    QG_HIP(qg_launch_eltwise(g, st));
    if (p->ep_cplx) {
        // the chain of the imaginary parts: the second half of packed C, of packed D and of every complex tensor operand
        g.C += g.n * g.cbytes;
        g.D += g.n * p->ept.dbytes;
        g.t = p->ept_im;
        for (int k = 0; k < p->ept_im.n; ++k) {
            if (g.a.e[k] && p->e_cplx[k]) g.a.e[k] += g.n * p->ept_im.st[k].ebytes;
            g.a.scalar[k] = args->e_scalar_im[k];
        }
        QG_HIP(qg_launch_eltwise(g, st));
    }
    return QG_OK;
}

int qgemul_plan_fuses_epilogue(const qgemul_plan* p) { return p && p->has_ep && fuses_epilogue(p) ? 1 : 0; }

int qgemul_plan_packed_layout(const qgemul_plan* p, int operand, int64_t out[4])
{
    if (!p || !out || (operand != QG_OPERAND_A && operand != QG_OPERAND_B)) return QG_EINVAL;
    const QPackedGeom& g = operand == QG_OPERAND_A ? p->pa : p->pb;
    out[0] = p->comp.on ? 0 : g.trailer;
    out[1] = g.offs ? g.rowsum_off : 0;
    out[2] = g.rows_p;
    out[3] = g.offs ? -g.bias : 0;
    return QG_OK;
}

// the stage entry that reads stage k's tensor operand (nullptr: the stage has no tensor operand)
static const QEpStage* stage_tensor(const qgemul_plan* p, int k)
{
    if (!p->ept.st[k].scalar) return &p->ept.st[k];
    if (p->ep_cplx && !p->ept_im.st[k].scalar) return &p->ept_im.st[k];
    return nullptr;
}

int64_t qgemul_packed_e_bytes(const qgemul_plan* p, int stage)
{
    if (!p || !p->has_ep || stage < 0 || stage >= p->ept.n) return 0;
    const QEpStage* t = stage_tensor(p, stage);
    if (!t) return 0;
    return (p->ep_cplx && p->e_cplx[stage] ? 2 : 1) * p->pc.Mp * p->pc.Np * (int64_t)t->ebytes;
}

int qgemul_pack_e(qgemul_plan* p, int stage, const void* src_dev, int64_t ld, void* packed_dev)
{
    if (!p || !src_dev || !packed_dev || !p->has_ep || stage < 0 || stage >= p->ept.n) return QG_EINVAL;
    const QEpStage* t = stage_tensor(p, stage);
    if (!t) return QG_EINVAL;
    if (ld && ld < p->desc.M) return QG_EINVAL;
    QG_ON_DEVICE(p->ctx);
    const bool cplx = p->ep_cplx && p->e_cplx[stage];
    // host element of the operand tensor: int32 / int64 raw values, {re, im} structs for a complex operand (QuBLAS.h:2512-2513)
    const qfmt f[2] = {t == &p->ept.st[stage] ? p->ep.stage[stage].e : p->ep_im.stage[stage].e, p->ep_im.stage[stage].e};
    const QHostElem h = qg_host_elem(f, cplx ? 1 : 0);
    for (int part = 0; part < (cplx ? 2 : 1); ++part)
        QG_HIP(qg_launch_pack_e(p->pc, part, src_dev, ld ? ld : p->desc.M, h.size, h.off[part], h.sb[part], packed_dev, t->ebytes, p->ctx->stream));
    return QG_OK;
}

// composite linear plan: per k-chunk, one MFMA launch per (A group, B group) storing raw dot products into its slab, then the
// exact combine pass (running sums between chunks; one round + overflow into C after the last)
static int execute_composite(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB, const QCGeom& pcg)
{
    const QComposite& q = p->comp;
    hipStream_t st = p->ctx->stream;
    const int64_t n = p->pa.rows_p * p->pb.rows_p;
    for (int c = 0; c < q.nc; ++c) {
        QLinCombine cb;
        memset(&cb, 0, sizeof cb);
        for (int i = 0; i < q.ga; ++i)
            for (int j = 0; j < q.gb; ++j) {
                int64_t offA = 0, offB = 0;
                const QPackedGeom sa = comp_sub_geom(q, p->pa, 0, p->desc.K, c, i, &offA);
                const QPackedGeom sb = comp_sub_geom(q, p->pb, 1, p->desc.K, c, j, &offB);
                char* slab = (char*)p->comp_slabs + (size_t)(i * q.gb + j) * (size_t)n * (size_t)q.slab_bytes;
                QMfmaArgs a;
                memset(&a, 0, sizeof a);
                a.A = (const int8_t*)packedA + offA;
                a.B = (const int8_t*)packedB + offB;
                a.C = slab;
                a.Mp = sa.rows_p;
                a.Np = sb.rows_p;
                a.Kp = sa.K_p;
                a.cbytes = q.slab_bytes;
                a.variant = q.var[i][j];
                a.to_c.identity = 1;   // raw dot products
                a.maskA = sa.trailer ? (const uint32_t*)((const char*)a.A + sa.trailer) : nullptr;
                a.maskB = sb.trailer ? (const uint32_t*)((const char*)a.B + sb.trailer) : nullptr;
                QG_HIP(qg_launch_mfma(sa.limbs, sb.limbs, a, st));
                cb.slab[cb.n_slabs] = slab;
                cb.sh[cb.n_slabs] = 8 * (sa.limb0 + sb.limb0);
                ++cb.n_slabs;
            }
        cb.slab_bytes = q.slab_bytes;
        cb.n = n;
        cb.acc_in = c > 0 ? p->comp_acc : nullptr;
        cb.acc_out = c + 1 < q.nc ? p->comp_acc : nullptr;
        cb.out = packedC;
        cb.cbytes = pcg.cbytes;
        cb.wide = q.wide;
        cb.to_c = p->an.lin.to_c[0];
        if (p->pa.offs) {   // centred operands: the last chunk's pass takes the centres back out
            cb.rsA = (const int64_t*)((const char*)packedA + p->pa.rowsum_off);
            cb.rsB = (const int64_t*)((const char*)packedB + p->pb.rowsum_off);
            cb.biasA = p->pa.bias;
            cb.biasB = p->pb.bias;
            cb.corr = p->desc.K;
            cb.tm = pcg.tm;
            cb.tn = pcg.tn;
            cb.tiles_n = pcg.tn ? pcg.Np / pcg.tn : 1;
        }
        QG_HIP(qg_launch_lin_combine(cb, st));
    }
    return QG_OK;
}

static int execute_kernel(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB, const qgemul_ep_args*, const HostC* hostc)
{
    hipStream_t st = p->ctx->stream;
    const QCGeom& pcg = p->has_ep ? p->pc_c : p->pc;
    if (p->comp.on) return execute_composite(p, packedC, packedA, packedB, pcg);
    switch (p->info.kernel) {
    case QG_KERNEL_MFMA_I8:
    case QG_KERNEL_MFMA_I8_LIMB: {
        QMfmaArgs a;
        memset(&a, 0, sizeof a);
        a.A = (const int8_t*)packedA;
        a.B = (const int8_t*)packedB;
        a.C = packedC;
        a.Mp = p->pa.rows_p;
        a.Np = p->pb.rows_p;
        a.Kp = p->pa.K_p;
        a.cbytes = pcg.cbytes;
        a.variant = p->variant;
        a.to_c = p->an.lin.to_c[0];
        a.maskA = p->pa.trailer ? (const uint32_t*)((const char*)packedA + p->pa.trailer) : nullptr;
        a.maskB = p->pb.trailer ? (const uint32_t*)((const char*)packedB + p->pb.trailer) : nullptr;
#ifdef QG_DIAG
        a.dbg = g_diag_stamps;
#endif
        centre_args(p, a, packedA, packedB);
        if (p->pa.digit6) {
            a.kara = 1;
            a.rsA = (const int64_t*)((const char*)packedA + p->pa.rowsum_off);
            a.rsB = (const int64_t*)((const char*)packedB + p->pb.rowsum_off);
            a.biasA = p->pa.bias;
            a.biasB = p->pb.bias;
            a.corr = p->desc.K * p->pa.bias * p->pb.bias;
        }
        if (wide_epilogue(p)) {
            a.C = p->wide_ws;
            a.cbytes = 4;
            memset(&a.to_c, 0, sizeof a.to_c);
            a.to_c.identity = 1;                    // raw int32 dot products ...
            QG_HIP(qg_launch_mfma(p->LA, p->LB, a, st));
            QEltwiseArgs f;
            memset(&f, 0, sizeof f);
            f.C = (const char*)p->wide_ws;
            f.D = (char*)packedC;
            f.n = pcg.Mp * pcg.Np;
            f.cbytes = 4;
            f.t.dbytes = pcg.cbytes;
            f.t.to_d = p->an.lin.to_c[0];           // ... shifted, overflow-handled and stored in 64-bit arithmetic
            QG_HIP(qg_launch_eltwise(f, st));
            return QG_OK;
        }
        if (hostc) {   // (only reached when stores_host_c(p): see qgemul_execute_host_c)
            a.C = hostc->C;
            a.c_host = 1;
            a.c_ld = hostc->ld;
            a.c_M = p->desc.M;
            a.c_N = p->desc.N;
            a.c_vec = (((uintptr_t)hostc->C & 15) == 0 && (hostc->ld * pcg.cbytes) % 16 == 0) ? 1 : 0;
        }
        QG_HIP(qg_launch_mfma(p->LA, p->LB, a, st));
        return QG_OK;
    }
    case QG_KERNEL_MFMA_CPLX: {
        QMfmaArgs a;
        memset(&a, 0, sizeof a);
        a.A = (const int8_t*)packedA;
        a.B = (const int8_t*)packedB;
        a.C = p->workspace;
        a.Mp = 2 * p->pa.rows_p;
        a.Np = 2 * p->pb.rows_p;
        a.Kp = p->pa.K_p;
        a.cbytes = 8;
        a.variant = p->variant;
        a.maskA = p->pa.trailer ? (const uint32_t*)((const char*)packedA + p->pa.trailer) : nullptr;
        a.maskB = p->pb.trailer ? (const uint32_t*)((const char*)packedB + p->pb.trailer) : nullptr;
        memset(&a.to_c, 0, sizeof a.to_c);
        a.to_c.identity = 1;  // raw 64-bit dot products
        QG_HIP(qg_launch_mfma(p->LA, p->LB, a, st));
        QCplxCombine g;
        g.D = p->workspace;
        g.C = (char*)packedC;
        g.M = p->desc.M;
        g.N = p->desc.N;
        g.Mh = p->pa.rows_p;
        g.Nh = p->pb.rows_p;
        g.Np = 2 * p->pb.rows_p;
        g.tm = p->cfg.TM;
        g.tn = p->cfg.TN;
        g.cbytes = pcg.cbytes;
        for (int i = 0; i < 4; ++i) g.sh[i] = p->an.lin.sh[i];
        g.to_c[0] = p->an.lin.to_c[0];
        g.to_c[1] = p->an.lin.to_c[1];
        QG_HIP(qg_launch_cplx_combine(g, st));
        return QG_OK;
    }
    case QG_KERNEL_TREE_I32: {
        static const bool no_lj = QG_DIAG_ENV("QG_NO_LEFT_JUSTIFIED");   // A/B switch (diagnostic library): the form such a descriptor had before
        static const bool no_pk = QG_DIAG_ENV("QG_NO_PACKED16");
        int fm = (p->an.fast_mode >= 6 && p->an.fast_mode <= 9 && no_lj) ? p->an.fast_mode_base : (p->an.fast_mode >= 7 && p->an.fast_mode <= 9 && no_pk) ? 6 : p->an.fast_mode;
        if (fm == 10) {   // 32-bit words: product shift 10 ... 23 -> k_tree_fast<., 18>; justified words (lj.e[0] > 0) -> <., 19 / 20>
            const bool mad = p->an.tree.lj.s >= 10 && p->an.tree.lj.s <= 23;
            fm = p->an.tree.lj.e[1] ? 14 : p->an.tree.lj.e[0] > 0 ? (mad ? 13 : 12) : (mad ? 11 : 10);   // (14: a wrapping word, k_tree_fast<., 21>)
        }
        if (fm >= 6 && fm <= 9 && p->an.lj_unsigned) fm += 16;   // (the unsigned counterparts: qg_launch_tree_fast)
        QG_HIP(qg_launch_tree_fast(p->dev_table, p->an.tree.n_levels_k, p->an.split_s, p->an.mul24_ok,
                                   (p->flags & QG_OPT_RUNTIME_MODES) ? 0 : fm, packedA, packedB, packedC,
                                   p->desc.M, p->desc.N, p->pa.K_p, pcg.cbytes, st));
        return QG_OK;
    }
    case QG_KERNEL_GEMV_I64:
        QG_HIP(qg_launch_gemv(p->dev_table, p->an.tree.n_levels_k, p->an.gemv_b_bit, 0, packedA, packedB, packedC, p->desc.M, p->pa.K_p,
                              pcg.cbytes, st, 1));
        return QG_OK;
    case QG_KERNEL_GEMV_I32:
        QG_HIP(qg_launch_gemv(p->dev_table, p->an.tree.n_levels_k, p->an.gemv_b_bit, (p->flags & QG_OPT_RUNTIME_MODES) ? 0 : p->an.gemv_fixed,
                              packedA, packedB, packedC, p->desc.M, p->pa.K_p,
                              pcg.cbytes, st));
        return QG_OK;
    case QG_KERNEL_TREE_CPLX_I32:
        // (the justified forms carry, in the second byte, the form the descriptor has without them: the diagnostic library's A/B switches)
        QG_HIP(qg_launch_tree_cplx_fast(p->dev_table, p->an.tree.n_levels_k,
                                        (p->flags & QG_OPT_RUNTIME_MODES) ? 0 : p->an.cplx_fixed_ok >= 5 ? (p->an.cplx_fixed_ok | p->an.cplx_fixed_base << 8) : p->an.cplx_fixed_ok,
                                        p->desc.cmul == QG_CMUL_TF ? 1 : 0, packedA, packedB, packedC, p->desc.M, p->desc.N,
                                        p->pa.K_p, pcg.cbytes, st));
        return QG_OK;
    case QG_KERNEL_TREE_I64:
        if (p->an.tree64_ok && !(p->flags & QG_OPT_GENERIC_TREE)) {
            QG_HIP(qg_launch_tree64(p->dev_table, p->an.tree.n_levels_k, packedA, packedB, packedC, p->desc.M, p->desc.N, p->pa.K_p,
                                    p->pa.cbytes, p->pb.cbytes, pcg.cbytes, st));
            return QG_OK;
        }
        [[fallthrough]];
    case QG_KERNEL_TREE_CPLX:
        QG_HIP(qg_launch_tree_generic(p->dev_table, p->desc.is_complex ? 2 : 1, packedA, packedB, packedC, p->desc.M, p->desc.N,
                                      p->desc.K, p->pa, p->pb, pcg, st));
        return QG_OK;
    case QG_KERNEL_TREE_I128:
        QG_HIP(qg_launch_tree_generic(p->dev_table, p->desc.is_complex ? 2 : 1, packedA, packedB, packedC, p->desc.M, p->desc.N,
                                      p->desc.K, p->pa, p->pb, pcg, st, 1));
        return QG_OK;
    default:
        return QG_EUNSUPPORTED;
    }
}

static int result_width(const qgemul_plan* p, int part = 0)
{
    const qfmt f = p->has_ep ? (part ? p->ep_im.d : p->ep.d) : p->desc.c[part];
    return (int)f.I + (int)f.F + (f.S ? 1 : 0);
}

int64_t qgemul_bitstream_bytes(const qgemul_plan* p, int format)
{
    if (!p) return 0;
    const int64_t n = p->desc.M * p->desc.N;
    // complex: "(re-bits, im-bits)" per element as characters; packed: the binary characters only
    const int64_t bits = p->desc.is_complex ? n * (result_width(p, 0) + result_width(p, 1) + (format == QG_BITS_PACKED ? 0 : 4)) : n * (int64_t)result_width(p);
    // the packed form is written with 32-bit atomics: sized to whole words
    return format == QG_BITS_PACKED ? ((bits + 7) / 8 + 3) / 4 * 4 : bits;
}

int qgemul_export_bitstream(qgemul_plan* p, const void* packedC, int tensor_chunk, int elem_chunk, int format, void* out_dev)
{
    if (!p || !packedC || !out_dev || (format != QG_BITS_ASCII && format != QG_BITS_PACKED)) return QG_EINVAL;
    const int w = p->desc.is_complex ? result_width(p, 0) + result_width(p, 1) + 4 : result_width(p);
    const int64_t n = p->desc.M * p->desc.N;
    if (w <= 0 || tensor_chunk < 0 || elem_chunk < 0) return QG_EINVAL;
    if (!p->desc.is_complex && (w > 64 || p->pc.cbytes > 8)) return QG_EUNSUPPORTED;   // (multi-word elements: not exported)
    if (elem_chunk > 0 && w % elem_chunk) return QG_EINVAL;      // the reference throws (QuBLAS.h:4599-4602)
    if (tensor_chunk > 0 && n % tensor_chunk) return QG_EINVAL;  // the reference's loop does not terminate (:4745)
    QG_ON_DEVICE(p->ctx);
    if (p->desc.is_complex) {
        const int wr = result_width(p, 0), wi = result_width(p, 1);
        if (wr < 0 || wi < 0 || wr > 64 || wi > 64) return QG_EINVAL;   // (a part without bits prints as the empty string)
        QBitsCplxArgs a;
        memset(&a, 0, sizeof a);
        a.c = p->pc;
        a.packed = (const char*)packedC;
        a.out = (char*)out_dev;
        a.width = w;
        a.nbits = wr + wi;
        a.tensor_chunk = tensor_chunk;
        a.packed_bits = format == QG_BITS_PACKED;
        // the element string "(" re ", " im ")" (QuBLAS.h:2553-2556), MSB first per part, then its chunks reversed (:4593-4611)
        uint8_t str[136];
        int k = 0;
        str[k++] = QG_BITS_LIT_OPEN;
        for (int j = 0; j < wr; ++j) str[k++] = (uint8_t)(wr - 1 - j);
        str[k++] = QG_BITS_LIT_COMMA;
        str[k++] = QG_BITS_LIT_SPACE;
        for (int j = 0; j < wi; ++j) str[k++] = (uint8_t)(64 + wi - 1 - j);
        str[k++] = QG_BITS_LIT_CLOSE;
        if (elem_chunk > 0) {
            const int nch = w / elem_chunk;
            for (int q = 0; q < nch; ++q)
                for (int r = 0; r < elem_chunk; ++r) a.tab[q * elem_chunk + r] = str[(nch - 1 - q) * elem_chunk + r];
        } else {
            memcpy(a.tab, str, (size_t)w);
        }
        QG_HIP(qg_launch_bitstream_cplx(a, p->ctx->stream));
        return QG_OK;
    }
    QBitsArgs a;
    memset(&a, 0, sizeof a);
    a.c = p->pc;
    a.packed = (const char*)packedC;
    a.out = (char*)out_dev;
    a.width = w;
    a.tensor_chunk = tensor_chunk;
    a.elem_chunk = elem_chunk;
    a.packed_bits = format == QG_BITS_PACKED;
    QG_HIP(qg_launch_bitstream(a, p->ctx->stream));
    return QG_OK;
}

static int time_execute(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB, const qgemul_ep_args* args,
                        int warmup, int iters, float* avg_ms)
{
    if (!p || !avg_ms || iters < 1) return QG_EINVAL;
    QG_ON_DEVICE(p->ctx);
    hipStream_t st = p->ctx->stream;
    auto once = [&]() { return p->has_ep ? qgemul_execute_ep(p, packedC, packedA, packedB, args) : qgemul_execute(p, packedC, packedA, packedB); };
    for (int i = 0; i < warmup; ++i) {
        int s = once();
        if (s) return s;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = QG_OK;
    float ms = 0;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, st);
    for (int i = 0; he == hipSuccess && rc == QG_OK && i < iters; ++i) rc = once();
    if (he == hipSuccess && rc == QG_OK) he = hipEventRecord(e1, st);
    if (he == hipSuccess && rc == QG_OK) he = hipEventSynchronize(e1);
    if (he == hipSuccess && rc == QG_OK) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) hipEventDestroy(e0);   // (released on every path)
    if (e1) hipEventDestroy(e1);
    if (he != hipSuccess) { g_last_hip = (int)he; return QG_EHIP; }
    if (rc != QG_OK) return rc;
    *avg_ms = ms / (float)iters;
    return QG_OK;
}

int qgemul_time_execute(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB, int warmup, int iters,
                        float* avg_ms)
{
    return time_execute(p, packedC, packedA, packedB, nullptr, warmup, iters, avg_ms);
}

int qgemul_time_execute_ep(qgemul_plan* p, void* packedD, const void* packedA, const void* packedB, const qgemul_ep_args* args,
                           int warmup, int iters, float* avg_ms)
{
    return time_execute(p, packedD, packedA, packedB, args, warmup, iters, avg_ms);
}

int qgemul_run(const qgemul_desc* d, void* C, const void* A, const void* B, const qgemul_opts* o)
{
    return qgemul_run_ep(d, nullptr, C, A, B, nullptr, o);
}

// ---- the one-shot call keeps a per-thread cache: context, the plan of the last descriptor, grow-only device buffers ----
// The reference is a synchronous library that user code calls in loops; creating a stream, a plan table and eight device
// allocations per call cost 2.9 ms for the README's 4x4x4 example (tools/measure_run_latency.py).  The cache belongs to
// the calling thread and is never touched by another one; qgemul_run_release() frees it, and so does the end of the thread
// (RunCacheReaper below) — except once the process is exiting: a worker thread that is still alive or detached then may run its
// thread-local destructors after HIP has been torn down, so the reaper only forgets its pointers (g_shutting_down).
namespace {
struct RunCache {
    qgemul_ctx* ctx = nullptr;
    int device = -2;
    qgemul_plan* plan = nullptr;
    qgemul_desc pd;
    qgemul_epilogue_cplx pe;          // a real chain is part[0]
    bool has_pe = false, pe_cplx = false;
    uint32_t pflags = 0;
    enum { NBUF = 6 + 16 };   // 0-5: host-layout A, B, C and packed A, B, C; behind them: the epilogue operands (2 per stage) or, on the
                              // root of a sharded call, the landing buffers of the other bands
    void* buf[NBUF] = {};
    size_t cap[NBUF] = {};
};
thread_local RunCache g_run;
// qgemul_run_sharded: one such cache per entry of the device list (slot i serves devices[i] of the calling thread's last list)
enum { QG_MAX_SHARDS = 16 };
thread_local RunCache g_shard[QG_MAX_SHARDS];
thread_local hipEvent_t g_shard_ev[QG_MAX_SHARDS] = {};
// What a thread has cached is released when the thread ends (worker threads that call Qgemul<>() and exit must not leak a stream
// and device buffers each).  For the main thread this runs inside exit() BEFORE any static object — HIP's included — is torn down.
std::atomic<bool> g_shutting_down{false};
void forget_caches();
struct RunCacheReaper {
    bool armed = false;
    ~RunCacheReaper()
    {
        if (!armed) return;
        int n = 0;
        // exit() has begun (atexit handlers run before static destruction, HIP's included) or the runtime no longer answers:
        // no HIP call from here, the driver reclaims the memory with the process
        if (g_shutting_down.load(std::memory_order_acquire) || hipGetDeviceCount(&n) != hipSuccess) { forget_caches(); return; }
        qgemul_run_release();
    }
};
thread_local RunCacheReaper g_reaper;
struct ShutdownHook {
    ShutdownHook() { atexit([] { g_shutting_down.store(true, std::memory_order_release); }); }
};

// descriptors are compared field by field: padding and reserved bytes of a caller's struct are not part of its meaning, and a
// descriptor that was not built with `{}` must still hit the cache
bool same_desc(const qgemul_desc& x, const qgemul_desc& y)
{
    if (x.abi != y.abi || x.transA != y.transA || x.is_complex != y.is_complex || x.cmul != y.cmul || x.flags != y.flags || x.M != y.M || x.N != y.N ||
        x.K != y.K || x.n_levels != y.n_levels || x.n_levels > QG_MAX_LEVELS)
        return false;
    for (int p = 0; p < 2; ++p) {
        if (!same_fmt(x.a[p], y.a[p]) || !same_fmt(x.b[p], y.b[p]) || !same_fmt(x.c[p], y.c[p])) return false;
        for (uint32_t l = 0; l < x.n_levels; ++l)
            if (!same_fmt(x.level_add[p][l], y.level_add[p][l]) || !same_fmt(x.level[p][l], y.level[p][l])) return false;
    }
    for (int i = 0; i < 8; ++i)
        if (!same_fmt(x.mul[i], y.mul[i])) return false;
    return true;
}
bool same_epilogue(const qgemul_epilogue& x, const qgemul_epilogue& y)
{
    if (x.n_stages != y.n_stages || x.n_stages > QG_MAX_EW || !same_fmt(x.d, y.d)) return false;
    for (uint32_t k = 0; k < x.n_stages; ++k) {
        const qgemul_ew_stage &a = x.stage[k], &b = y.stage[k];
        if (a.op != b.op || a.x_first != b.x_first || a.e_scalar != b.e_scalar || !same_fmt(a.e, b.e) || !same_fmt(a.r, b.r) || !same_fmt(a.t, b.t))
            return false;
    }
    return true;
}

int cache_buffer(RunCache& c, int i, size_t bytes, void** out)   // (the caller has made the cache's device current)
{
    if (bytes > c.cap[i]) {
        if (c.buf[i]) {
            hipStreamSynchronize(c.ctx->stream);
            hipFree(c.buf[i]);
            c.buf[i] = nullptr;
            c.cap[i] = 0;
        }
        const size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipMalloc(&c.buf[i], want);
        if (e != hipSuccess) { g_last_hip = (int)e; return QG_EHIP; }
        c.cap[i] = want;
    }
    *out = c.buf[i];
    return QG_OK;
}
} // namespace

static void release_cache(RunCache& c)
{
    if (c.ctx) {
        DeviceScope scope(c.ctx->device);
        hipStreamSynchronize(c.ctx->stream);
        if (c.plan) qgemul_plan_destroy(c.plan);
        for (int i = 0; i < RunCache::NBUF; ++i) { if (c.buf[i]) hipFree(c.buf[i]); c.buf[i] = nullptr; c.cap[i] = 0; }
        qgemul_ctx_destroy(c.ctx);
    }
    c.plan = nullptr;
    c.ctx = nullptr;
    c.device = -2;
}

namespace {
void forget_caches()
{
    auto forget = [](RunCache& c) {
        c.plan = nullptr;
        c.ctx = nullptr;
        c.device = -2;
        for (int i = 0; i < RunCache::NBUF; ++i) { c.buf[i] = nullptr; c.cap[i] = 0; }
    };
    forget(g_run);
    for (int i = 0; i < QG_MAX_SHARDS; ++i) { g_shard_ev[i] = nullptr; forget(g_shard[i]); }
}
} // namespace

void qgemul_run_release(void)
{
    release_cache(g_run);
    for (int i = 0; i < QG_MAX_SHARDS; ++i) {
        if (g_shard_ev[i]) { hipEventDestroy(g_shard_ev[i]); g_shard_ev[i] = nullptr; }
        release_cache(g_shard[i]);
    }
}

static int run_view(const qgemul_desc* d, const EpView* ev, void* C, const void* A, const void* B, const void* const* E, const qgemul_opts* o);

int qgemul_run_ep(const qgemul_desc* d, const qgemul_epilogue* ep, void* C, const void* A, const void* B, const void* const* E,
                  const qgemul_opts* o)
{
    const EpView v = {ep, nullptr, nullptr};
    return run_view(d, ep ? &v : nullptr, C, A, B, E, o);
}

int qgemul_run_epc(const qgemul_desc* d, const qgemul_epilogue_cplx* ep, void* C, const void* A, const void* B, const void* const* E,
                   const qgemul_opts* o)
{
    if (!ep) return QG_EINVAL;
    const EpView v = {&ep->part[0], &ep->part[1], ep->e_complex};
    return run_view(d, &v, C, A, B, E, o);
}

static int run_view(const qgemul_desc* d, const EpView* ev, void* C, const void* A, const void* B, const void* const* E, const qgemul_opts* o)
{
    if (!d || !C || !A || !B) return QG_EINVAL;
    const qgemul_epilogue* ep = ev ? ev->re : nullptr;
    qgemul_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.device = -1;
    if (o) opts = *o;
    g_reaper.armed = true;
    static ShutdownHook hook;   // (installed once, at the first call: see RunCacheReaper)
    RunCache& c = g_run;
    if (opts.device < 0 && c.ctx) {   // "current device": follow hipSetDevice calls the caller made between two calls
        int cur = c.device;
        if (hipGetDevice(&cur) == hipSuccess) opts.device = cur;
    }
    if (opts.flags & QG_OPT_ALL_DEVICES) {
        if (ep) return QG_EUNSUPPORTED;   // the element-wise chain runs on one device
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QG_ENOGPU;
        int list[QG_MAX_SHARDS];
        int n = 0;
        for (int i = 0; i < ndev && n < QG_MAX_SHARDS; ++i) list[n++] = i;
        opts.flags &= ~(uint32_t)QG_OPT_ALL_DEVICES;
        return qgemul_run_sharded(d, C, A, B, &opts, list, n);
    }
    const bool same_plan = c.plan && c.pflags == opts.flags && same_desc(c.pd, *d) && c.has_pe == (ep != nullptr) &&
                           (!ep || (same_epilogue(c.pe.part[0], *ep) && c.pe_cplx == (ev->im != nullptr) &&
                                    (!ev->im || (same_epilogue(c.pe.part[1], *ev->im) && !memcmp(c.pe.e_complex, ev->e_cplx, ep->n_stages))))) &&
                           (opts.device < 0 || opts.device == c.device);
    if (!same_plan) {
        // validate before touching the device so that descriptor errors are reported without a GPU
        qgemul_info info;
        int st = classify_view(d, ev, opts.flags, &info);
        if (st != QG_OK) return st;
    }
    if (ep)
        for (uint32_t k = 0; k < ep->n_stages; ++k)
            if (!E || !E[k]) return QG_EINVAL;
    if (d->M == 0 || d->N == 0) return QG_OK;
    int st = QG_OK;
    if (!c.ctx || (opts.device >= 0 && opts.device != c.device)) {
        release_cache(c);   // (only the single-device cache: the sharded entry's contexts and plans stay warm)
        st = qgemul_ctx_create(opts.device, &c.ctx);
        if (st != QG_OK) { c.ctx = nullptr; return st; }
        c.device = c.ctx->device;
    } else {
        QG_HIP(hipSetDevice(c.device));
    }
    qgemul_ctx* ctx = c.ctx;
    if (!same_plan) {
        if (c.plan) { qgemul_plan_destroy(c.plan); c.plan = nullptr; }
        st = plan_create_view(ctx, d, ev, opts.flags, &c.plan);
        if (st != QG_OK) { c.plan = nullptr; return st; }
        c.pd = *d;
        c.has_pe = ep != nullptr;
        c.pe_cplx = ev && ev->im;
        if (ep) c.pe.part[0] = *ep;
        if (c.pe_cplx) { c.pe.part[1] = *ev->im; memcpy(c.pe.e_complex, ev->e_cplx, sizeof c.pe.e_complex); }
        c.pflags = opts.flags;
    }
    qgemul_plan* p = c.plan;
    void *dA, *dB, *dC, *pA, *pB, *pC;
    do {
        const int64_t lda = opts.lda ? opts.lda : (d->transA ? d->K : d->M);
        const int64_t ldb = opts.ldb ? opts.ldb : d->K;
        const int64_t ldc = opts.ldc ? opts.ldc : d->M;
        if (lda < (d->transA ? d->K : d->M) || ldb < d->K || ldc < d->M) { st = QG_EINVAL; break; }
        const size_t bytesA = (size_t)(((d->transA ? d->M : d->K) - 1) * lda + (d->transA ? d->K : d->M)) * p->ha.size;
        const size_t bytesB = (size_t)((d->N - 1) * ldb + d->K) * p->hb.size;
        const size_t bytesC = (size_t)((d->N - 1) * ldc + d->M) * p->hc.size;
        if ((st = cache_buffer(c, 0, bytesA, &dA)) || (st = cache_buffer(c, 1, bytesB, &dB)) || (st = cache_buffer(c, 2, bytesC, &dC)) ||
            (st = cache_buffer(c, 3, (size_t)p->info.packed_bytes[0], &pA)) || (st = cache_buffer(c, 4, (size_t)p->info.packed_bytes[1], &pB)) ||
            (st = cache_buffer(c, 5, (size_t)p->info.packed_bytes[2], &pC)))
            break;
        hipStream_t s = ctx->stream;
        // everything below is queued on the context's stream; ONE synchronisation at the end (the source buffers are the
        // caller's and the call is synchronous, so they stay valid until then)
        if (hipMemcpyAsync(dA, A, bytesA, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(dB, B, bytesB, hipMemcpyHostToDevice, s) != hipSuccess) { st = QG_EHIP; break; }
        // the caller's C may have padding between columns (ldc > M): keep those bytes as they are
        if (ldc != d->M && hipMemcpyAsync(dC, C, bytesC, hipMemcpyHostToDevice, s) != hipSuccess) { st = QG_EHIP; break; }
        if ((st = qgemul_pack(p, QG_OPERAND_A, dA, lda, pA)) || (st = qgemul_pack(p, QG_OPERAND_B, dB, ldb, pB))) break;
        if (!ep && stores_host_c(p)) {
            // the kernel's epilogue writes the reference layout: no packed C, no unpack pass
            if ((st = qgemul_execute_host_c(p, dC, ldc, pA, pB))) break;
            if (hipMemcpyAsync(C, dC, bytesC, hipMemcpyDeviceToHost, s) != hipSuccess) { st = QG_EHIP; break; }
            break;
        }
        if (!ep) {
            if ((st = qgemul_execute(p, pC, pA, pB))) break;
        } else {
            qgemul_ep_args ea;
            memset(&ea, 0, sizeof ea);
            auto raw = [](const void* q, qfmt f) { return (1 + (int)f.I + (int)f.F) <= 32 ? (int64_t) * (const int32_t*)q : *(const int64_t*)q; };
            for (uint32_t k = 0; k < ep->n_stages && !st; ++k) {
                const qgemul_ew_stage& sr = ep->stage[k];
                const qgemul_ew_stage* si = ev->im ? &ev->im->stage[k] : nullptr;
                const bool cplx = si && ev->e_cplx[k];
                const bool t_re = sr.op != QG_EW_PASS && !sr.e_scalar, t_im = si && si->op != QG_EW_PASS && !si->e_scalar;
                if (!t_re && !t_im) {
                    // scalar operand: one element ({re, im} for a complex one); a real scalar feeds both parts, except where the
                    // imaginary part's stage takes the zero of the operand's type (real - complex, QuBLAS.h:3686)
                    if (sr.op != QG_EW_PASS) ea.e_scalar[k] = raw(E[k], sr.e);
                    if (si && si->op != QG_EW_PASS) {
                        if (cplx) {
                            const qfmt f[2] = {sr.e, si->e};
                            ea.e_scalar_im[k] = raw((const char*)E[k] + qg_host_elem(f, 1).off[1], si->e);
                        } else {
                            ea.e_scalar_im[k] = si->op == QG_EW_MUL ? raw(E[k], si->e) : 0;
                        }
                    }
                    continue;
                }
                const qfmt f[2] = {t_re ? sr.e : si->e, si ? si->e : sr.e};
                const size_t bytesE = (size_t)d->M * (size_t)d->N * (size_t)qg_host_elem(f, cplx ? 1 : 0).size;
                void *dE, *pE;
                if ((st = cache_buffer(c, 6 + 2 * (int)k, bytesE, &dE)) || (st = cache_buffer(c, 7 + 2 * (int)k, (size_t)qgemul_packed_e_bytes(p, (int)k), &pE)))
                    break;
                if (hipMemcpyAsync(dE, E[k], bytesE, hipMemcpyHostToDevice, s) != hipSuccess) { st = QG_EHIP; break; }
                if ((st = qgemul_pack_e(p, (int)k, dE, 0, pE))) break;
                ea.e_packed[k] = pE;
                // (real - complex with a tensor operand: the imaginary part's stage has the scalar 0, set by the memset above)
            }
            if (st) break;
            if ((st = qgemul_execute_ep(p, pC, pA, pB, &ea))) break;
        }
        if ((st = qgemul_unpack_c(p, pC, dC, ldc))) break;
        if (hipMemcpyAsync(C, dC, bytesC, hipMemcpyDeviceToHost, s) != hipSuccess) { st = QG_EHIP; break; }
    } while (0);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (st == QG_OK && e != hipSuccess) { g_last_hip = (int)e; st = QG_EHIP; }
    return st;
}

// ---- several GPUs in one process: row bands of C, one per entry of the device list (include/qgemul.h) ----
// One host thread drives every device: all work is queued asynchronously on each device's own stream (H2D of the band of A
// and of B, pack, GEMM, peer copy of the packed C band to the root), the root's stream waits for each band's event, unpacks it
// into the one host-layout C and copies that back.  Bands are whole blocks of 256 rows (every packed row tile divides 256).
int qgemul_run_sharded(const qgemul_desc* d, void* C, const void* A, const void* B, const qgemul_opts* o, const int* devices, int n)
{
    if (!d || !C || !A || !B || !devices || n < 1 || n > QG_MAX_SHARDS) return QG_EINVAL;
    g_reaper.armed = true;
    static ShutdownHook hook;
    qgemul_opts opts;
    memset(&opts, 0, sizeof opts);
    if (o) opts = *o;
    opts.flags &= ~(uint32_t)QG_OPT_ALL_DEVICES;
    {   // validate the whole problem before touching a device (a band of an unsupported descriptor is unsupported too)
        qgemul_info info;
        int st = qgemul_classify(d, opts.flags, &info);
        if (st != QG_OK) return st;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QG_ENOGPU;
    for (int i = 0; i < n; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return QG_EINVAL;
    if (d->M == 0 || d->N == 0) return QG_OK;
    const int64_t lda = opts.lda ? opts.lda : (d->transA ? d->K : d->M);
    const int64_t ldb = opts.ldb ? opts.ldb : d->K;
    const int64_t ldc = opts.ldc ? opts.ldc : d->M;
    if (lda < (d->transA ? d->K : d->M) || ldb < d->K || ldc < d->M) return QG_EINVAL;

    // contiguous bands of whole 256-row blocks, sizes differing by at most one block (qublas_amd/dist.py: row_partition)
    const int64_t ALIGN = 256, units = (d->M + ALIGN - 1) / ALIGN;
    int64_t row0[QG_MAX_SHARDS], rows[QG_MAX_SHARDS];
    {
        int64_t u0 = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t u = units / n + (i < units % n ? 1 : 0);
            const int64_t r0 = u0 * ALIGN < d->M ? u0 * ALIGN : d->M, r1 = (u0 + u) * ALIGN < d->M ? (u0 + u) * ALIGN : d->M;
            row0[i] = r0;
            rows[i] = r1 - r0;
            u0 += u;
        }
    }
    int st = QG_OK;
    const int root = 0;   // devices[0] assembles C
    // contexts first: the root's is needed by every band
    for (int i = 0; i < n && st == QG_OK; ++i) {
        RunCache& c = g_shard[i];
        if (c.ctx && c.device != devices[i]) release_cache(c);
        if (!c.ctx) {
            st = qgemul_ctx_create(devices[i], &c.ctx);
            if (st != QG_OK) { c.ctx = nullptr; break; }
            c.device = c.ctx->device;
        }
        if (!g_shard_ev[i]) {
            DeviceScope scope(c.device);
            if (hipEventCreateWithFlags(&g_shard_ev[i], hipEventDisableTiming) != hipSuccess) { g_shard_ev[i] = nullptr; st = QG_EHIP; }
        }
    }
    if (st != QG_OK) return st;
    RunCache& rc = g_shard[root];
    const qfmt* cf = d->c;
    const QHostElem hcel = qg_host_elem(cf, d->is_complex);
    const size_t bytesC = (size_t)((d->N - 1) * ldc + d->M) * hcel.size;
    void* dC = nullptr;   // host-layout C on the root
    {
        DeviceScope scope(rc.device);
        if ((st = cache_buffer(rc, 2, bytesC, &dC)) != QG_OK) return st;
        // the caller's C may have padding between columns (ldc > M): keep those bytes as they are
        if (ldc != d->M && hipMemcpyAsync(dC, C, bytesC, hipMemcpyHostToDevice, rc.ctx->stream) != hipSuccess) return QG_EHIP;
    }
    for (int i = 0; i < n && st == QG_OK; ++i) {
        if (rows[i] == 0) continue;
        RunCache& c = g_shard[i];
        DeviceScope scope(c.device);
        if (scope.err != hipSuccess) { st = QG_EHIP; break; }
        qgemul_desc bd = *d;
        bd.M = rows[i];
        if (!(c.plan && c.pflags == opts.flags && !c.has_pe && same_desc(c.pd, bd))) {
            if (c.plan) { qgemul_plan_destroy(c.plan); c.plan = nullptr; }
            st = qgemul_plan_create(c.ctx, &bd, opts.flags, &c.plan);
            if (st != QG_OK) { c.plan = nullptr; break; }
            c.pd = bd;
            c.has_pe = false;
            c.pflags = opts.flags;
        }
        qgemul_plan* p = c.plan;
        hipStream_t s = c.ctx->stream;
        const size_t ea = (size_t)p->ha.size, eb = (size_t)p->hb.size;
        // the band of A in host layout, tight on the device: A declared dim<M,K> (column-major) is strided in the band's rows,
        // A declared dim<K,M> (QgemulTransposedA) is one contiguous run of columns
        const size_t bytesA = d->transA ? (size_t)((rows[i] - 1) * lda + d->K) * ea : (size_t)rows[i] * (size_t)d->K * ea;
        const size_t bytesB = (size_t)((d->N - 1) * ldb + d->K) * eb;
        void *dA, *dB, *pA, *pB, *pC, *pCroot = nullptr;
        if ((st = cache_buffer(c, 0, bytesA, &dA)) || (st = cache_buffer(c, 1, bytesB, &dB)) ||
            (st = cache_buffer(c, 3, (size_t)p->info.packed_bytes[0], &pA)) || (st = cache_buffer(c, 4, (size_t)p->info.packed_bytes[1], &pB)) ||
            (st = cache_buffer(c, 5, (size_t)p->info.packed_bytes[2], &pC)))
            break;
        hipError_t he;
        int64_t band_lda;
        if (d->transA) {
            he = hipMemcpyAsync(dA, (const char*)A + (size_t)row0[i] * (size_t)lda * ea, bytesA, hipMemcpyHostToDevice, s);
            band_lda = lda;
        } else {
            he = hipMemcpy2DAsync(dA, (size_t)rows[i] * ea, (const char*)A + (size_t)row0[i] * ea, (size_t)lda * ea, (size_t)rows[i] * ea,
                                  (size_t)d->K, hipMemcpyHostToDevice, s);
            band_lda = rows[i];
        }
        if (he != hipSuccess || hipMemcpyAsync(dB, B, bytesB, hipMemcpyHostToDevice, s) != hipSuccess) { st = QG_EHIP; break; }
        if ((st = qgemul_pack(p, QG_OPERAND_A, dA, band_lda, pA)) || (st = qgemul_pack(p, QG_OPERAND_B, dB, ldb, pB)) ||
            (st = qgemul_execute(p, pC, pA, pB)))
            break;
        // the packed band goes to the root (a peer copy; the same device twice: a device-to-device copy), the root unpacks it
        if (i == root) {
            pCroot = pC;
        } else {
            // one landing buffer per band on the root: the slots behind the six fixed ones, grown on demand
            const int slot = 6 + (i - 1);
            if (slot >= RunCache::NBUF) { st = QG_EUNSUPPORTED; break; }
            {
                DeviceScope rscope(rc.device);
                if ((st = cache_buffer(rc, slot, (size_t)p->info.packed_bytes[2], &pCroot)) != QG_OK) break;
            }
            if (hipMemcpyPeerAsync(pCroot, rc.device, pC, c.device, (size_t)p->info.packed_bytes[2], s) != hipSuccess) { st = QG_EHIP; break; }
        }
        if (hipEventRecord(g_shard_ev[i], s) != hipSuccess) { st = QG_EHIP; break; }
        {
            DeviceScope rscope(rc.device);
            if (i != root && hipStreamWaitEvent(rc.ctx->stream, g_shard_ev[i], 0) != hipSuccess) { st = QG_EHIP; break; }
            QCGeom g = p->pc;          // the band's packed geometry, written at row offset row0 of the full C
            g.ldc = ldc;
            char* dst = (char*)dC + (size_t)row0[i] * hcel.size;
            if (qg_launch_unpack_c(g, pCroot, dst, rc.ctx->stream, (opts.flags & QG_OPT_GENERIC_LAYOUT) ? 1 : 0) != hipSuccess) { st = QG_EHIP; break; }
        }
    }
    if (st == QG_OK) {
        DeviceScope scope(rc.device);
        if (hipMemcpyAsync(C, dC, bytesC, hipMemcpyDeviceToHost, rc.ctx->stream) != hipSuccess) st = QG_EHIP;
    }
    // the call is synchronous: every stream drains before the caller's buffers may change (root last: it waits for the others)
    for (int i = n - 1; i >= 0; --i) {
        if (!g_shard[i].ctx) continue;
        DeviceScope scope(g_shard[i].device);
        const hipError_t e = hipStreamSynchronize(g_shard[i].ctx->stream);
        if (st == QG_OK && e != hipSuccess) { g_last_hip = (int)e; st = QG_EHIP; }
    }
    return st;
}

} // extern "C"
