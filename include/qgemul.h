/*
 * qgemul.h — C-ABI boundary of the MI355X (gfx950) fixed-point GEMM engine.
 *
 * This is the drop-in boundary for ONE path of pikapuma/QuBLAS: the fixed-point matrix
 * multiply `Qgemul<QgemulAddArgs<…>, QgemulMulArgs<…>, QgemulTransposedA<…>>(C, A, B)`
 * (reference: readme.md:84-87 — the only place the function exists in the snapshot; its
 * semantics are the composition of Qmul  include/QuBLAS.h:3980-3985 / :3146-3170,
 * Qreduce include/QuBLAS.h:4960-4990 / :5014-5018 and the converting constructor :2398-2411).
 *
 * The reference resolves every quantisation decision at compile time from template tags.
 * The host header (include/QuBLAS_amd.h, or the binding in include/qgemul_reference_binding.hpp used with
 * the reference's own header) evaluates the same merger rules at compile time and lowers the
 * result to the plain-data descriptor below; nothing in this file is a template, nothing
 * depends on PyTorch, and every pointer is a plain pointer.
 *
 * Conventions
 *   - A format is (I, F, S, Q, O) = intBits, fracBits, isSigned, QuMode, OfMode
 *     (reference tags include/QuBLAS.h:2346-2359).  W = I + F magnitude bits.  A raw integer r
 *     stands for the real value r * 2^-F.  Storage always has a sign bit (QuBLAS.h:2384-2385).
 *   - Host ("reference") layout of a tensor element: int32_t when 1+W <= 32, int64_t when
 *     1+W <= 64, holding the sign-extended raw value (ArbiInt<N<=64>, QuBLAS.h:347-353); for
 *     65 <= 1+W <= 128 — C only — two little-endian uint64_t words, 8-byte aligned, the upper one
 *     carrying the sign (ArbiInt<N>64> is std::array<uint64_t, ceil(N/64)>, QuBLAS.h:572-573).  A complex
 *     element is the C struct { real; imag; } of those (QuBLAS.h:2512-2513).  Matrices are
 *     column-major, element (i,j) at i + j*ld (QuBLAS.h:2680-2692).
 *   - Mode codes are the reference's numeric values (QuBLAS.h:1986-1999, :2209-2225).
 */
#ifndef QGEMUL_H_
#define QGEMUL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QGEMUL_ABI_VERSION 1u
#define QG_MAX_LEVELS 40 /* tree levels = ceil(log2 K); 40 covers K up to 2^40 */

/* QuMode codes — RND::* / TRN::* `value` members, QuBLAS.h:1986-1999 */
enum {
    QG_RND_POS_INF = 0,
    QG_RND_NEG_INF = 1,
    QG_RND_ZERO = 2,
    QG_RND_INF = 3,
    QG_RND_CONV = 4,
    QG_TRN_TCPL = 5,
    QG_TRN_SMGN = 6
};

/* OfMode codes — SAT::* / WRP::* `value` members, QuBLAS.h:2209-2225 */
enum {
    QG_SAT_TCPL = 0,
    QG_SAT_ZERO = 1,
    QG_SAT_SMGN = 2,
    QG_WRP_TCPL = 3,
    QG_WRP_TCPL_SAT = 4 /* a stub in the reference: intConvert returns its input (QuBLAS.h:2336-2344) and the target's storage word
                           (int32_t / int64_t) keeps what fits.  Runs where the value provably stays in its format (a no-op) and
                           as C's OfMode (the root lands in C's host word unclamped); elsewhere QG_EUNSUPPORTED */
};

/* complex multiply algorithm — BasicComplexMul QuBLAS.h:3426-3445, TFComplexMul :3510-3534 */
enum { QG_CMUL_NONE = 0, QG_CMUL_BASIC = 1, QG_CMUL_TF = 2 };

/* slots of qgemul_desc.mul[] */
enum {
    QG_MUL_REAL = 0, /* real GEMM: the product format (MulMerger result, QuBLAS.h:3107-3120) */
    /* BasicComplexMul, x = a+bi, y = c+di (QuBLAS.h:3439-3440) */
    QG_B_AC = 0, QG_B_BD = 1, QG_B_AD = 2, QG_B_BC = 3,
    QG_B_RE = 4, /* Qsub<acbdT>(ac, bd) */
    QG_B_IM = 5, /* Qadd<adbcT>(ad, bc) */
    /* TFComplexMul (QuBLAS.h:3524-3529); the lowering has already applied the reference's two
     * quirks: B is quantised with badT and C with cdbT (:3525-3526), and (b-a) always uses the
     * default merge because baT is never honoured (:3515). */
    QG_T_AB = 0,  /* Qadd(a, b) */
    QG_T_CD = 1,  /* Qadd(c, d) */
    QG_T_BA = 2,  /* Qsub(b, a) */
    QG_T_A = 3,   /* Qmul(ab, c) */
    QG_T_B = 4,   /* Qmul(cd, b) */
    QG_T_C = 5,   /* Qmul(ba, d) */
    QG_T_RE = 6,  /* Qsub(A, B) */
    QG_T_IM = 7   /* Qsub(B, C) */
};

typedef struct qfmt {
    int16_t I;  /* intBits  (may be negative) */
    int16_t F;  /* fracBits (may be negative) */
    uint8_t S;  /* isSigned */
    uint8_t Q;  /* QuMode code */
    uint8_t O;  /* OfMode code */
    uint8_t pad;
} qfmt;

/*
 * Fully-resolved description of one Qgemul call.  Every arithmetic node of the reference
 * expression has its RESULT format written out; the dataflow between nodes is fixed:
 *
 *   p[k]   = product(A'[i,k], B[k,j])                  mul[] slots above
 *   level l: next[t] = cvt_{level[l]}( add_{level_add[l]}(cur[2t], cur[2t+1]) ),
 *            odd leftover next[last] = cvt_{level[l]}(cur[len-1])        (QuBLAS.h:4973-4980; flags: QG_DESC_LEFTOVER0_COPY)
 *   C[i,j] = cvt_{c}( root )                                            (QuBLAS.h:2398-2411)
 *
 * where op_{f}(x) means: align/compute exactly, round to f.F with f.Q (fracConvert,
 * QuBLAS.h:2002-2204), then overflow-handle to (f.I, f.F, f.S) with f.O (intConvert, :2227-2334);
 * and cvt_{f} is the identity when source and target formats are equal in all five fields.
 * For a real GEMM level_add[l] == level[l] (Qadd<T_l> yields T_l directly).  For a complex GEMM
 * the reference ignores a complex type passed as Qadd tag (QuBLAS.h:3549-3564 + :3097-3099), so
 * level_add[l] is the default merge of the incoming format and level[l] is the level buffer's
 * type (QuBLAS.h:4966).  Index [0] = real part, [1] = imaginary part.
 */
/* qgemul_desc.flags
 * QG_DESC_LEFTOVER0_COPY: the odd leftover of tree level 0 is copied into the level buffer UNCONVERTED.  In the reference that
 * copy (`res[last] = quants[len-1]`, QuBLAS.h:4977-4980) is the identity whenever level 0's type is the element type itself.
 * The Qreduce lowering of a signed SAT::SMGN element type names the element's SAT::TCPL twin as the leaf format (so that the
 * raw minimum -2^W, which Qu::fill() can produce, reaches the adders as it is); leaf and level 0 then differ in the descriptor
 * although they are one type in the reference, and this flag restores the identity copy.  Only odd K is affected. */
enum { QG_DESC_LEFTOVER0_COPY = 1u, QG_DESC_REFERENCE_ARTEFACTS = 2u };
/* QG_DESC_REFERENCE_ARTEFACTS (opt-in; never set by the Qgemul lowerings themselves): reproduce, instead of refusing, a result of
 * the reference that is an artefact of its implementation rather than of the arithmetic it documents.  One is covered: C of an
 * UNSIGNED WRP::TCPL format with exactly 32 value bits.  The reference's mask for it is ArbiInt<32>::allOnes(), whose data is -1
 * (QuBLAS.h:361-377), so `val & mask` (:2328-2331) masks nothing and the value lands in the 33-bit storage (an int64_t word)
 * unwrapped: -7 -> -7, 2^33 + 5 -> 2^33 + 5 (tests/golden/ref_scalar_6).  With the flag such a C behaves as WRP::TCPL_SAT does
 * (the value narrowed to the storage word, host-word containers); without it the descriptor is QG_EUNSUPPORTED when a value
 * can leave the format.  The same format as a product / level type, and RND over a shift of exactly 32 or 64 bits (:361-377 via
 * :2043-2160), stay refused with or without the flag. */

typedef struct qgemul_desc {
    uint32_t abi;       /* QGEMUL_ABI_VERSION */
    uint8_t transA;     /* QgemulTransposedA<true>: A is declared dim<K,M>, A'[i,k] = A[k,i] */
    uint8_t is_complex; /* operands are Qcomplex */
    uint8_t cmul;       /* QG_CMUL_* (QG_CMUL_NONE for real) */
    uint8_t flags;      /* QG_DESC_* (0 for every Qgemul; the Qreduce lowering may set QG_DESC_LEFTOVER0_COPY) */
    int64_t M, N, K;    /* C is M x N, reduction length K (runtime values, never template depth) */
    qfmt a[2], b[2], c[2];
    qfmt mul[8];
    uint32_t n_levels;  /* ceil(log2 K), 0 when K == 1 */
    uint32_t reserved2;
    qfmt level_add[2][QG_MAX_LEVELS];
    qfmt level[2][QG_MAX_LEVELS];
} qgemul_desc;

/* leading dimensions of the host-layout operands, in elements; 0 = tight (rows of the declared dim) */
typedef struct qgemul_opts {
    int64_t lda, ldb, ldc;
    int32_t device;        /* HIP device ordinal for qgemul_run; -1 = current */
    uint32_t flags;        /* QG_OPT_* */
} qgemul_opts;

enum {
    QG_OPT_FORCE_TREE = 1u,   /* run the exact tree kernel even when the linear class is provable */
    QG_OPT_CHECK_RANGE = 2u,  /* validate that A and B raw values lie inside their formats */
    QG_OPT_GENERIC_TREE = 4u, /* tree class: use the any-descriptor 64-bit kernel, not the 32-bit fast path */
    QG_OPT_RUNTIME_MODES = 8u, /* 32-bit tree kernel: use the runtime-mode variant even when a fixed-mode one applies */
    /* element-wise epilogue placement.  Default: inside the 3x3-limb MFMA kernel's epilogue when the planner has bounded
     * the chain by 32-bit arithmetic (measured 0.473 vs 0.490 ms at 4096^3), otherwise ONE linear pass over the stored C
     * (HBM-bound, 5-6 TB/s).  The single-limb kernels also have a fused variant but it is slower than the pass (one
     * workgroup per CU: the matrix cores idle during the longer epilogue), so it is opt-in. */
    QG_OPT_FUSED_EPILOGUE = 16u,  /* fuse wherever a fused variant exists (32-bit chains on the single-limb kernels too) */
    QG_OPT_UNFUSED_EPILOGUE = 32u, /* never fuse: always the pass after the kernel */
    /* result-identical kernel choices, for equivalence tests and same-process A/B timing */
    QG_OPT_GENERIC_LAYOUT = 64u,   /* pack / unpack with the any-format kernels even where a fast path exists */
    /* qgemul_pack_f64 on an element whose QuMode is RND::CONV: the reference's Qu_s(double) takes a 2400-bit path whose CONV
     * branch returns a multi-word artefact (the format maximum for every negative input; tests/test_from_double.py), so by
     * default such a pack is REJECTED (QG_EUNSUPPORTED).  With this flag the engine converts with the arithmetic definition
     * of RND::CONV (round half to even) — what the <= 62-bit conversions on the Qgemul path do and the reference pins. */
    QG_OPT_ARITHMETIC_CONV = 256u,
    /* qgemul_run: shard the rows of C over every gfx950 device this process can see (qgemul_run_sharded with all of them) */
    QG_OPT_ALL_DEVICES = 512u,
    QG_OPT_BALANCED_LIMBS = 1024u, /* linear class: never store an operand CENTRED (x - c in balanced int8 limbs, the centre taken back out
                                      with row sums in the epilogue: one limb fewer for signed formats of 16 / 24 / 32 bits and for
                                      unsigned formats) — the plain balanced limbs of every round before; result-identical (tests) */
    QG_OPT_LOCKSTEP_TILES = 128u   /* large single-limb problems: the 64-byte-k-tile kernel whose waves run in lock step
                                    * (k_mfma16) instead of the two-group kernel on 128-byte k-tiles (k_mfma_pp) */
};

/* status codes */
enum {
    QG_OK = 0,
    QG_EINVAL = -1,       /* malformed descriptor / null pointer / bad size */
    QG_EUNSUPPORTED = -2, /* e.g. WRP::TCPL_SAT, an intermediate wider than 120 bits (unrounded products: 127), an operand element wider
                             than 64 storage bits, a combination the reference itself cannot compile or computes as a width artefact */
    QG_EHIP = -3,         /* a HIP runtime call failed; qgemul_last_hip_error() has the code */
    QG_ERCCL = -4,
    QG_ERANGE = -5,       /* QG_OPT_CHECK_RANGE: an input raw value is outside its format */
    QG_ENOGPU = -6        /* no gfx950 device visible: the engine has no CPU fallback */
};

/* exactness class of a descriptor */
enum {
    QG_CLASS_LINEAR = 1, /* every intermediate conversion is provably the identity: exact integer
                            dot product + ONE round/overflow into C (MFMA / wide-int path) */
    QG_CLASS_TREE = 2    /* products and tree nodes must be quantised one by one, in tree order */
};

typedef struct qgemul_info {
    int32_t cls;            /* QG_CLASS_* */
    int32_t supported;      /* 1 if the GPU engine can run it */
    int32_t max_bits;       /* widest signed intermediate the expression can produce */
    int32_t in_bits[2];     /* storage bits (1+W) of A and B elements (max over parts) */
    int32_t limbs[2];       /* int8 limbs per A / B element on the MFMA path (0 = not MFMA) */
    int32_t kernel;         /* QG_KERNEL_* chosen by the planner */
    int32_t host_elem_bytes[3]; /* sizeof host element of A, B, C (complex: whole struct) */
    int32_t host_imag_off[3];   /* byte offset of .imag inside a complex host element */
    int64_t packed_bytes[3];    /* device-private packed sizes of A, B, C for this M,N,K */
    double  ops;                /* 2*M*N*K for real; x3 (TF) or x4 (Basic) real MACs for complex */
    char    reason[96];         /* why unsupported / why tree */
} qgemul_info;

enum {
    QG_KERNEL_NONE = 0,
    QG_KERNEL_MFMA_I8 = 1,      /* class L, operands <= 8 storage bits: v_mfma_i32_*_i8 */
    QG_KERNEL_MFMA_I8_LIMB = 2, /* class L, wider operands split into int8 limbs */
    QG_KERNEL_TREE_I32 = 3,     /* class T, all intermediates fit 32 bits */
    QG_KERNEL_TREE_I64 = 4,     /* class T, 64-bit intermediates */
    QG_KERNEL_TREE_CPLX = 5,    /* class T, complex, any descriptor (64-bit) */
    QG_KERNEL_TREE_CPLX_I32 = 6,/* class T, complex, K = 2^p and 32-bit intermediates */
    QG_KERNEL_MFMA_CPLX = 7,    /* class L, complex: four real int8-limb dot products on MFMA + one combine pass */
    QG_KERNEL_GEMV_I32 = 8,     /* class T, N = 1 (batched Qreduce / GEMV), K = 2^p >= 16, 32-bit values: one wave per row (or per 256/K rows) */
    QG_KERNEL_GEMV_I64 = 9,     /* the same with 64-bit tree values on elements of at most 32 storage bits (32-bit words, wide level types) */
    QG_KERNEL_TREE_I128 = 10    /* class T with intermediates / level formats / C beyond 62 bits (up to 120): the general tree kernel on
                                   128-bit values — the reference's multi-word ArbiInt<N > 64>, QuBLAS.h:566-912; real or complex */
};

/* ---- descriptor analysis: pure host code, works without a GPU ---- */
int qgemul_classify(const qgemul_desc* d, uint32_t opt_flags, qgemul_info* out);
const char* qgemul_strerror(int status);
uint32_t qgemul_abi_version(void);
int qgemul_last_hip_error(void);

/* ---- one-shot entry: what Qgemul<…>(C, A, B) calls.  Host pointers in reference layout;
 *      C is fully overwritten; the call is synchronous (QuBLAS is synchronous, single-threaded). */
int qgemul_run(const qgemul_desc* d, void* C, const void* A, const void* B, const qgemul_opts* o);
/* qgemul_run / qgemul_run_ep keep a cache per calling thread (context, the plan of the last descriptor, grow-only device
 * buffers) so that calls in a loop cost microseconds, not milliseconds; this frees the calling thread's cache. */
void qgemul_run_release(void);

/* ---- resident-data API (benchmarks, multi-GPU row shards, repeated calls) ---- */
typedef struct qgemul_ctx qgemul_ctx;   /* one per host thread / device; owns streams + workspace */
typedef struct qgemul_plan qgemul_plan; /* a descriptor bound to kernels and packed layouts */

int qgemul_ctx_create(int device, qgemul_ctx** out);
void qgemul_ctx_destroy(qgemul_ctx* c);
int qgemul_ctx_sync(qgemul_ctx* c);
void* qgemul_ctx_stream(qgemul_ctx* c); /* the hipStream_t the engine launches on */

int qgemul_plan_create(qgemul_ctx* c, const qgemul_desc* d, uint32_t opt_flags, qgemul_plan** out);
void qgemul_plan_destroy(qgemul_plan* p);
int qgemul_plan_info(const qgemul_plan* p, qgemul_info* out);

/* device memory owned by the caller, allocated through the engine (plain hipMalloc/hipFree) */
int qgemul_dev_alloc(qgemul_ctx* c, size_t bytes, void** out);
int qgemul_dev_free(qgemul_ctx* c, void* p);
int qgemul_memcpy_h2d(qgemul_ctx* c, void* dst_dev, const void* src_host, size_t bytes);
int qgemul_memcpy_d2h(qgemul_ctx* c, void* dst_host, const void* src_dev, size_t bytes);

/* which operand */
enum { QG_OPERAND_A = 0, QG_OPERAND_B = 1, QG_OPERAND_C = 2 };

/* reference-layout (device-resident copy) -> packed.  `src_dev` is a device pointer to the same
 * bytes the host tensor holds; ld in elements (0 = tight). */
int qgemul_pack(qgemul_plan* p, int operand, const void* src_dev, int64_t ld, void* packed_dev);
/* quantise-on-load: `src_dev` is a device pointer to a column-major tensor of doubles with the
 * operand's declared shape (complex: {re, im} pairs); every value is converted exactly as Qu_s(double) does
 * (QuBLAS.h:2387-2393: the element type's own QuMode, then its OfMode) and packed in one pass.  An element type with
 * QuMode<RND::CONV> is refused (QG_EUNSUPPORTED) unless the plan was created with QG_OPT_ARITHMETIC_CONV. */
int qgemul_pack_f64(qgemul_plan* p, int operand, const double* src_dev, int64_t ld, void* packed_dev);
/* packed C -> reference layout (device-resident), ready for one D2H copy */
int qgemul_unpack_c(qgemul_plan* p, const void* packed_dev, void* dst_dev, int64_t ld);
/* the hot path: packed A, packed B -> packed C, asynchronous on the ctx stream */
int qgemul_execute(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB);
/* The hot path with C in the REFERENCE layout on the device (column-major, element (i, j) at i + j * ldc; ldc in elements,
 * 0 = M): packed A, packed B -> host-layout C.  Where the kernel's epilogue can store its runs of rows straight into that
 * layout (linear class on the two-group MFMA kernels with a 4- or 8-byte C element: qgemul_plan_stores_host_c() = 1) no
 * packed C and no unpack pass exist; every other plan runs qgemul_execute into a buffer of the plan plus qgemul_unpack_c —
 * the call always works.  Asynchronous on the ctx stream; not for plans with an epilogue. */
int qgemul_execute_host_c(qgemul_plan* p, void* C_dev, int64_t ldc, const void* packedA, const void* packedB);
int qgemul_plan_stores_host_c(const qgemul_plan* p);
/* synthetic operand straight into packed form: raw values uniform over the format's full range
 * (dist 0, what Qu::fill() does, QuBLAS.h:526-536) or |raw| < 2^(W/2) (dist 1), from
 * splitmix64(seed ^ linear_index) — the same generator oracle/qoracle.c implements on the host. */
int qgemul_fill_packed(qgemul_plan* p, int operand, uint64_t seed, int dist, void* packed_dev);
/* time `iters` back-to-back qgemul_execute launches with HIP events on the ctx stream;
 * returns average milliseconds per launch */
int qgemul_time_execute(qgemul_plan* p, void* packedC, const void* packedA, const void* packedB,
                        int warmup, int iters, float* avg_ms);

/* ---- fused element-wise epilogue ----
 * The reference's lazy tensor operators (Qmul/Qadd/Qsub on tensors, QuBLAS.h:3780-3877, front-ends
 * :4079-4100) evaluate  Qop<tags...>(x[i], e[i])  per element on operator[] (:3795-3798, :3828-3831,
 * :3861-3864; a scalar operand is used as is, autoCall :3767-3778), and a tensor constructed from such an
 * expression converts every element into its own element type (:2732-2746, converting constructor
 * :2398-2411).  So after  Qgemul<...>(C, A, B)  the statement
 *        Qu<dim<M,N>, DT> D = Qadd<t2...>(Qmul<t1...>(C, s), Bias);
 * The tensor front-ends take Qu tensors only (:4080-4100), not expressions, so a chain materialises a tensor per
 * operator; after  Qgemul<...>(C, A, B)  the statements
 *        Qu<dim<M,N>, T1> t = Qmul<t1...>(C, s);      Qu<dim<M,N>, DT> D = Qadd<t2...>(t, Bias);
 * are, per element,  D[i] = cvt_DT( Qadd<t2>( cvt_T1( Qmul<t1>(C[i], s) ), Bias[i] ) ).  The epilogue below runs that
 * chain inside the GEMM kernel's own epilogue, on the value the kernel has just converted into C's element type, so
 * neither C nor the intermediate tensors go to memory.  (Complex GEMMs: qgemul_epilogue_cplx below.)
 *   x_0 = C[i,j] (format desc.c[0]);   y_k = Qop_k(x_k, e_k) or Qop_k(e_k, x_k), format stage[k].r;
 *   x_{k+1} = cvt_{stage[k].t}(y_k) for k < n-1;   D[i,j] = cvt_d(y_{n-1})   (cvt = the identity when source and
 *   target agree in all five fields).  n = 0:  D = C converted element by element (:2766-2777). */
#define QG_MAX_EW 4
enum { QG_EW_ADD = 1, QG_EW_SUB = 2, QG_EW_MUL = 3,
       QG_EW_PASS = 4 /* complex chains only: this part goes through the operator unchanged (qgemul_epilogue_cplx) */ };
typedef struct qgemul_ew_stage {
    uint8_t op;        /* QG_EW_* */
    uint8_t x_first;   /* 1: Qop(x, e)   0: Qop(e, x)  — the order matters for QG_EW_SUB */
    uint8_t e_scalar;  /* 1: e is a scalar (isScalar operand)   0: a tensor of dim<M,N> */
    uint8_t reserved;
    qfmt e;            /* element format of the operand */
    qfmt r;            /* resolved result format of this Qop<tags...> (MulMerger :3107-3120 / AddMerger :3126-3139) */
    qfmt t;            /* element type of the tensor this stage's result is assigned to (ignored for the last stage: d) */
} qgemul_ew_stage;
typedef struct qgemul_epilogue {
    uint32_t n_stages; /* 0 .. QG_MAX_EW (0: D is C converted into d) */
    uint32_t reserved;
    qgemul_ew_stage stage[QG_MAX_EW];
    qfmt d;            /* element format of the destination tensor */
} qgemul_epilogue;
/* run-time operands of the stages: packed tensors (qgemul_pack_e) or one raw scalar value each */
typedef struct qgemul_ep_args {
    const void* e_packed[QG_MAX_EW];
    int64_t e_scalar[QG_MAX_EW];
    int64_t e_scalar_im[QG_MAX_EW];   /* complex chains (qgemul_epilogue_cplx): the scalar the IMAGINARY parts' stage k uses */
} qgemul_ep_args;

/* classify / plan with an epilogue: info.host_elem_bytes[2] and info.packed_bytes[2] then describe D, and
 * qgemul_unpack_c unpacks D.  ep == NULL is qgemul_classify / qgemul_plan_create. */
int qgemul_classify_ep(const qgemul_desc* d, const qgemul_epilogue* ep, uint32_t opt_flags, qgemul_info* out);
int qgemul_plan_create_ep(qgemul_ctx* c, const qgemul_desc* d, const qgemul_epilogue* ep, uint32_t opt_flags, qgemul_plan** out);
/* 1 when qgemul_execute_ep runs the chain inside the GEMM kernel, 0 when it runs as its own pass after it */
int qgemul_plan_fuses_epilogue(const qgemul_plan* p);
/* Where the non-plane parts of a packed operand of the linear class sit (introspection for tests and tools; device-private layout):
 * out[0] = byte offset of the 256-byte plane-mask trailer (0: none), out[1] = byte offset of the int64 row sums of a CENTRED
 * operand (0: none; QG_OPT_BALANCED_LIMBS), out[2] = rows of the packed operand (padded), out[3] = the centre taken off every
 * stored value.  Composite plans (limb groups / k-chunks) report 0 for the trailer: every sub-operand has its own. */
int qgemul_plan_packed_layout(const qgemul_plan* p, int operand, int64_t out[4]);
/* bytes of one packed tensor operand of stage k (0 for a scalar stage) */
int64_t qgemul_packed_e_bytes(const qgemul_plan* p, int stage);
/* reference-layout tensor operand of stage k (device-resident copy, column-major M x N, ld in elements,
 * 0 = tight; int32/int64 raw values like any tensor) -> the plan's packed-C layout */
int qgemul_pack_e(qgemul_plan* p, int stage, const void* src_dev, int64_t ld, void* packed_dev);
/* the hot path with the epilogue: packed A, packed B, stage operands -> packed D */
int qgemul_execute_ep(qgemul_plan* p, void* packedD, const void* packedA, const void* packedB, const qgemul_ep_args* args);
int qgemul_time_execute_ep(qgemul_plan* p, void* packedD, const void* packedA, const void* packedB, const qgemul_ep_args* args,
                           int warmup, int iters, float* avg_ms);
/* one-shot: host pointers in reference layout; E[k] points to stage k's tensor (tight, column-major M x N) or to its
 * one scalar element; D (ldc from opts) is fully overwritten */
int qgemul_run_ep(const qgemul_desc* d, const qgemul_epilogue* ep, void* D, const void* A, const void* B,
                  const void* const* E, const qgemul_opts* o);

/* ---- element-wise operators after a COMPLEX GEMM ----
 * Complex Qadd / Qsub (QuBLAS.h:3549-3589) and every operator between a complex and a REAL value (Qmul :3604-3644,
 * Qadd :3648-3676, Qsub :3680-3707) act on the real and the imaginary parts independently, each part with its own tags
 * (realT<...> / imagT<...>, :3537-3547), and a complex tensor's converting assignment converts part by part (:2527-2530).
 * A chain of them after a complex Qgemul is therefore TWO real chains, part[0] over the real parts and part[1] over the
 * imaginary parts, with the same number of stages:
 *   complex (+|-) complex      both parts: ADD / SUB with that part of the operand
 *   complex  *  real           both parts: MUL with the real operand
 *   complex  +  real, real + complex, complex - real
 *                              real part: ADD / SUB with the operand; imaginary part: QG_EW_PASS (carried over in its own
 *                              format, :3670 / :3654 / :3701)
 *   real - complex             real part: SUB (e, x); imaginary part: SUB (e, x) with a scalar operand of the real
 *                              operand's format whose value is 0 (:3686: Qsub<tags>(Qu_s<realArgs1...>(), f2.imag))
 * complex x complex multiplication (BasicComplexMul / TFComplexMul, :3426-3534) mixes the parts and is not an epilogue
 * stage (the headers refuse to lower it).  The chain always runs as its own pass after the complex kernel.
 * Operands: stage k's tensor operand is ONE packed buffer; e_complex[k] = 1: a complex tensor packed like the plan's
 * packed C ([2][M][N], both parts in the container of the wider one), part p reads its half; 0: a real tensor, which
 * either part may read.  Scalars: e_scalar[k] for part[0], e_scalar_im[k] for part[1]. */
typedef struct qgemul_epilogue_cplx {
    qgemul_epilogue part[2];
    uint8_t e_complex[QG_MAX_EW];
    uint8_t reserved[4];
} qgemul_epilogue_cplx;
int qgemul_classify_epc(const qgemul_desc* d, const qgemul_epilogue_cplx* ep, uint32_t opt_flags, qgemul_info* out);
int qgemul_plan_create_epc(qgemul_ctx* c, const qgemul_desc* d, const qgemul_epilogue_cplx* ep, uint32_t opt_flags, qgemul_plan** out);
/* one-shot, host pointers: E[k] points to stage k's tensor (tight, column-major M x N; {re, im} elements when
 * e_complex[k]) or to its one scalar element; qgemul_packed_e_bytes / qgemul_pack_e / qgemul_execute_ep /
 * qgemul_unpack_c serve plans of both kinds */
int qgemul_run_epc(const qgemul_desc* d, const qgemul_epilogue_cplx* ep, void* D, const void* A, const void* B,
                   const void* const* E, const qgemul_opts* o);

/* ---- several GPUs in one process ----
 * The M x N outputs are independent: device i computes a band of whole 256-row blocks of C from the matching rows of A' and
 * all of B (replicated); the packed C bands travel to devices[0] with peer copies (over xGMI where the devices are linked),
 * are unpacked there into ONE host-layout C and copied back — no other exchange step.  Same arguments, layouts, status
 * codes and per-thread caching as qgemul_run (one context, plan and set of grow-only buffers per device of the list;
 * qgemul_run_release() frees them); opts->device is ignored.  A device may appear more than once in the list (two contexts
 * on one card: how the partition / reassembly logic is tested on a one-GPU box).  n <= 16.  Real and complex, any class.
 * (Multi-PROCESS sharding — one rank per GPU, the library-owned RCCL gather — is qgemul_comm_* below.) */
int qgemul_run_sharded(const qgemul_desc* d, void* C, const void* A, const void* B, const qgemul_opts* o, const int* devices, int n);

/* ---- one process per GPU: the library-owned RCCL gather (SURVEY.md 8-e, 8-b "library owns ... RCCL comms") ----
 * Rank r holds the packed operands of ITS band of rows (a plan for rows_r x N x K, B replicated), runs qgemul_execute on its
 * context, and the ONE exchange step of the path moves the packed C bands to the root rank, which unpacks each band at its row
 * offset (qgemul_unpack_c on a plan of the band's shape, destination pointer advanced by row0 elements, ld = the full M).
 * The reference has no counterpart (no communication of any kind).  RCCL is bound at the first of these calls (the process's
 * own librccl if one is loaded — a PyTorch process keeps ONE RCCL —, else librccl.so.1): single-GPU users never load it.
 *   qgemul_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId); the caller hands it to the other ranks by any means
 *                           it has (MPI, a file, a TCP store): the only rendezvous the library needs
 *   qgemul_comm_create      ncclCommInitRank on the context's device (collective: every rank calls it)
 *   qgemul_gather_packed_c  asynchronous.  The band was produced on the context's stream; it travels on the communicator's own
 *                           stream (grouped ncclSend / ncclRecv of bytes), so the next GEMM may run meanwhile.  On the root,
 *                           recv[r] / recv_bytes[r] = where rank r's band lands (recv[root] may equal `send`: no copy);
 *                           elsewhere they are ignored.  A shard cut into row chunks = one call per chunk.  `slot`
 *                           (0 .. QG_COMM_SLOTS-1) names the buffer generation the call reads / writes: double-buffered
 *                           callers alternate slots so that a fence waits only for the generation it is about to reuse.
 *   qgemul_comm_fence       the context's stream waits (on the device) for the gathers issued under `slot` (-1: all of
 *                           them): call it before work that overwrites a buffer a gather still reads / before the root unpacks
 *   qgemul_comm_sync        the host waits for them
 *   qgemul_comm_barrier / qgemul_comm_max_f64   every rank's streams drained / the maximum of one double over the ranks
 *                           (timing of a multi-rank run without any other communication library)
 * Errors: QG_ERCCL, qgemul_last_rccl_error() = the ncclResult_t (-1: no usable librccl). */
typedef struct qgemul_comm qgemul_comm;
#define QG_COMM_ID_BYTES 128
#define QG_COMM_SLOTS 4
int qgemul_comm_unique_id(void* id_out);
int qgemul_comm_create(qgemul_ctx* c, int nranks, int rank, const void* unique_id, qgemul_comm** out);
void qgemul_comm_destroy(qgemul_comm* m);
int qgemul_comm_info(const qgemul_comm* m, int* nranks, int* rank, int* rccl_version);   /* ncclCommCount, ncclCommUserRank, ncclGetVersion */
int qgemul_gather_packed_c(qgemul_comm* m, const void* send, size_t send_bytes, void* const* recv, const size_t* recv_bytes, int root, int slot);
int qgemul_comm_fence(qgemul_comm* m, int slot);
int qgemul_comm_sync(qgemul_comm* m);
int qgemul_comm_barrier(qgemul_comm* m);
int qgemul_comm_max_f64(qgemul_comm* m, double* inout);
int qgemul_last_rccl_error(void);
int qgemul_ctx_device(const qgemul_ctx* c);   /* the HIP device ordinal of a context */

/* ---- BitStream export of the result tensor ----
 * What  BitStream<tensorProcessT, elemProcessT>(C)  returns in the reference (QuBLAS.h:4811-4827; demo main.cpp:9-18):
 * per element the low (isS + intB + fracB) bits of its raw value, MSB first (:2433-2438), elements in storage order
 * (column-major), with the element string's chunks of `elem_chunk` characters reversed (r2l<elem_chunk>, :4593-4611)
 * and the tensor's chunks of `tensor_chunk` elements reversed (r2l<tensor_chunk>, :4738-4752); chunk 0 = l2r.
 * The element width must be a multiple of elem_chunk and M*N of tensor_chunk (the reference throws / does not terminate
 * otherwise): QG_EINVAL.
 * QG_BITS_ASCII writes the M*N*width characters '0' / '1'; QG_BITS_PACKED the same stream 8 characters per byte, first
 * character in bit 7.  For a plan with an epilogue the tensor is D.
 * Complex tensors: an element's string is "(" + real bits + ", " + imaginary bits + ")" (:2553-2556), width = wr + wi + 4
 * characters, and r2l<elem_chunk> reverses chunks of THAT string, punctuation included (:4672-4681), so elem_chunk must
 * divide wr + wi + 4.  QG_BITS_ASCII writes exactly these characters; QG_BITS_PACKED writes only the binary characters of
 * the stream, in stream order (what the reference's own reader keeps of it, :4779-4781): wr + wi bits per element. */
enum { QG_BITS_ASCII = 0, QG_BITS_PACKED = 1 };
int64_t qgemul_bitstream_bytes(const qgemul_plan* p, int format);
int qgemul_export_bitstream(qgemul_plan* p, const void* packedC, int tensor_chunk, int elem_chunk, int format, void* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* QGEMUL_H_ */
