"""GPU parity (run with -m gpu on an MI355X): the HIP engine, called through the C-ABI
(qgemul_run and the resident-data entry points), against
  * the golden vectors the real reference header produced (tests/golden/ref_gemm_*.jsonl.gz), and
  * the CPU restatement oracle/qoracle.c on seeded inputs at sizes it finishes in seconds.
Bit-exact: integer work, no tolerance.  Nothing here reads /root/reference."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import (CLASS_LINEAR, Qcomplex, Qu, RND, SAT, TRN, WRP, TFComplexMul, BasicComplexMul, Tags,
                             desc_from_dict, lower)

pytestmark = pytest.mark.gpu


def run_gpu(d, A, B, ec, oracle, flags=0, **ld):
    M, N = d.M, d.N
    ldc = ld.get("ldc", 0) or M
    out = np.zeros(ldc * N, dtype=oracle.host_dtype(ec))
    return capi.run(d, out, A, B, flags=flags, **ld)


def fields_equal(a, b):
    if a.dtype.names:
        return all(np.array_equal(a[n], b[n]) for n in a.dtype.names)
    return np.array_equal(a, b)


@pytest.mark.parametrize("j", G.gemm_cases("real") + G.gemm_cases("cplx"), ids=lambda j: j["name"])
def test_golden_vectors(oracle, j):
    d = desc_from_dict(j)
    A, B = G.case_inputs(j, oracle)
    _, _, ec = G.case_elems(j)
    got = run_gpu(d, A, B, ec, oracle)
    exp = G.case_expected(j, oracle)
    assert fields_equal(got, exp), j["name"]


@pytest.mark.parametrize("j", [j for j in G.gemm_cases("real") if "_L_" in j["name"] or j["name"].endswith("classL")],
                         ids=lambda j: j["name"])
def test_golden_vectors_linear_cases_on_tree_kernel(oracle, j):
    """The same golden cases forced through the exact tree kernel: both kernels must agree with the reference."""
    d = desc_from_dict(j)
    A, B = G.case_inputs(j, oracle)
    _, _, ec = G.case_elems(j)
    got = run_gpu(d, A, B, ec, oracle, flags=capi.OPT_FORCE_TREE)
    assert fields_equal(got, G.case_expected(j, oracle)), j["name"]


E43 = Qu(4, 3)
E88Z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
W16 = Qu(16, 3)


def _vs_oracle(oracle, ea, eb, ec, M, N, K, *, dist=0, flags=0, nthreads=8, expect_kernel=None, **kw):
    d = lower(ea, eb, ec, M, N, K, **kw)
    info = capi.classify(d, flags)
    if expect_kernel is not None:
        assert capi.KERNEL_NAMES[info.kernel] == expect_kernel, (capi.KERNEL_NAMES[info.kernel], info.reason)
    A = oracle.fill(ea, M * K, 1, dist)
    B = oracle.fill(eb, K * N, 2, dist)
    got = run_gpu(d, A, B, ec, oracle, flags=flags)
    exp = oracle.gemm(d, A, B, ec, nthreads=nthreads)
    assert fields_equal(got, exp)
    return got


def saturated_fraction(c, e: Qu):
    return float(np.mean((c == e.raw_max) | (c == e.raw_min)))


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("shape", [(128, 128, 128), (256, 384, 512), (130, 70, 200), (1, 1, 1), (257, 129, 64)])
def test_mfma_i8_vs_oracle(oracle, shape, ta):
    M, N, K = shape
    c = _vs_oracle(oracle, E43, E43, W16, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)], transposed_a=ta,
                   expect_kernel="mfma_i8")
    assert saturated_fraction(c, W16) < 0.5  # the comparison is not hidden by saturation


@pytest.mark.parametrize("shape", [(512, 512, 4096), (256, 256, 2048), (500, 130, 4096), (64, 1000, 8192), (512, 512, 1920), (1024, 1024, 2048)])
def test_small_single_limb_long_k(oracle, shape):
    """Long-k single-limb problems of at most one 64 x 64 tile per CU run with TWO k groups per workgroup (each its own LDS ring
    and half of the k-tiles, accumulators summed through LDS before the epilogue); 15 k-tiles (K = 1920) and 1024^2 (one
    workgroup per CU, but the 128-tile form) keep one group.  All against the oracle, narrow and wide C."""
    M, N, K = shape
    for ec in (W16, Qu(6, 1, True, RND.CONV, SAT.SMGN)):
        c = _vs_oracle(oracle, E43, E43, ec, M, N, K, dist=1, mul_args=Tags(9, 6), add_args=[Qu(22, 6)], expect_kernel="mfma_i8")
    assert len(np.unique(c)) > 8


@pytest.mark.parametrize("qm", [RND.POS_INF, RND.NEG_INF, RND.ZERO, RND.INF, RND.CONV, TRN.TCPL, TRN.SMGN])
@pytest.mark.parametrize("om", [SAT.TCPL, SAT.ZERO, SAT.SMGN, WRP.TCPL])
def test_mfma_epilogue_modes(oracle, qm, om):
    # narrow C with few frac bits: the epilogue rounds and overflows for real
    _vs_oracle(oracle, E43, E43, Qu(9, 1, True, qm, om), 128, 128, 256, mul_args=Tags(9, 6), add_args=[Qu(19, 6)],
               expect_kernel="mfma_i8")
    _vs_oracle(oracle, E43, E43, Qu(6, 2, False, qm, om), 64, 64, 128, dist=1, mul_args=Tags(9, 6), add_args=[Qu(19, 6)],
               expect_kernel="mfma_i8")


def test_config2_1024_linear(oracle):
    """BASELINE.json configuration 2: 1024^3 int<4,3> on MFMA_I32_I8, bit-exact vs the CPU restatement."""
    c = _vs_oracle(oracle, E43, E43, E43, 1024, 1024, 1024, mul_args=Tags(9, 6), add_args=[Qu(19, 6)],
                   expect_kernel="mfma_i8")
    c2 = _vs_oracle(oracle, E43, E43, W16, 1024, 1024, 1024, mul_args=Tags(9, 6), add_args=[Qu(19, 6)],
                    expect_kernel="mfma_i8")
    assert saturated_fraction(c, E43) > 0.9       # documented: full-range inputs saturate a narrow C
    assert saturated_fraction(c2, W16) < 0.6


@pytest.mark.parametrize("ta", [False, True])
def test_limb_mfma_int88_vs_oracle(oracle, ta):
    """int<8,8> operands (17 storage bits) -> 3x3 int8 limbs on MFMA, 64-bit recombination."""
    wide = Qu(24, 8, True, RND.CONV, SAT.SMGN)
    _vs_oracle(oracle, E88Z, E88Z, wide, 256, 128, 512, mul_args=Tags(17, 16), add_args=[Qu(29, 16)], transposed_a=ta,
               expect_kernel="mfma_i8_limb")
    _vs_oracle(oracle, E88Z, E88Z, E88Z, 130, 100, 192, dist=1, mul_args=Tags(17, 16), add_args=[Qu(29, 16)],
               transposed_a=ta, expect_kernel="mfma_i8_limb")


def test_limb_mfma_mixed_widths(oracle):
    u44 = Qu(4, 4, False)
    # (an unsigned 8-bit format takes two plain balanced limbs; centred — x - 128, tests/test_gpu_centred.py — one)
    _vs_oracle(oracle, u44, u44, Qu(14, 8, False), 128, 128, 256, mul_args=Tags(8, 8), add_args=[Qu(18, 8, False)],
               expect_kernel="mfma_i8")
    _vs_oracle(oracle, u44, u44, Qu(14, 8, False), 128, 128, 256, mul_args=Tags(8, 8), add_args=[Qu(18, 8, False)], flags=capi.OPT_BALANCED_LIMBS,
               expect_kernel="mfma_i8_limb")
    _vs_oracle(oracle, E88Z, E43, Qu(20, 11), 128, 192, 128, mul_args=Tags(13, 11), add_args=[Qu(22, 11)],
               expect_kernel="mfma_i8_limb")
    _vs_oracle(oracle, E43, Qu(7, 5), Qu(20, 8), 192, 128, 128, mul_args=Tags(12, 8), add_args=[Qu(21, 8)],
               expect_kernel="mfma_i8_limb")


@pytest.mark.parametrize("K", [1, 2, 3, 5, 37, 64, 100, 1000, 1024])
def test_tree_kernel_any_k(oracle, K):
    t1 = Qu(6, 5, True, RND.CONV, SAT.SMGN)
    t2 = Qu(8, 4, True, RND.ZERO, SAT.TCPL)
    _vs_oracle(oracle, E43, E43, W16, 33, 17, K, add_args=[t1, t2], mul_args=Qu(5, 4, True, RND.INF, SAT.TCPL))
    _vs_oracle(oracle, E88Z, E88Z, E88Z, 20, 9, K, dist=1)


def test_config1_and_config3_semantics_tree(oracle):
    """Configuration 1/3 semantics (int<8,8>, TCPL, SAT::ZERO, default tags: tree class) at a size the
    oracle finishes quickly; the 4x4x4 known answer itself is in the golden set."""
    c = _vs_oracle(oracle, E88Z, E88Z, E88Z, 96, 80, 4096, dist=1, expect_kernel="tree_i32")
    assert float(np.mean(c == 0)) < 0.9
    c2 = _vs_oracle(oracle, E88Z, E88Z, E88Z, 96, 80, 4096, dist=1, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_i64")
    assert np.array_equal(c, c2)
    _vs_oracle(oracle, E88Z, E88Z, E88Z, 70, 50, 4096, dist=0, expect_kernel="tree_i32")


FAST_TREE_CASES = [
    # (A, B, C, mul tags, level list, K, dist, what)
    (Qu(8, 8, True, TRN.TCPL, SAT.ZERO), Qu(8, 8, True, TRN.TCPL, SAT.ZERO), Qu(8, 8, True, TRN.TCPL, SAT.ZERO), None, None, 64, 1, "split mul24 TCPL/ZERO"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 10), Tags(fracBits=10, QuMode=RND.CONV), [Qu(12, 10, True, RND.CONV, SAT.SMGN)], 256, 1, "split mul24 CONV"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 6), Tags(fracBits=10, QuMode=RND.POS_INF, OfMode=SAT.ZERO), [Qu(10, 8, True, RND.NEG_INF), Qu(12, 6, True, RND.ZERO)], 128, 1, "split POS_INF"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 6), Tags(fracBits=10, QuMode=RND.INF), [Qu(10, 9, True, RND.INF, WRP.TCPL)], 64, 0, "split INF + wrap levels"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 6), Tags(fracBits=10, QuMode=RND.ZERO), [Qu(14, 10, True, TRN.SMGN)], 64, 1, "split ZERO"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 6), Tags(fracBits=10, QuMode=TRN.SMGN), None, 64, 1, "split SMGN"),
    (Qu(8, 10), Qu(8, 10), Qu(12, 6), Tags(fracBits=10, QuMode=RND.NEG_INF), None, 64, 1, "split NEG_INF"),
    (Qu(12, 12), Qu(6, 5), Qu(16, 8), Tags(fracBits=11), [Qu(16, 11)], 64, 1, "split without mul24"),
    (Qu(4, 3), Qu(4, 3), Qu(16, 3), None, None, 1024, 1, "direct mul24 default tags"),
    (Qu(4, 3), Qu(4, 3), Qu(16, 3), Qu(5, 4, True, RND.INF, SAT.TCPL), [Qu(6, 5, True, RND.CONV, SAT.SMGN), Qu(8, 4, True, RND.ZERO, SAT.TCPL)], 512, 0, "direct, per-level list"),
    (Qu(4, 4, False), Qu(4, 4, False), Qu(14, 4, False), None, None, 64, 0, "unsigned direct"),
    (Qu(4, 4, False), Qu(4, 3), Qu(14, 4), Tags(OfMode=WRP.TCPL, intBits=6), [Qu(7, 4, False, TRN.TCPL, WRP.TCPL)], 64, 0, "unsigned wrap levels"),
    (Qu(13, 12), Qu(3, 1), Qu(16, 8), Tags(fracBits=12, QuMode=RND.CONV), [Qu(15, 12)], 32, 1, "direct without mul24"),
    (Qu(6, -3), Qu(6, -3), Qu(16, -3), Tags(FullPrec=True), [Qu(20, -6)], 64, 0, "negative fracBits, FullPrec"),
]


@pytest.mark.parametrize("case", FAST_TREE_CASES, ids=lambda c: c[7])
def test_fast_tree_kernel_modes(oracle, case):
    ea, eb, ec, mul, levels, K, dist, _ = case
    kw = dict(mul_args=mul, add_args=levels)
    a = _vs_oracle(oracle, ea, eb, ec, 70, 37, K, dist=dist, expect_kernel="tree_i32", **kw)
    b = _vs_oracle(oracle, ea, eb, ec, 70, 37, K, dist=dist, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_i64", **kw)
    assert np.array_equal(a, b)
    a = _vs_oracle(oracle, ea, eb, ec, 33, 65, K, dist=0, expect_kernel="tree_i32", transposed_a=True, **kw)


def test_config5_complex_tf(oracle):
    r = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
    i = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r, i)
    wide = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    _vs_oracle(oracle, c5, c5, c5, 64, 48, 2048, mul_args=TFComplexMul(), expect_kernel="tree_cplx_i32")
    a = _vs_oracle(oracle, c5, c5, wide, 64, 48, 2048, dist=1, mul_args=TFComplexMul(), expect_kernel="tree_cplx_i32")
    b = _vs_oracle(oracle, c5, c5, wide, 64, 48, 2048, dist=1, mul_args=TFComplexMul(), flags=capi.OPT_GENERIC_TREE,
                   expect_kernel="tree_cplx")
    assert fields_equal(a, b)
    _vs_oracle(oracle, c5, c5, wide, 40, 24, 100, dist=1, mul_args=BasicComplexMul(), transposed_a=True, expect_kernel="tree_cplx_i32")   # K = 100: zero-padded to 128 leaves
    _vs_oracle(oracle, c5, c5, wide, 40, 24, 13, dist=1, mul_args=BasicComplexMul(), transposed_a=True, expect_kernel="tree_cplx_i32")  # 4 levels: continued with an identity level
    _vs_oracle(oracle, c5, c5, wide, 70, 33, 256, dist=0, mul_args=BasicComplexMul(), transposed_a=True, expect_kernel="tree_cplx_i32")


def test_complex_fixed_mode_variant(oracle):
    """BASELINE configuration 5's formats ("RND + SAT": RND::POS_INF + SAT::TCPL everywhere) take the fixed-mode variant of
    the complex kernel ((v + 2^(d-1)) >> d, one clamp); the runtime-mode variant (QG_OPT_RUNTIME_MODES), the generic
    64-bit kernel and the oracle must all give the same matrix — narrow and wide C, TF and Basic, level types too."""
    r = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
    i = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r, i)
    wide = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    lv = Qcomplex(Qu(9, 2, True, RND.POS_INF, SAT.TCPL), Qu(10, 0, True, RND.POS_INF, SAT.TCPL))
    for ec, dist, kw in ((c5, 0, dict(mul_args=TFComplexMul())), (wide, 1, dict(mul_args=TFComplexMul())),
                         (wide, 1, dict(mul_args=BasicComplexMul())), (wide, 0, dict(mul_args=TFComplexMul(), add_args=[lv])),
                         (c5, 1, dict(mul_args=BasicComplexMul(), add_args=[lv, lv]))):
        a = _vs_oracle(oracle, c5, c5, ec, 70, 37, 512, dist=dist, expect_kernel="tree_cplx_i32", **kw)
        b = _vs_oracle(oracle, c5, c5, ec, 70, 37, 512, dist=dist, flags=capi.OPT_RUNTIME_MODES, expect_kernel="tree_cplx_i32", **kw)
        c = _vs_oracle(oracle, c5, c5, ec, 70, 37, 512, dist=dist, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_cplx", **kw)
        assert fields_equal(a, b) and fields_equal(a, c)


def test_complex_fast_kernel_tags_and_levels(oracle):
    """Complex 32-bit kernel with per-sub-op tags (incl. the crossed cdbT/badT use) and complex level types."""
    r55 = Qu(5, 5)
    c55 = Qcomplex(r55, r55)
    wide = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    tA = Qu(9, 4, True, RND.CONV, SAT.SMGN)
    tB = Qu(7, 2, True, TRN.SMGN, SAT.ZERO)
    tC = Qu(10, 5, True, RND.ZERO, WRP.TCPL)
    tD = Qu(8, 3, True, RND.INF, SAT.TCPL)
    tf = TFComplexMul(abT=tA, cdT=tD, abcT=tC, cdbT=tB, badT=tA, ABT=tD, BCT=tC)
    l1 = Qcomplex(Qu(12, 4, True, RND.CONV, SAT.SMGN), Qu(11, 6, True, TRN.SMGN, SAT.ZERO))
    l2 = Qcomplex(Qu(16, 2, True, RND.ZERO), Qu(16, 3, True, RND.INF, WRP.TCPL))
    for kw in (dict(mul_args=tf), dict(mul_args=tf, add_args=[l1, l2]), dict(mul_args=TFComplexMul(), add_args=[l1]),
               dict(mul_args=BasicComplexMul(acT=tA, bdT=tB, adT=tC, bcT=tD, acbdT=tC, adbcT=tA), add_args=[l2, l1])):
        a = _vs_oracle(oracle, c55, c55, wide, 37, 70, 64, expect_kernel="tree_cplx_i32", **kw)
        b = _vs_oracle(oracle, c55, c55, wide, 37, 70, 64, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_cplx", **kw)
        assert fields_equal(a, b)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 7, 8, 12, 16])
def test_short_trees_on_the_register_counter_kernels(oracle, K):
    """K <= 16 (fewer than 5 tree levels) used to fall to the general tree kernel (4096 x 4096 x 16: 1.02 ms where K = 32 takes
    0.11 ms).  The planner now continues a short tree with identity levels — x + 0 in x's own format — over operands
    zero-padded to 32 leaves, so the 32-bit / 64-bit / complex register-counter kernels take them, in every step form."""
    e88, e88z = Qu(8, 8), Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    cases = [
        (e88, e88, dict(), "tree_i32"), (e88z, e88z, dict(), "tree_i32"),
        (e88, Qu(12, 8), dict(add_args=[Qu(12, 8)]), "tree_i32"),
        (e88z, Qu(12, 6), dict(add_args=[Qu(10, 8, True, TRN.TCPL, SAT.ZERO), Qu(12, 6)]), "tree_i32"),
        (e88, Qu(12, 8), dict(add_args=[Qu(12, 8, True, RND.CONV)], mul_args=Qu(10, 6, True, RND.CONV)), "tree_i32"),
        (Qu(15, 16), Qu(20, 12), dict(add_args=[Qu(24, 16)]), "tree_i64"),
    ]
    for e, ec, kw, kern in cases:
        d = lower(e, e, ec, 70, 41, K, **kw)
        assert capi.KERNEL_NAMES[capi.classify(d).kernel] == kern
        a = _vs_oracle(oracle, e, e, ec, 70, 41, K, expect_kernel=kern, **kw)
        b = _vs_oracle(oracle, e, e, ec, 70, 41, K, flags=capi.OPT_GENERIC_TREE, **kw)
        assert np.array_equal(a, b)
    P = lambda i, f: Qu(i, f, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(P(6, 3), P(6, -3))
    for mul in (TFComplexMul(), BasicComplexMul()):
        for flags in (0, capi.OPT_RUNTIME_MODES):
            _vs_oracle(oracle, c5, c5, c5, 33, 29, K, flags=flags, mul_args=mul, expect_kernel="tree_cplx_i32")


def test_32_bit_fixed_point_words(oracle):
    """Operands of 32 storage bits (Q15.16 and friends): the unrounded product of two of them needs all 64 bits of an int64
    (|a * b| <= 2^62), which the planner admits for the ONE exact 64-bit multiply that feeds its rounding shift, while every
    other intermediate stays within 62 bits.  Real (64-bit tree kernel), one column, complex (general kernel), full-range
    operands incl. the raw minimum, several modes."""
    q = Qu(15, 16)
    cases = [
        (q, q, q, dict(), "tree_i32", "gemv_i32"),                                                                     # default tags: every product and node saturates (the 32-bit-word form)
        (q, q, Qu(24, 16, True, RND.CONV, SAT.SMGN), dict(add_args=[Qu(24, 16), Qu(28, 12, True, RND.ZERO, SAT.ZERO)]), "tree_i64", "gemv_i64"),
        (Qu(20, 11, True, TRN.SMGN, WRP.TCPL), Qu(3, 28), Qu(20, 11), dict(mul_args=Tags(20, 11)), "tree_i32", "gemv_i32"),   # product and levels Qu<20,11>: a 32-bit word too (product shift 28)
        (q, Qu(15, 16, False), Qu(18, 13), dict(), "tree_i64", "gemv_i32"),                                          # signed 32 x unsigned 31 bits
    ]
    for ea, eb, ec, kw, kern, col in cases:      # (col: the one-column kernel; 32-bit words there: tests/test_gpu_gemv.py)
        for M, N, K, ta in ((37, 29, 64, False), (16, 40, 300, True), (50, 1, 1024, False)):
            d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
            # (one column: the one-column kernels — 32-bit words there too when the tree kernel has them, tests/test_gpu_gemv.py)
            assert capi.KERNEL_NAMES[capi.classify(d).kernel] == (col if N == 1 else kern)
            A, B = oracle.fill(ea, M * K, 3, 0), oracle.fill(eb, K * N, 4, 0)
            A[:2] = ea.raw_min
            B[:2] = eb.raw_min                                                                            # (-2^31) * (-2^31) = 2^62 is present
            got = run_gpu(d, A, B, ec, oracle)
            assert fields_equal(got, oracle.gemm(d, A, B, ec, nthreads=8)), (str(ea), str(eb), M, N, K)
            if kern == "tree_i32" and N > 1:     # ... and the 64-bit tree kernel on the same descriptor
                assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_RUNTIME_MODES).kernel] == "tree_i64"
                assert fields_equal(run_gpu(d, A, B, ec, oracle, flags=capi.OPT_RUNTIME_MODES), got)
    cq = Qcomplex(q, q)
    for mul in (BasicComplexMul(), TFComplexMul()):
        d = lower(cq, cq, cq, 21, 17, 64, mul_args=mul)
        assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "tree_cplx"
        A, B = oracle.fill(cq, 21 * 64, 5, 0), oracle.fill(cq, 64 * 17, 6, 0)
        assert fields_equal(run_gpu(d, A, B, cq, oracle), oracle.gemm(d, A, B, cq, nthreads=8))
    # two UNSIGNED 32-bit words: the product needs 64 magnitude bits — the 128-bit tree kernel (round 2 refused it; tests/test_wide.py)
    u = Qu(16, 16, False)
    d = lower(u, u, u, 8, 8, 64)
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "tree_i128"
    A, B = oracle.fill(u, 8 * 64, 7, 0), oracle.fill(u, 64 * 8, 8, 0)
    assert fields_equal(run_gpu(d, A, B, u, oracle), oracle.gemm(d, A, B, u, nthreads=4))


def test_real_tree_kernel_step_forms(oracle):
    """The 32-bit tree kernel's step forms (the planner reports the form in info.reason): one format everywhere (default
    tags), per-level formats in the compact form of qg_fix.h — roundings that add a constant and shift (TRN::TCPL,
    RND::POS_INF, RND::NEG_INF), overflows that clamp (SAT::TCPL, SAT::SMGN), test the range (SAT::ZERO) or wrap (WRP::TCPL),
    split and direct products, level types with fewer and with more fraction bits — and run-time modes for the rest
    (RND::CONV here).  Every form must equal the oracle and the run-time-mode kernel."""
    e88 = Qu(8, 8)
    e88z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    cases = [
        (e88, e88, dict(), "one format, SAT::TCPL, left-justified"),                                                  # saturating v_mad / v_add on x * 2^15
        (Qu(4, 3), Qu(4, 3), dict(), "one format, SAT::TCPL, left-justified, packed 16-bit"),                         # configuration 2 as literally configured: v_pk_mad_i16 / v_pk_add_i16 ... clamp
        (Qu(4, 3), Qu(9, 2), dict(mul_args=Qu(5, 4, True, RND.POS_INF, SAT.TCPL), add_args=[Qu(5, 4)]), "one format, SAT::TCPL, left-justified, packed 16-bit"),
        (Qu(3, 2, False), Qu(9, 2), dict(mul_args=Qu(4, 2, True, RND.NEG_INF, SAT.TCPL), add_args=[Qu(4, 2)]), "one format, SAT::TCPL, left-justified, packed 16-bit"),   # unsigned operands in a signed format
        (e88, Qu(12, 8), dict(mul_args=Qu(8, 8, True, RND.POS_INF, SAT.TCPL), add_args=[Qu(8, 8)]), "one format, SAT::TCPL, left-justified"),   # the addend rides in the multiply-add
        (Qu(6, 5), Qu(9, 2), dict(mul_args=Qu(7, 6, True, RND.NEG_INF, SAT.TCPL), add_args=[Qu(7, 6)]), "one format, SAT::TCPL, left-justified, packed nodes"),   # 14 bits, product shift 4: 32-bit products, packed nodes
        (Qu(7, 8), Qu(7, 8), dict(), "one format, SAT::TCPL, left-justified, packed nodes"),                         # 16-bit words: the high half of the justified product IS the value
        (Qu(7, 8, True, RND.POS_INF, SAT.TCPL), Qu(12, 4), dict(), "one format, SAT::TCPL, left-justified, packed nodes"),
        # unsigned operands and formats: [0, 2^W - 1] is the uint32 range of x * 2^(32 - W): v_mad_u32_u24 / v_add_u32 / v_pk_*_u16 with the clamp bit
        (Qu(8, 8, False), Qu(8, 8, False), dict(), "one format, SAT::TCPL, left-justified, packed nodes"),
        (Qu(8, 0, False), Qu(8, 0, False), dict(), "one format, SAT::TCPL, left-justified, packed 16-bit"),
        (Qu(12, 5, False), Qu(12, 5, False), dict(), "one format, SAT::TCPL, left-justified"),
        (Qu(5, 4, False), Qu(9, 2, False), dict(mul_args=Qu(7, 4, False, RND.POS_INF, SAT.TCPL), add_args=[Qu(7, 4, False)]), "one format, SAT::TCPL, left-justified, packed 16-bit"),
        (Qu(7, 6, False), Qu(9, 2, False), dict(mul_args=Qu(8, 6, False, RND.NEG_INF, SAT.TCPL), add_args=[Qu(8, 6, False)]), "one format, SAT::TCPL, left-justified, packed nodes"),
        (Qu(8, 0, False), Qu(8, 0, False), dict(mul_args=Qu(8, 0, True), add_args=[Qu(8, 0, True)]), "one format, SAT::TCPL, left-justified, packed 16-bit"),   # unsigned operands in a signed format: the signed form
        (e88z, e88z, dict(), "one format, SAT::ZERO"),
        (e88, Qu(12, 8), dict(add_args=[Qu(12, 8)]), "per-level formats, compact (clamps)"),                          # a wider level type (split product)
        (e88z, Qu(12, 6, True, TRN.TCPL, SAT.ZERO), dict(add_args=[Qu(10, 8, True, TRN.TCPL, SAT.ZERO), Qu(12, 6, True, TRN.TCPL, SAT.ZERO)]),
         "per-level formats, compact"),                                                                               # SAT::ZERO with per-level formats
        (Qu(4, 3), Qu(9, 2, True, RND.NEG_INF, SAT.SMGN), dict(mul_args=Qu(6, 5, True, RND.POS_INF, SAT.TCPL),
                                                                add_args=[Qu(8, 3, True, RND.POS_INF, SAT.TCPL), Qu(9, 2, True, RND.NEG_INF, SAT.SMGN)]),
         "per-level formats, compact (clamps)"),                                                                      # direct product, rounding nodes
        (Qu(4, 3), Qu(8, 7), dict(mul_args=Tags(6, 7), add_args=[Qu(7, 7), Qu(8, 9, True, TRN.TCPL, WRP.TCPL)]), "per-level formats, compact"),   # left shifts, a wrapping level
        (e88z, Qu(20, 9, True, TRN.TCPL, SAT.ZERO), dict(add_args=[Qu(20, 9, True, TRN.TCPL, SAT.ZERO)]), "per-level formats, compact (unbiased)"),   # 30 bits: too wide to bias
        # the README's own call (readme.md:28-36, :84-87): product and level 0 int<6,3> SAT::ZERO, later levels int<6,-3> with default modes
        (Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, 3, True, TRN.TCPL, SAT.ZERO),
         dict(mul_args=Qu(6, 3, True, TRN.TCPL, SAT.ZERO), add_args=[Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)], transposed_a=True), "per-level formats, compact"),
        # roundings that look at the value's sign or parity: rounding kinds of the unbiased form (split and direct products)
        (e88, Qu(12, 8), dict(add_args=[Qu(12, 8, True, RND.CONV)], mul_args=Qu(10, 6, True, RND.CONV)), "per-level formats, compact (unbiased)"),
        (Qu(4, 3), Qu(8, 1, True, RND.ZERO, SAT.ZERO), dict(mul_args=Qu(6, 4, True, RND.INF, SAT.SMGN), add_args=[Qu(8, 3, True, TRN.SMGN, SAT.TCPL), Qu(8, 1, True, RND.ZERO, SAT.ZERO)]),
         "per-level formats, compact (unbiased)"),
    ]
    for K in (64, 300):
        for e, ec, kw, form in cases:
            M, N = 70, 41
            d = lower(e, e, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_i32" and info.reason.decode().endswith(form), (info.reason, form)
            a = _vs_oracle(oracle, e, e, ec, M, N, K, expect_kernel="tree_i32", **kw)
            b = _vs_oracle(oracle, e, e, ec, M, N, K, flags=capi.OPT_RUNTIME_MODES, expect_kernel="tree_i32", **kw)
            assert np.array_equal(a, b)


def test_reference_artefacts_flag_on_the_device(oracle):
    """QG_DESC_REFERENCE_ARTEFACTS: C of an unsigned WRP::TCPL format with exactly 32 value bits holds the unwrapped value, as the
    reference stores it (tests/golden/ref_scalar_6: the reference's own outputs) — as a K = 1 Qgemul on the device, and as a
    GEMM with a real reduction against the oracle."""
    import golden_io as G
    from qublas_amd.desc import ONE
    seen = 0
    for t in G.scalar_tables(6):
        src, dst = Qu.from_tuple(t["from"]), Qu.from_tuple(t["to"])
        if dst.isSigned or dst.intBits + dst.fracBits != 32:
            continue
        xs = np.arange(t["lo"], t["hi"] + 1, t["step"], dtype=np.int64)
        d = lower(src, ONE, dst, len(xs), 1, 1, mul_args=src, reference_artefacts=True)
        got = run_gpu(d, xs.astype(oracle.host_dtype(src)), np.ones(1, np.int32), dst, oracle)
        assert [int(v) for v in got] == t["y"], (t["from"], t["to"])
        seen += 1
    assert seen >= 3
    u32 = Qu(32, 0, False, TRN.TCPL, WRP.TCPL)
    e = Qu(12, 4)
    for M, N, K in ((33, 17, 64), (128, 128, 300)):
        d = lower(e, e, u32, M, N, K, mul_args=Tags(25, 8), add_args=[Qu(40, 8)], reference_artefacts=True)
        A, B = oracle.fill(e, M * K, 5), oracle.fill(e, K * N, 6)
        got = run_gpu(d, A, B, u32, oracle)
        exp = oracle.gemm(d, A, B, u32, nthreads=8)
        assert np.array_equal(got, exp) and (exp < 0).any()       # (negative sums stay negative: unwrapped)
        assert capi.classify_status(lower(e, e, u32, M, N, K, mul_args=Tags(25, 8), add_args=[Qu(40, 8)]))[0] == capi.QG_EUNSUPPORTED


@pytest.mark.parametrize("K", [1, 2, 33, 100, 1000, 4096])
def test_wrapping_32_bit_word_tree_form(oracle, K):
    """A signed WRP::TCPL format of exactly 32 bits with default tags — what `(int32_t)(((int64_t)a * b) >> 16)` and a plain `+=`
    compute: the product's word is the low 32 bits of the shifted exact product (`v_alignbit_b32`), a node a plain 32-bit add
    (`k_tree_fast<., 21>`).  Full-range operands (everything wraps) and small ones, truncating and rounding products, shifts
    16 / 31 / 4 / 28, against the oracle and against the 64-bit tree kernel."""
    w = Qu(15, 16, True, TRN.TCPL, WRP.TCPL)
    w31 = Qu(0, 31, True, RND.NEG_INF, WRP.TCPL)
    cases = [(w, w, w, {}), (w31, w31, w31, {}),
             (Qu(15, 16), Qu(15, 16), Qu(20, 4), dict(mul_args=Qu(15, 16, True, RND.POS_INF, WRP.TCPL), add_args=[w])),
             (Qu(8, 12), Qu(4, 8), w, dict(mul_args=w, add_args=[w])),
             (Qu(20, 11, True, TRN.TCPL, WRP.TCPL), Qu(3, 28), Qu(20, 11, True, TRN.TCPL, WRP.TCPL), dict(mul_args=Qu(20, 11, True, TRN.TCPL, WRP.TCPL)))]
    for ea, eb, ec, kw in cases:
        for M, N in ((33, 17), (1, 3), (70, 41)):
            d = lower(ea, eb, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_i32" and info.reason.decode().endswith("wrapping word adds"), (str(ea), info.reason)
            assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_RUNTIME_MODES).kernel] == "tree_i64"
            for dist in (0, 1, 2):
                A, B = oracle.fill(ea, M * K, 5, dist % 2), oracle.fill(eb, K * N, 6, dist % 2)
                if dist == 2:
                    A, B = (A >> 9).astype(A.dtype), (B >> 9).astype(B.dtype)
                got = run_gpu(d, A, B, ec, oracle)
                exp = oracle.gemm(d, A, B, ec, nthreads=8)
                assert np.array_equal(got, exp), (str(ea), M, N, K, dist)
                assert np.array_equal(run_gpu(d, A, B, ec, oracle, flags=capi.OPT_RUNTIME_MODES), exp)


@pytest.mark.parametrize("K", [1, 5, 33, 100, 1000, 4096])
def test_justified_word_tree_form(oracle, K):
    """Default tags on words of fewer than 32 bits whose product needs a net right shift (Q11.12, the 24-bit words of much
    fixed-point code; Q13.12; Q14.16; Q3.20): held as x * 2^(32 - bits), the product's word out of the exact 64-bit product (compare
    form for net shifts 1 ... 9 and 24 ... 31, one saturating multiply-add for 10 ... 23), low bits cleared, a node one saturating
    add (`k_tree_fast<., 19 / 20>`).  They ran on the 64-bit tree kernel, which is the second opinion here."""
    cases = [(Qu(11, 12), Qu(11, 12), Qu(11, 12), {}), (Qu(13, 12), Qu(13, 12), Qu(13, 12), {}), (Qu(14, 16), Qu(14, 16), Qu(14, 16), {}),
             (Qu(3, 20), Qu(3, 20), Qu(3, 20), {}), (Qu(1, 29), Qu(1, 29), Qu(1, 29), {}),
             (Qu(11, 12, True, RND.POS_INF, SAT.TCPL), Qu(11, 12, True, RND.POS_INF, SAT.TCPL), Qu(8, 4, True, RND.CONV, SAT.SMGN), {}),
             (Qu(15, 16), Qu(7, 8), Qu(11, 12), dict(mul_args=Qu(11, 12, True, RND.NEG_INF, SAT.TCPL), add_args=[Qu(11, 12)])),
             (Qu(15, 15, False), Qu(14, 16), Qu(14, 16), dict(mul_args=Qu(14, 16), add_args=[Qu(14, 16)]))]
    for ea, eb, ec, kw in cases:
        for M, N in ((33, 17), (1, 3), (70, 41)):
            d = lower(ea, eb, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_i32" and info.reason.decode().endswith("justified words"), (str(ea), info.reason)
            assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_RUNTIME_MODES).kernel] == "tree_i64"
            for dist in (0, 1, 2):
                A, B = oracle.fill(ea, M * K, 5, dist % 2), oracle.fill(eb, K * N, 6, dist % 2)
                if dist == 2:
                    A, B = (A >> 7).astype(A.dtype), (B >> 7).astype(B.dtype)
                got = run_gpu(d, A, B, ec, oracle)
                exp = oracle.gemm(d, A, B, ec, nthreads=8)
                assert np.array_equal(got, exp), (str(ea), M, N, K, dist)
                assert np.array_equal(run_gpu(d, A, B, ec, oracle, flags=capi.OPT_RUNTIME_MODES), exp)


@pytest.mark.parametrize("K", [1, 2, 5, 32, 33, 100, 1000, 4096])
def test_32_bit_word_tree_form(oracle, K):
    """Q15.16 with default tags — every product and every tree node quantised into the 32-bit word — and relatives: the product
    from the exact 64-bit product (truncating and rounding shifts, operands of other widths), a node one saturating 32-bit add
    (`fast_mode` 10 on the 32-bit tree kernel's frame).  Against the oracle, with full-range operands (the sums saturate) and
    small ones (they do not), and against the 64-bit tree kernel the same descriptor takes with QG_OPT_RUNTIME_MODES."""
    q = Qu(15, 16)
    cases = [(q, q, q, {}), (Qu(15, 16, True, RND.POS_INF, SAT.TCPL), Qu(15, 16, True, RND.POS_INF, SAT.TCPL), Qu(20, 4), {}),
             (Qu(8, 12), Qu(4, 8), q, dict(mul_args=Qu(15, 16, True, RND.NEG_INF, SAT.TCPL), add_args=[q])),
             (Qu(20, 11), Qu(20, 11), Qu(20, 11), {}), (Qu(2, 29), Qu(2, 29), Qu(9, 3, True, RND.CONV, SAT.SMGN), {}),
             (Qu(15, 15, False), q, q, dict(mul_args=q, add_args=[q])),
             (Qu(0, 31), Qu(0, 31), Qu(0, 31), {}), (Qu(30, 1), Qu(30, 1), Qu(30, 1), {}),            # (Q31: product shift 31; shift 1)
             # shifts 10 ... 23 take the product's word from one saturating multiply-add (k_tree_fast<., 18>): both ends, and 24 beside them
             (Qu(21, 10), Qu(21, 10), Qu(21, 10), {}), (Qu(8, 23), Qu(8, 23), Qu(8, 23), {}), (Qu(7, 24), Qu(7, 24), Qu(7, 24), {}),
             (Qu(8, 23, True, RND.POS_INF, SAT.TCPL), Qu(8, 23), Qu(8, 23, True, RND.NEG_INF, SAT.TCPL), {})]
    # integers (no shift at all) are not this form: the 64-bit tree kernel
    i20, i24, i31 = Qu(20, 0), Qu(24, 0), Qu(31, 0)
    d = lower(i20, i24, i31, 33, 17, K, mul_args=i31, add_args=[i31])
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "tree_i64"
    A, B = oracle.fill(i20, 33 * K, 5, 0), oracle.fill(i24, K * 17, 6, 0)
    assert np.array_equal(run_gpu(d, A, B, i31, oracle), oracle.gemm(d, A, B, i31, nthreads=8))
    assert capi.KERNEL_NAMES[capi.classify(lower(q, q, Qu(20, 12), 33, 17, K)).kernel] == "tree_i64"     # a C beyond 32 bits: the 64-bit kernel's conversion
    for ea, eb, ec, kw in cases:
        for M, N in ((33, 17), (1, 3), (70, 41)):
            d = lower(ea, eb, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_i32" and info.reason.decode().endswith("saturating word adds"), (str(ea), info.reason)
            assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_RUNTIME_MODES).kernel] == "tree_i64"
            for dist in (0, 1, 2):
                A, B = oracle.fill(ea, M * K, 5, dist % 2), oracle.fill(eb, K * N, 6, dist % 2)
                if dist == 2:
                    A, B = (A >> 9).astype(A.dtype), (B >> 9).astype(B.dtype)
                got = run_gpu(d, A, B, ec, oracle)
                exp = oracle.gemm(d, A, B, ec, nthreads=8)
                assert np.array_equal(got, exp), (str(ea), M, N, K, dist)
                assert np.array_equal(run_gpu(d, A, B, ec, oracle, flags=capi.OPT_RUNTIME_MODES), exp)


@pytest.mark.parametrize("K", [1, 2, 3, 5, 16, 17, 31, 32, 33, 100, 1000])
def test_justified_forms_edge_shapes(oracle, K):
    """The left-justified and packed 16-bit forms (real and complex) on the shapes the fuzzers do not draw: K below one k-chunk and
    below two leaves (trees of 0 ... 10 levels, padded to 32 leaves), single rows and columns, a second column of the lane's pair
    that does not exist (N = 17: column 16 alone in its pair), both distributions (dist 1 saturates most nodes)."""
    P = lambda i, f: Qu(i, f, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(P(6, 3), P(6, -3))
    cases = [(Qu(4, 3), Qu(4, 3), {}, "tree_i32", "packed 16-bit"),
             (Qu(8, 8), Qu(8, 8), {}, "tree_i32", "left-justified"),
             (Qu(7, 8), Qu(7, 8), {}, "tree_i32", "packed nodes"),
             (Qu(8, 0, False), Qu(8, 0, False), {}, "tree_i32", "packed 16-bit"),
             (Qu(8, 8, False), Qu(12, 4, False), {}, "tree_i32", "packed nodes"),
             (Qu(12, 5, False), Qu(12, 5, False), {}, "tree_i32", "left-justified"),
             (Qu(5, 6), Qu(9, 2), {}, "tree_i32", "packed nodes"),
             (c5, c5, dict(mul_args=TFComplexMul()), "tree_cplx_i32", "packed 16-bit"),
             (Qcomplex(P(6, 3), P(6, 3)), c5, dict(mul_args=BasicComplexMul()), "tree_cplx_i32", "packed 16-bit"),
             (Qcomplex(P(8, 4), P(8, 4)), c5, dict(mul_args=TFComplexMul()), "tree_cplx_i32", "left-justified")]
    for e, ec, kw, kernel, form in cases:
        for M, N in ((1, 1), (1, 17), (33, 1), (65, 17), (4, 48)):
            if N == 1 and kernel == "tree_i32":
                continue                                   # (one column: the gemv kernel's shapes, tests/test_gpu_gemv.py)
            d = lower(e, e, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == kernel and info.reason.decode().endswith(form), (M, N, K, info.reason)
            for dist in (0, 1):
                _vs_oracle(oracle, e, e, ec, M, N, K, dist=dist, **kw)


def test_complex_fixed_mode_step_forms(oracle):
    """RND::POS_INF + SAT::TCPL everywhere (BASELINE configuration 5's modes): the complex kernel runs its steps in the
    compact branch-free form (one scalar load per step, alignment and exact left shifts folded into 24-bit multiply-adds,
    one v_med3_i32 per value) when every step fits it, else in the table-driven fixed form; both must equal the oracle and
    the run-time-mode kernel.  The planner reports the form in info.reason."""
    P = lambda i, f: Qu(i, f, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(P(6, 3), P(6, -3))
    cases = [
        # (elements, C, lowering keywords, expected form)
        (c5, c5, dict(mul_args=TFComplexMul()), "fixed modes, one clamp, packed 16-bit"),    # configuration 5 itself: every in-loop value is int<6,3>
        (c5, Qcomplex(P(12, 4), P(10, 2)), dict(mul_args=TFComplexMul()), "fixed modes, one clamp, packed 16-bit"),   # ... whatever C is
        (Qcomplex(P(5, 4), P(7, 2)), c5, dict(mul_args=TFComplexMul(abT=Tags(7, 2), cdT=Tags(5, 4), abcT=Tags(6, 3), cdbT=Tags(6, 3), badT=Tags(6, 3), ABT=Tags(6, 3), BCT=Tags(6, 3)),
                                               add_args=[Qcomplex(P(6, 3), P(6, 3))]),
         "fixed modes, one clamp, packed 16-bit"),                                            # products of different alignments into one format
        (c5, c5, dict(mul_args=TFComplexMul(ABT=Tags(7, 3))), "fixed modes, compact"),           # one wider difference: not uniform
        (Qcomplex(P(8, 4), P(8, 4)), c5, dict(mul_args=TFComplexMul()), "fixed modes, one clamp, left-justified"),   # 13 bits and a shift by 4: no room in 16-bit halves
        # SAT::SMGN everywhere: one range [-hi, hi], but not the range of a left-justified int32: v_med3 with the bounds in registers
        (Qcomplex(Qu(6, 3, True, RND.NEG_INF, SAT.SMGN), Qu(6, 3, True, RND.NEG_INF, SAT.SMGN)), c5, dict(mul_args=TFComplexMul()), "fixed modes, one clamp for the whole loop"),
        (Qcomplex(Qu(6, 3, True, RND.NEG_INF, SAT.SMGN), Qu(6, 3, True, RND.NEG_INF, SAT.SMGN)), c5, dict(mul_args=BasicComplexMul()), "fixed modes, one clamp for the whole loop"),
        (Qcomplex(P(6, 3), P(6, 3)), c5, dict(mul_args=BasicComplexMul()), "fixed modes, one clamp, packed 16-bit"),   # Basic with equal part formats
        (Qcomplex(P(5, 4), P(7, 2)), c5, dict(mul_args=BasicComplexMul(acT=Tags(6, 9), bdT=Tags(6, 9), adT=Tags(6, 9), bcT=Tags(6, 9), acbdT=Tags(6, 9), adbcT=Tags(6, 9)),
                                               add_args=[Qcomplex(P(6, 9), P(6, 9))]),
         "fixed modes, one clamp, left-justified"),                                            # Basic: products shift left by 1, 5, 3, 3 bits: plane factors
        (c5, c5, dict(mul_args=BasicComplexMul()), "fixed modes, one clamp, packed 16-bit"),      # b d lives in int<6,-3>: same integer bits, its own mask
        (Qcomplex(P(9, 4), P(9, -2)), c5, dict(mul_args=BasicComplexMul()), "fixed modes, one clamp, left-justified"),   # ... the same in 32-bit words
        (Qcomplex(P(6, 3), P(5, 1)), c5, dict(mul_args=BasicComplexMul()), "fixed modes, compact"),  # different integer bits: per-step records
        (c5, Qcomplex(P(12, 4), P(10, 2)), dict(mul_args=TFComplexMul(abcT=Tags(9, 5), ABT=Tags(11, 2)), add_args=[Qcomplex(P(14, 0), P(12, -3))]),
         "fixed modes, compact"),                                                                 # tags, one level type (fewer fraction bits: rounding nodes)
        (c5, Qcomplex(P(12, 4), P(10, 2)), dict(mul_args=BasicComplexMul(), add_args=[Qcomplex(P(10, 2), P(10, 0)), Qcomplex(P(14, 5), P(12, 3))]),
         "fixed modes, compact"),                                                                 # level 1 has MORE fraction bits than level 0: a node shifts left
        (c5, Qcomplex(P(12, 4), P(10, 2)), dict(mul_args=BasicComplexMul(acT=Tags(20, 8))), "fixed modes, table"),   # a 29-bit product format: beyond v_mad_i32_i24's operands
        # the reference's DEFAULT modes (TRN::TCPL / SAT::TCPL), RND::NEG_INF and SAT::SMGN are compact too
        (Qcomplex(Qu(6, 3), Qu(6, -3)), Qcomplex(Qu(9, 3), Qu(9, 1)), dict(mul_args=TFComplexMul()), "fixed modes, one clamp, packed 16-bit"),
        (Qcomplex(Qu(5, 4, True, RND.NEG_INF, SAT.SMGN), Qu(6, 2, True, RND.NEG_INF, SAT.SMGN)), Qcomplex(Qu(8, 2, True, RND.NEG_INF, SAT.SMGN), Qu(8, 2, True, TRN.TCPL, SAT.TCPL)),
         dict(mul_args=BasicComplexMul(), add_args=[Qcomplex(Qu(12, 3, True, RND.NEG_INF, SAT.SMGN), Qu(12, 1, True, RND.NEG_INF, SAT.SMGN))]), "fixed modes, compact"),
        # value-dependent roundings (RND::ZERO / INF / CONV, TRN::SMGN) and SAT::ZERO / WRP::TCPL: rounding / overflow kinds of the compact form,
        # branch-free unless an unsigned WRP::TCPL is among them (the last case)
        (Qcomplex(Qu(6, 3, True, RND.CONV), Qu(6, 3, True, RND.CONV)), Qcomplex(Qu(9, 3), Qu(9, 1)), dict(mul_args=TFComplexMul()), "compact, branch-free rounding / overflow kinds"),
        (Qcomplex(Qu(8, 3, True, TRN.TCPL, SAT.ZERO), Qu(4, 5, False, RND.POS_INF, SAT.SMGN)), Qcomplex(Qu(9, 3, True, RND.ZERO, SAT.ZERO), Qu(7, 1, True, TRN.SMGN, WRP.TCPL)),
         dict(mul_args=BasicComplexMul()), "compact, branch-free rounding / overflow kinds"),     # RND::ZERO, TRN::SMGN, SAT::ZERO, signed WRP::TCPL: all three feature bits
        (Qcomplex(Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3, True, TRN.TCPL, SAT.ZERO)), Qcomplex(Qu(9, 3, True, TRN.TCPL, SAT.ZERO), Qu(9, 1, True, TRN.TCPL, SAT.ZERO)),
         dict(mul_args=TFComplexMul()), "compact, branch-free rounding / overflow kinds"),        # SAT::ZERO everywhere
        (Qcomplex(Qu(6, 3, True, TRN.TCPL, WRP.TCPL), Qu(6, 1, True, TRN.TCPL, WRP.TCPL)), Qcomplex(Qu(7, 3, True, TRN.TCPL, WRP.TCPL), Qu(5, 1, True, TRN.TCPL, WRP.TCPL)),
         dict(mul_args=TFComplexMul()), "compact, branch-free rounding / overflow kinds"),        # signed WRP::TCPL everywhere
        (Qcomplex(Qu(6, 3, True, RND.INF), Qu(6, 1, True, RND.INF, SAT.ZERO)), Qcomplex(Qu(7, 1, True, RND.INF), Qu(5, 0, True, RND.INF, WRP.TCPL)),
         dict(mul_args=BasicComplexMul(acT=Tags(8, 4)), add_args=[Qcomplex(Qu(12, 2, True, RND.INF), Qu(12, 1, True, RND.INF, SAT.ZERO))]),
         "compact, branch-free rounding / overflow kinds"),                                       # RND::INF: the inverted sign bit through a 2^31 bias
        (Qcomplex(Qu(5, 4, True, RND.INF, WRP.TCPL), Qu(6, 2, False, RND.ZERO, WRP.TCPL)), Qcomplex(Qu(6, 2, True, RND.INF, SAT.ZERO), Qu(5, 1, False, RND.CONV, WRP.TCPL)),
         dict(mul_args=TFComplexMul(), add_args=[Qcomplex(Qu(9, 3, True, RND.INF, WRP.TCPL), Qu(8, 1, False, TRN.SMGN, WRP.TCPL)),
                                                 Qcomplex(Qu(7, 2, True, RND.ZERO, SAT.ZERO), Qu(9, 3, True, RND.CONV, SAT.SMGN))]), "compact, rounding / overflow kinds"),
    ]
    for K in (64, 250):
        for e, ec, kw, form in cases:
            M, N = 45, 38
            d = lower(e, e, ec, M, N, K, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_cplx_i32" and info.reason.decode().endswith(form), (info.reason, form)
            assert capi.classify(d, capi.OPT_RUNTIME_MODES).reason.decode().endswith("run-time modes")
            a = _vs_oracle(oracle, e, e, ec, M, N, K, expect_kernel="tree_cplx_i32", **kw)
            b = _vs_oracle(oracle, e, e, ec, M, N, K, flags=capi.OPT_RUNTIME_MODES, expect_kernel="tree_cplx_i32", **kw)
            assert fields_equal(a, b)


def test_leading_dimensions(oracle):
    M, N, K = 70, 50, 96
    ea = eb = E43
    d = lower(ea, eb, W16, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(19, 6)])
    lda, ldb, ldc = M + 5, K + 3, M + 7
    A = oracle.fill(ea, lda * K, 5)
    B = oracle.fill(eb, ldb * N, 6)
    out = np.full(ldc * N, -12345, dtype=np.int32)
    capi.run(d, out, A, B, lda=lda, ldb=ldb, ldc=ldc)
    exp = np.full(ldc * N, -12345, dtype=np.int32)
    oracle.gemm(d, A, B, W16, lda=lda, ldb=ldb, ldc=ldc, out=exp)
    assert np.array_equal(out, exp)  # also: the padding rows between columns of C are untouched


def test_range_check(oracle):
    d = lower(E43, E43, W16, 16, 16, 16)
    A = oracle.fill(E43, 256, 1)
    B = oracle.fill(E43, 256, 2)
    bad = A.copy()
    bad[7] = 1000  # outside int<4,3>
    out = np.zeros(256, np.int32)
    capi.run(d, out, A, B, flags=capi.OPT_CHECK_RANGE)
    with pytest.raises(capi.QgemulError) as ei:
        capi.run(d, out, bad, B, flags=capi.OPT_CHECK_RANGE)
    assert ei.value.status == capi.QG_ERANGE


def test_resident_api_and_device_fill(oracle):
    """pack / execute / unpack on resident buffers, and the device-side synthetic generator equals the host one."""
    M, N, K = 256, 256, 512
    d = lower(E43, E43, W16, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(19, 6)])
    A = oracle.fill(E43, M * K, 1)
    B = oracle.fill(E43, K * N, 2)
    exp = oracle.gemm(d, A, B, W16, nthreads=8)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        dC = ctx.alloc(M * N * 4)
        plan.fill(capi.OPERAND_A, 1, 0, pA)
        plan.fill(capi.OPERAND_B, 2, 0, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, dC)
        got = np.zeros(M * N, np.int32)
        ctx.d2h(got, dC)
        assert np.array_equal(got, exp)
        ms = plan.time_execute(pC, pA, pB, 2, 5)
        assert ms > 0
        for p in (pA, pB, pC, dC):
            ctx.free(p)
        plan.close()


@pytest.mark.parametrize("elem,K,dist", [
    (Qu(8, 8, True, TRN.TCPL, SAT.ZERO), 4096, 1),    # configuration 3 as configured: split product, SAT::ZERO, biased nodes
    (Qu(8, 8, True, TRN.TCPL, SAT.ZERO), 64, 0),
    (Qu(8, 8, True, TRN.TCPL, SAT.TCPL), 256, 1),     # split product, clamp nodes
    (Qu(4, 3), 1024, 1),                               # direct product with a TCPL shift, clamp nodes
    (Qu(4, 3, True, TRN.TCPL, SAT.ZERO), 128, 1),
    (Qu(4, 4, False), 64, 1),                          # unsigned: lower bound 0
    (Qu(4, 4, False, TRN.TCPL, SAT.ZERO), 64, 1),
    (Qu(6, 0), 32, 1),                                 # integer format: no product shift
])
def test_fast_tree_fixed_mode_variants(oracle, elem, K, dist):
    """Default-tag shapes take the fixed-mode variants of the 32-bit tree kernel (one format everywhere);
    they must equal the oracle and the runtime-mode variant of the same kernel."""
    wide = Qu(elem.intBits + 6, elem.fracBits, elem.isSigned)
    for ec in (elem, wide):
        a = _vs_oracle(oracle, elem, elem, ec, 70, 40, K, dist=dist, expect_kernel="tree_i32")
        b = _vs_oracle(oracle, elem, elem, ec, 70, 40, K, dist=dist, flags=capi.OPT_RUNTIME_MODES, expect_kernel="tree_i32")
        assert np.array_equal(a, b)
    _vs_oracle(oracle, elem, elem, elem, 33, 65, K, dist=0, mul_args=elem, add_args=[elem], transposed_a=True, expect_kernel="tree_i32")


def test_complex_linear_class_on_mfma(oracle):
    """Complex operands whose BasicComplexMul sub-ops and tree levels are all exact: four real int8-limb dot
    products on MFMA + one combine pass, against the oracle and against the exact tree kernel."""
    r = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
    i = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r, i)
    BL = BasicComplexMul(acT=Qu(14, 6), bdT=Qu(14, -6), adT=Qu(14, 0), bcT=Qu(14, 0), acbdT=Qu(15, 6), adbcT=Qu(15, 0))
    lw = Qcomplex(Qu(30, 6), Qu(30, 0))
    wide = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    narrow = Qcomplex(Qu(9, 2, True, RND.CONV, SAT.SMGN), Qu(7, -1, True, RND.ZERO, WRP.TCPL))
    for ec, shape, ta in ((wide, (200, 130, 512), False), (narrow, (130, 257, 256), True), (c5, (64, 64, 2048), False)):
        M, N, K = shape
        a = _vs_oracle(oracle, c5, c5, ec, M, N, K, mul_args=BL, add_args=[lw], transposed_a=ta, expect_kernel="mfma_cplx")
        b = _vs_oracle(oracle, c5, c5, ec, M, N, K, mul_args=BL, add_args=[lw], transposed_a=ta, flags=capi.OPT_FORCE_TREE)
        assert fields_equal(a, b)
    # symmetric 8-bit parts: single limb per part
    r8 = Qu(4, 3)
    c8 = Qcomplex(r8, r8)
    # adbcT needs 11 int bits: ad + bc reaches exactly 2^15 at a=b=c=d=-128, and two of those overflow Qu<10,6> by one LSB
    B8 = BasicComplexMul(acT=Qu(9, 6), bdT=Qu(9, 6), adT=Qu(9, 6), bcT=Qu(9, 6), acbdT=Qu(11, 6), adbcT=Qu(11, 6))
    l8 = Qcomplex(Qu(22, 6), Qu(22, 6))
    _vs_oracle(oracle, c8, c8, Qcomplex(Qu(16, 3), Qu(16, 3)), 300, 200, 1024, mul_args=B8, add_args=[l8], expect_kernel="mfma_cplx")


@pytest.mark.parametrize("amax,bmax", [(32639, 32639), (32640, 32639), (32639, 32640), (65535, 127), (127, 127), (0, 65535)])
def test_limb_plane_mask_dispatch(oracle, amax, bmax):
    """3x3-limb operands carry a plane mask; when the third int8 limb plane of BOTH operands is empty (all values inside
    [-32640, 32639]) a 2x2-limb kernel does the work, otherwise the 3x3 one — chosen on the device.  Values sit exactly on
    the boundary: 32639 is the largest two-limb value, 32640 / -32641 need the third limb."""
    M, N, K = 192, 130, 256
    wide = Qu(26, 8)
    d = lower(E88Z, E88Z, wide, M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "mfma_i8_limb"
    rng = np.random.default_rng(amax * 7 + bmax)

    def operand(n, vmax):
        x = rng.integers(-vmax - 1, vmax + 1, n, dtype=np.int64) if vmax else np.zeros(n, dtype=np.int64)
        if vmax:
            x[:4] = [vmax, -vmax - 1, vmax, 0]      # the extremes are present
        return np.clip(x, E88Z.raw_min, E88Z.raw_max).astype(np.int32)
    A, B = operand(M * K, amax), operand(K * N, bmax)
    got = run_gpu(d, A, B, wide, oracle)
    assert np.array_equal(got, oracle.gemm(d, A, B, wide, nthreads=8))
    # re-packing other data into the SAME cached device buffers must not leave a stale mask behind
    A2 = operand(M * K, 65535)
    got2 = run_gpu(d, A2, B, wide, oracle)
    assert np.array_equal(got2, oracle.gemm(d, A2, B, wide, nthreads=8))


@pytest.mark.parametrize("K", [17, 33, 100, 255, 257, 1000, 1023, 1025, 3000])
def test_any_k_on_the_32bit_tree_kernels(oracle, K):
    """K need not be a power of two: the 32-bit tree kernels run on operands zero-padded to 2^ceil(log2 K) leaves, where a
    node whose right child is zero IS the reference's converting copy of an odd leftover (QuBLAS.h:4977-4980).  Against the
    oracle (which implements the leftover rule literally) and against the general 64-bit kernel, with level types that
    round and saturate for real, real and complex."""
    lv = [Qu(9, 6, True, RND.CONV, SAT.SMGN), Qu(11, 4, True, RND.ZERO, SAT.TCPL), Qu(12, 2, True, TRN.SMGN, WRP.TCPL)]
    for kw in (dict(), dict(mul_args=Qu(5, 4, True, RND.INF, SAT.TCPL), add_args=lv)):
        a = _vs_oracle(oracle, E43, E43, W16, 37, 21, K, expect_kernel="tree_i32", **kw)
        b = _vs_oracle(oracle, E43, E43, W16, 37, 21, K, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_i64", **kw)
        assert np.array_equal(a, b)
    _vs_oracle(oracle, E88Z, E88Z, E88Z, 20, 33, K, dist=1, expect_kernel="tree_i32")      # biased SAT::ZERO fixed-mode variant
    r55 = Qu(5, 5)
    c55 = Qcomplex(r55, r55)
    wide = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    l1 = Qcomplex(Qu(12, 4, True, RND.CONV, SAT.SMGN), Qu(11, 6, True, TRN.SMGN, SAT.ZERO))
    if K <= 1025:
        a = _vs_oracle(oracle, c55, c55, wide, 9, 7, K, mul_args=TFComplexMul(), add_args=[l1], expect_kernel="tree_cplx_i32")
        b = _vs_oracle(oracle, c55, c55, wide, 9, 7, K, mul_args=TFComplexMul(), add_args=[l1], flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_cplx")
        assert fields_equal(a, b)


def test_one_shot_call_from_several_threads(oracle):
    """qgemul_run keeps a cache per calling thread (context, plan, device buffers): concurrent callers with different
    descriptors must neither crash nor see each other's buffers; alternating descriptors in one thread re-plans."""
    import threading
    specs = [(E43, W16, 96, 80, 128, dict(mul_args=Tags(9, 6), add_args=[Qu(19, 6)])),
             (E88Z, Qu(24, 8), 64, 72, 192, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)])),
             (E88Z, E88Z, 40, 56, 100, dict()),
             (E43, E43, 33, 17, 77, dict())]
    jobs = []
    for e, ec, M, N, K, kw in specs:
        d = lower(e, e, ec, M, N, K, **kw)
        A, B = oracle.fill(e, M * K, 11, 0), oracle.fill(e, K * N, 12, 0)
        jobs.append((d, A, B, ec, oracle.gemm(d, A, B, ec, nthreads=4)))
    errors = []

    def worker(order):
        try:
            for _ in range(6):
                for i in order:
                    d, A, B, ec, exp = jobs[i]
                    got = run_gpu(d, A, B, ec, oracle)
                    if not np.array_equal(got, exp):
                        errors.append(("mismatch", i))
            capi.run_release()
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
    threads = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3], [3, 2, 1, 0], [1, 1, 3, 0], [2, 0, 2, 1])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("K", [32, 100, 1000, 4096])
def test_wide_formats_tree_kernel(oracle, K):
    """tree class with formats beyond 31 bits (a rounding product behind wide level types, wide operands): the 64-bit
    2x2-per-lane kernel (default) against the oracle and against the general one-output-per-lane kernel"""
    wide_lv = [Qu(29, 16, True, RND.CONV, SAT.SMGN), Qu(33, 12, True, RND.ZERO, SAT.TCPL)]
    for ea, ec, kw in ((E88Z, Qu(30, 8, True, RND.INF, SAT.TCPL), dict(mul_args=Tags(12, 10), add_args=wide_lv)),
                       (Qu(14, 10), Qu(30, 10), dict(mul_args=Tags(20, 14), add_args=[Qu(34, 14)])),
                       (E43, W16, dict(add_args=[Qu(40, 3)]))):
        a = _vs_oracle(oracle, ea, ea, ec, 70, 45, K, expect_kernel="tree_i64", **kw)
        b = _vs_oracle(oracle, ea, ea, ec, 70, 45, K, flags=capi.OPT_GENERIC_TREE, expect_kernel="tree_i64", **kw)
        assert np.array_equal(a, b)


def test_single_limb_left_shifting_epilogue(oracle):
    """found by tests/extended_fuzz.py: int<10,-3> operands (one int8 limb), exact product and sums at fracBits -6, C at
    fracBits 7 — the epilogue shifts the int32 dot product LEFT by 13 bits, past 32 bits, before C's SAT::SMGN sees it.  Such
    descriptors convert the raw dot products in the 64-bit pass; every overflow mode, and a C wider than 32 bits."""
    e = Qu(10, -3)
    for om in (SAT.TCPL, SAT.ZERO, SAT.SMGN, WRP.TCPL):
        for c in (Qu(5, 7, True, TRN.SMGN, om), Qu(30, 10, True, TRN.TCPL, om)):
            for dist in (0, 1):
                _vs_oracle(oracle, e, e, c, 36, 50, 494, dist=dist, mul_args=Qu(21, -6), add_args=[Qu(33, -6)], expect_kernel="mfma_i8")
    _vs_oracle(oracle, e, e, Qu(5, 7, True, TRN.SMGN, SAT.SMGN), 300, 260, 4096, mul_args=Qu(21, -6), add_args=[Qu(33, -6)], expect_kernel="mfma_i8")


def test_single_limb_into_a_c_wider_than_32_bits(oracle):
    """second find of tests/extended_fuzz_shapes.py: one-limb operands into Qu<24,9> (33 value bits) — the 32-bit epilogue
    cannot even hold C's clamp bounds; every overflow mode, shifts in both directions"""
    a, b = Qu(7, -2, False), Qu(7, -1)
    for om in (SAT.TCPL, SAT.ZERO, SAT.SMGN, WRP.TCPL):
        for c in (Qu(24, 9, True, TRN.TCPL, om), Qu(34, 0, True, RND.CONV, om), Qu(28, 3, False, RND.ZERO, om)):
            _vs_oracle(oracle, a, b, c, 129, 1, 100, mul_args=Qu(15, -3), add_args=[Qu(33, -3)], transposed_a=True, expect_kernel="mfma_i8")
            _vs_oracle(oracle, E43, E43, c, 70, 90, 256, mul_args=Tags(9, 6), add_args=[Qu(21, 6)], expect_kernel="mfma_i8")


@pytest.mark.parametrize("ea,eb", [(Qu(6, 5), Qu(6, 5)), (Qu(7, 5, False), Qu(3, 8)), (Qu(9, 2), Qu(5, 4, False)), (Qu(11, 0), Qu(4, 5))])
@pytest.mark.parametrize("shape", [(1536, 1408, 128), (2048, 1200, 100), (1290, 1700, 64), (256, 384, 512), (1, 1, 1)])
def test_karatsuba_two_digit_kernel(oracle, ea, eb, shape):
    """operands of 9..12 value+sign bits: two unsigned base-64 digits of the biased value, THREE MFMA products per k-step
    (Karatsuba), bias removed with the operands' row sums — against the oracle, both A orientations, full and small range,
    and against the exact tree kernel"""
    M, N, K = shape
    pf = Qu(ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits, ea.isSigned or eb.isSigned)
    kw = dict(mul_args=pf, add_args=[Qu(pf.intBits + 14, pf.fracBits, pf.isSigned)])
    c = Qu(14, 3, True, RND.CONV, SAT.SMGN)
    # the Karatsuba layout carries int64 row sums behind the planes: visible in the packed size (problems with more than 128
    # tiles of 128x128 only; smaller ones are latency-bound and keep the four-product kernel)
    info = capi.classify(lower(ea, eb, c, M, N, K, **kw))
    rows_p, k_p = -(-M // 128) * 128, -(-K // 64) * 64
    kara = info.packed_bytes[0] == 2 * rows_p * k_p + 256 + 8 * rows_p
    mid, small = -(-M // 128) * -(-N // 128), -(-M // 64) * -(-N // 64)
    assert kara == (not (mid <= 128 and small > mid))      # i.e. whenever the planner chose 128x128 tiles
    for ta in (False, True):
        for dist in (0, 1):
            a = _vs_oracle(oracle, ea, eb, c, M, N, K, dist=dist, transposed_a=ta, expect_kernel="mfma_i8_limb", **kw)
    b = _vs_oracle(oracle, ea, eb, c, M, N, K, dist=1, transposed_a=True, flags=capi.OPT_FORCE_TREE, **kw)
    assert np.array_equal(a, b)
