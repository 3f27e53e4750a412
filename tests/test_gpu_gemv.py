"""GPU parity of the one-column tree kernel (qg_gemv.hip; SURVEY.md 8-f #1: batched Qreduce / fixed-point GEMV) against
the CPU restatement, through the C-ABI.  N = 1, K = 2^p >= 16; every product and every tree node quantised in the
reference's order.  Bit-exact."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, WRP, Tags, lower, lower_reduce

pytestmark = pytest.mark.gpu

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E43 = Qu(4, 3)


def _check(oracle, d, ea, eb, ec, *, dist=0, expect="gemv_i32", flags=0, ones=False):
    info = capi.classify(d, flags)
    assert capi.KERNEL_NAMES[info.kernel] == expect, (capi.KERNEL_NAMES[info.kernel], info.reason)
    M, K = d.M, d.K
    A = oracle.fill(ea, M * K, 5, dist)
    B = np.ones(K, dtype=np.int32) if ones else oracle.fill(eb, K, 6, dist)
    got = capi.run(d, np.zeros(M, dtype=oracle.host_dtype(ec)), A, B, flags=flags)
    exp = oracle.gemm(d, A, B, ec, nthreads=8)
    assert np.array_equal(got, exp)
    return got


@pytest.mark.parametrize("K", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 65536])
def test_batched_qreduce_default_levels(oracle, K):
    """Qreduce<>(v) of `rows` vectors: every pair add quantised into the element type (TRN::TCPL / SAT::ZERO)."""
    rows = 777 if K <= 8192 else 40
    got = _check(oracle, lower_reduce(E88, rows, K), E88, None, E88, dist=1, ones=True)
    assert len(np.unique(got)) > 8


@pytest.mark.parametrize("K", [16, 64, 128, 256, 4096, 16384])
@pytest.mark.parametrize("levels", [[Qu(12, 6, True, RND.CONV, SAT.SMGN)],
                                    [Qu(9, 8, True, RND.ZERO, SAT.TCPL), Qu(11, 5, True, RND.INF, WRP.TCPL), Qu(14, 3, True, TRN.SMGN, SAT.ZERO)]],
                         ids=["one_level_type", "three_level_types"])
def test_batched_qreduce_level_lists(oracle, K, levels):
    _check(oracle, lower_reduce(E88, 300, K, levels), E88, None, levels[-1], ones=True)


def test_qreduce_rows_are_masked_by_a_zero_one_vector(oracle):
    """the Qreduce lowering's B is a 0/1 vector: zeros drop leaves exactly (product a*0 = 0 in a's format)"""
    d = lower_reduce(E88, 500, 1024, [Qu(14, 8)])
    A = oracle.fill(E88, 500 * 1024, 5, 0)
    B = (np.arange(1024) % 3 != 0).astype(np.int32)
    got = capi.run(d, np.zeros(500, dtype=np.int32), A, B)
    assert np.array_equal(got, oracle.gemm(d, A, B, Qu(14, 8), nthreads=8))


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("K", [16, 32, 128, 256, 2048, 32768])
def test_gemv_general_vector(oracle, K, ta):
    """C[M x 1] = A * b with a real vector b: products rounded into the default product format, default tree levels"""
    M = 513
    _check(oracle, lower(E43, E43, Qu(12, 3), M, 1, K, transposed_a=ta), E43, E43, Qu(12, 3))
    pm = Qu(5, 4, True, RND.INF, SAT.TCPL)
    _check(oracle, lower(E43, E43, Qu(12, 3), M, 1, K, mul_args=pm, add_args=[Qu(8, 4, True, RND.CONV, SAT.SMGN)], transposed_a=ta), E43, E43, Qu(12, 3))


def test_gemv_wide_products(oracle):
    """int<8,8> x int<8,8>: 34-bit unrounded products, formed in 64 bits, rounded into 17-bit values"""
    got = _check(oracle, lower(E88, E88, Qu(15, 8), 300, 1, 4096), E88, E88, Qu(15, 8), dist=1)
    assert len(np.unique(got)) > 8


def test_kernel_choice_is_invariant(oracle):
    """the one-column kernel, the 32-column tree kernel's path (GENERIC_TREE) and (for a linear-class descriptor) the MFMA
    kernel agree"""
    d = lower_reduce(E43, 400, 1024, [Qu(16, 3)])           # exact levels: linear class -> MFMA by default
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "mfma_i8"
    A = oracle.fill(E43, 400 * 1024, 5, 0)
    B = np.ones(1024, dtype=np.int32)
    outs = [capi.run(d, np.zeros(400, dtype=np.int32), A, B, flags=f) for f in (0, capi.OPT_FORCE_TREE, capi.OPT_FORCE_TREE | capi.OPT_GENERIC_TREE)]
    assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_FORCE_TREE).kernel] == "gemv_i32"
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    assert np.array_equal(outs[0], oracle.gemm(d, A, B, Qu(16, 3), nthreads=8))


@pytest.mark.parametrize("K", [9, 17, 100, 255, 1000, 5000])
def test_any_length(oracle, K):
    """lengths that are not powers of two: operands zero-padded to 2^ceil(log2 K) leaves (a node with a zero right child is
    the reference's converting copy of an odd leftover); default levels and a level list that rounds for real"""
    _check(oracle, lower_reduce(E88, 333, K), E88, None, E88, dist=1, ones=True)
    lv = [Qu(9, 8, True, RND.ZERO, SAT.TCPL), Qu(11, 5, True, RND.INF, WRP.TCPL), Qu(14, 3, True, TRN.SMGN, SAT.ZERO)]
    d = lower_reduce(E88, 333, K, lv)
    _check(oracle, d, E88, None, Qu.from_tuple([d.c[0].I, d.c[0].F, d.c[0].S, d.c[0].Q, d.c[0].O]), ones=True)
    _check(oracle, lower(E43, E43, Qu(12, 3), 100, 1, K), E43, E43, Qu(12, 3))


def test_very_short_vectors(oracle):
    """Fewer than 5 tree levels: the tree is continued with identity levels (x + 0 in x's own format) and the row zero-padded to
    32 leaves, so that the one-column kernel takes these too (round 1 sent them to the general kernel)."""
    for K in (1, 2, 5, 8, 13, 16):
        for levels in (None, [Qu(12, 8)], [Qu(10, 8, True, TRN.TCPL, SAT.ZERO), Qu(14, 6)]):
            d = lower_reduce(E88, 64, K, levels)
            assert capi.KERNEL_NAMES[capi.classify(d).kernel] in ("gemv_i32", "mfma_i8_limb")   # (a level type that holds every sum: linear class)
            assert levels or K == 1 or capi.KERNEL_NAMES[capi.classify(d).kernel] == "gemv_i32"   # (K = 1: no add at all, linear too)
            from qublas_amd.desc import reduce_result_type
            ec = reduce_result_type(E88, levels or [], K)
            A = oracle.fill(E88, 64 * K, 3, 1)
            exp = oracle.gemm(d, A, np.ones(K, dtype=np.int32), ec)
            for flags in (0, capi.OPT_RUNTIME_MODES, capi.OPT_GENERIC_TREE):
                got = capi.run(d, np.zeros(64, dtype=oracle.host_dtype(ec)), A, np.ones(K, dtype=np.int32), flags=flags)
                assert np.array_equal(got, exp), (K, levels, flags)


@pytest.mark.parametrize("K", [64, 256, 1000, 8192])
def test_fixed_mode_nodes_equal_runtime_mode_nodes(oracle, K):
    """default tags give one level format with SAT::ZERO (int<8,8> TCPL/ZERO) or SAT::TCPL (int<4,3>): the fixed-mode kernel
    variants (add + range test / clamp) against the run-time-mode variant (QG_OPT_RUNTIME_MODES) and the oracle"""
    for e, ones in ((E88, True), (E43, True), (E43, False)):
        d = lower_reduce(e, 257, K) if ones else lower(e, e, e, 257, 1, K)
        a = _check(oracle, d, e, e, e, ones=ones)
        b = _check(oracle, d, e, e, e, ones=ones, flags=capi.OPT_RUNTIME_MODES)
        assert np.array_equal(a, b)


def test_one_column_step_forms_and_the_smgn_minimum(oracle):
    """Per-level formats on the one-column kernels (the README's Qreduce<list>): compact records, against the oracle and the
    run-time-mode kernel, long and short rows.  And the one raw value for which `a * 1 into a's own format` is NOT the
    identity: -2^W of a signed SAT::SMGN element type (that conversion would clamp it, the reference's Qreduce adds it as it
    is: tests/golden/ref_scalar_7, tests/test_reduce.py).  Found by tests/extended_fuzz_tree_forms.py: the 0/1-vector
    shortcut passed it through while the other kernels clamped it.  lower_reduce now names a's format with SAT::TCPL as the
    leaf format of such element types, which every kernel treats alike."""
    from qublas_amd.desc import SAT, TRN, WRP, RND, lower_reduce, reduce_result_type
    t1 = Qu(6, 3, True, TRN.TCPL, SAT.ZERO)
    cases = [
        (t1, [t1, Qu(6, -3)], "per-level formats, compact"),                                   # the README's level list
        (Qu(8, 8), [Qu(10, 8), Qu(12, 8)], "per-level formats, compact (clamps)"),
        (Qu(4, 3), [Qu(6, 5, True, RND.POS_INF, SAT.SMGN), Qu(9, 2, True, RND.NEG_INF, WRP.TCPL)], "per-level formats, compact"),
        (Qu(3, 4, True, TRN.TCPL, SAT.SMGN), [Qu(4, 6, True, TRN.TCPL, SAT.ZERO)], "per-level formats, compact"),   # SMGN elements, a level with more fraction bits
    ]
    for e, levels, form in cases:
        for rows, K in ((700, 64), (37, 1024), (5, 4096)):
            d = lower_reduce(e, rows, K, levels)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "gemv_i32" and info.reason.decode().endswith(form), (info.reason, form)
            ec = reduce_result_type(e, levels, K)
            A = oracle.fill(e, rows * K, 11, 0)
            A[:3] = e.raw_min                        # the format's raw minimum is present
            B = np.ones(K, dtype=np.int32)
            exp = oracle.gemm(d, A, B, ec)
            for flags in (0, capi.OPT_RUNTIME_MODES):
                out = np.zeros(rows, dtype=oracle.host_dtype(ec))
                capi.run(d, out, A, B, flags=flags)
                assert np.array_equal(out, exp), (str(e), K, flags)


def test_one_column_64_bit_values(oracle):
    """Elements of at most 32 storage bits whose sums or level types need more than 31 bits (Q15.16, 24-bit elements under
    40-bit levels ...): the one-column kernels with 64-bit tree values (gemv_i64) instead of the general 64-bit tree kernel,
    which is built for square tiles.  Qreduce lowerings and GEMVs, long and short rows, several modes, against the oracle."""
    from qublas_amd.desc import SAT, TRN, WRP, RND, lower_reduce, reduce_result_type
    q = Qu(15, 16)
    cases = [
        (q, [Qu(24, 16)]),
        (q, None),                                                       # every node saturates at 32 bits
        (q, [Qu(20, 12, True, RND.CONV, SAT.SMGN), Qu(30, 8, True, RND.ZERO, SAT.ZERO)]),
        (Qu(12, 11), [Qu(28, 11), Qu(34, 6, True, TRN.SMGN, WRP.TCPL)]),
        (Qu(15, 16, False), [Qu(30, 16, False)]),                        # unsigned elements of 32 storage bits
    ]
    for e, levels in cases:
        for rows, K in ((300, 16), (37, 64), (700, 128), (9, 256), (41, 1024), (5, 8192)):
            d = lower_reduce(e, rows, K, levels)
            info = capi.classify(d)
            w32 = e is q and levels is None and K >= 256      # (every node in the 32-bit word itself: test_one_column_32_bit_words)
            assert capi.KERNEL_NAMES[info.kernel] == ("gemv_i32" if w32 else "gemv_i64"), (str(e), K, info.reason)
            ec = reduce_result_type(e, levels or [], K)
            A = oracle.fill(e, rows * K, 21, 0)
            A[:2] = e.raw_min
            A[2:4] = e.raw_max
            got = capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, np.ones(K, dtype=np.int32))
            assert np.array_equal(got, oracle.gemm(d, A, np.ones(K, dtype=np.int32), ec)), (str(e), rows, K)
    # a GEMV (B is a vector of the element type, products rounded into a 32-bit format)
    for K in (64, 2048):
        d = lower(q, q, Qu(28, 12), 130, 1, K, mul_args=Qu(15, 16, True, RND.POS_INF, SAT.TCPL), add_args=[Qu(28, 16), Qu(28, 12)])
        assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "gemv_i64"
        A, B = oracle.fill(q, 130 * K, 5, 0), oracle.fill(q, K, 6, 0)
        got = capi.run(d, np.zeros(130, dtype=oracle.host_dtype(Qu(28, 12))), A, B)
        assert np.array_equal(got, oracle.gemm(d, A, B, Qu(28, 12)))



@pytest.mark.parametrize("K", [256, 300, 1024, 4096, 5000, 65536])
def test_one_column_32_bit_words(oracle, K):
    """Qreduce / GEMV on 32-bit words — Q15.16, Q31, int32 with default levels: product and every node in ONE signed SAT::TCPL
    format of exactly 32 bits.  The values stay 32-bit words and a node is one saturating add (`k_gemv<., 6>`, rows of at least
    256 leaves) where the 64-bit-value form spent a 64-bit add and a run-time step per node.  Against the oracle with full-range
    elements (the sums saturate, in the tree's own order) and small ones (they do not), and against the 64-bit-value form the
    same descriptor takes under QG_OPT_RUNTIME_MODES."""
    from qublas_amd.desc import reduce_result_type
    rows = 333 if K <= 5000 else 21
    form = "one 32-bit format, saturating adds"
    for e in (Qu(15, 16), Qu(0, 31), Qu(31, 0)):
        d = lower_reduce(e, rows, K)
        info = capi.classify(d)
        assert capi.KERNEL_NAMES[info.kernel] == "gemv_i32" and info.reason.decode().endswith(form), (str(e), info.reason)
        assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_RUNTIME_MODES).kernel] == "gemv_i64"
        ec = reduce_result_type(e, [], K)
        ones = np.ones(K, dtype=np.int32)
        for dist in (0, 1, 2):
            A = oracle.fill(e, rows * K, 31 + dist, dist % 2)
            if dist == 2:
                A = (A >> 13).astype(A.dtype)
            A[:2] = e.raw_min
            A[K:K + 2] = e.raw_max
            got = capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, ones)
            exp = oracle.gemm(d, A, ones, ec, nthreads=8)
            assert np.array_equal(got, exp), (str(e), K, dist)
            assert np.array_equal(capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, ones, flags=capi.OPT_RUNTIME_MODES), exp)
        # a 0/1 mask instead of all ones
        mask = (np.arange(K) % 3 != 0).astype(np.int32)
        A = oracle.fill(e, rows * K, 40, 0)
        assert np.array_equal(capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, mask), oracle.gemm(d, A, mask, ec, nthreads=8))
    # GEMVs: B a vector of elements, the product rounded into the word (several roundings, shifts 16 / 31 / 4), C of other widths
    q = Qu(15, 16)
    for ea, eb, ec, kw in ((q, q, q, {}), (Qu(0, 31), Qu(0, 31), Qu(0, 31), {}),
                           (q, q, Qu(9, 3, True, RND.CONV, SAT.SMGN), dict(mul_args=Qu(15, 16, True, RND.POS_INF, SAT.TCPL), add_args=[q])),
                           (Qu(20, 11), Qu(20, 11), Qu(20, 11, True, RND.NEG_INF, SAT.TCPL), dict(mul_args=Qu(20, 11, True, RND.NEG_INF, SAT.TCPL), add_args=[Qu(20, 11)])),
                           (Qu(8, 12), Qu(4, 8), Qu(20, 11), dict(mul_args=Qu(15, 16, True, RND.CONV, SAT.ZERO), add_args=[q]))):   # (the last: the general product step)
        d = lower(ea, eb, ec, 130, 1, K, **kw)
        info = capi.classify(d)
        assert capi.KERNEL_NAMES[info.kernel] == "gemv_i32" and info.reason.decode().endswith(form), (str(ea), info.reason)
        for dist in (0, 1):
            A, B = oracle.fill(ea, 130 * K, 5, dist), oracle.fill(eb, K, 6, dist)
            got = capi.run(d, np.zeros(130, dtype=oracle.host_dtype(ec)), A, B)
            exp = oracle.gemm(d, A, B, ec, nthreads=8)
            assert np.array_equal(got, exp), (str(ea), str(ec), K, dist)
            assert np.array_equal(capi.run(d, np.zeros(130, dtype=oracle.host_dtype(ec)), A, B, flags=capi.OPT_RUNTIME_MODES), exp)
