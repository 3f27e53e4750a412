"""Composite linear plans (run with -m gpu): the linear class beyond the single-launch MFMA range — operands of more than
three int8 limbs (24 ... 62 storage bits) as limb GROUPS, and reduction lengths beyond the int32 accumulators' exact range
(K * min(LA, LB) >= 2^17) as k-CHUNKS — against the CPU restatement at full K.  The reference has no such boundary
(Reducer, QuBLAS.h:4960-4990; ArbiInt elements up to 64 bits :347-564), and round 2 sent these descriptors to the 64-bit VALU
tree kernel.  Bit-exact; every case also asserts that the planner reports an MFMA kernel."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, WRP, Tags, lower

pytestmark = pytest.mark.gpu

E43 = Qu(4, 3)
E88Z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E1212 = Qu(12, 12)


def _run(oracle, ea, eb, ec, M, N, K, *, kernel, reason, dist=0, flags=0, lda=0, ldb=0, ldc=0, ta=False, **kw):
    d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
    info = capi.classify(d, flags)
    assert capi.KERNEL_NAMES[info.kernel] == kernel, (capi.KERNEL_NAMES[info.kernel], info.reason)
    assert reason in info.reason.decode(), info.reason
    ra, ca = (K, M) if ta else (M, K)
    A = oracle.fill(ea, (lda or ra) * ca, 1, dist)
    B = oracle.fill(eb, (ldb or K) * N, 2, dist)
    out = np.full((ldc or M) * N, -7, dtype=oracle.host_dtype(ec))
    got = capi.run(d, out.copy(), A, B, lda=lda, ldb=ldb, ldc=ldc, flags=flags)
    exp = oracle.gemm(d, A, B, ec, lda=lda, ldb=ldb, ldc=ldc, nthreads=8, out=out.copy())
    assert np.array_equal(got, exp)
    return got


def test_k_chunks_single_limb(oracle):
    """int<4,3>, K = 262 144: three chunks of at most 130 816 reduction indices on the single-limb kernel"""
    c = _run(oracle, E43, E43, Qu(27, 6), 40, 24, 262144, kernel="mfma_i8", reason="3 k-chunk(s) x 1 x 1", mul_args=Tags(9, 6), add_args=[Qu(27, 6)])
    assert len(np.unique(c)) > 100
    # a narrow C that really rounds and saturates, ragged K (the last chunk is short and not a multiple of the k-tile)
    _run(oracle, E43, E43, Qu(12, 2, True, RND.CONV, SAT.SMGN), 33, 17, 131000 + 77, kernel="mfma_i8", reason="2 k-chunk(s)", dist=1,
         mul_args=Tags(9, 6), add_args=[Qu(27, 6)])


def test_k_chunks_three_limbs(oracle):
    """int<8,8>, K = 65 536: two chunks on the 3 x 3-limb kernel"""
    c = _run(oracle, E88Z, E88Z, Qu(33, 16), 24, 16, 65536, kernel="mfma_i8_limb", reason="2 k-chunk(s) x 1 x 1", mul_args=Tags(17, 16),
             add_args=[Qu(33, 16)])
    assert len(np.unique(c)) > 100
    _run(oracle, E88Z, E88Z, Qu(20, 4, True, RND.ZERO, WRP.TCPL), 24, 16, 50000, kernel="mfma_i8_limb", reason="2 k-chunk(s)", ta=True,
         mul_args=Tags(17, 16), add_args=[Qu(33, 16)])


@pytest.mark.parametrize("shape", [(100, 60, 300), (256, 256, 4096), (130, 257, 1000), (1, 3, 1), (700, 520, 128)])
def test_four_limb_operands(oracle, shape):
    """int<12,12> (25 storage bits): 2 x 2 groups of two limbs, four 2 x 2-limb launches and the exact combine"""
    M, N, K = shape
    c = _run(oracle, E1212, E1212, Qu(37, 24), M, N, K, kernel="mfma_i8_limb", reason="1 k-chunk(s) x 2 x 2", mul_args=Tags(25, 24),
             add_args=[Qu(37, 24)])
    if M * N > 1000:
        assert len(np.unique(c)) > 500
    _run(oracle, E1212, E1212, Qu(9, 3, True, RND.INF, SAT.ZERO), M, N, K, kernel="mfma_i8_limb", reason="2 x 2 limb groups", dist=1,
         mul_args=Tags(25, 24), add_args=[Qu(37, 24)])


def test_mixed_group_shapes(oracle):
    # 4 limbs x 1 limb, 1 x 4, 5 x 3 (groups 3 + 2 against 3), 3 x 5, unsigned 4-limb operands; transposed A, leading dimensions
    e55 = Qu(16, 16)            # 33 storage bits: 5 limbs, 8-byte host elements
    e20 = Qu(10, 10)            # 21 storage bits: 3 limbs
    u28 = Qu(14, 14, False)     # unsigned, 28 value bits: 4 limbs
    _run(oracle, E1212, E43, Qu(30, 15), 130, 70, 200, kernel="mfma_i8_limb", reason="x 2 x 1", mul_args=Tags(17, 15), add_args=[Qu(26, 15)])
    _run(oracle, E43, E1212, Qu(30, 15), 70, 130, 200, kernel="mfma_i8_limb", reason="x 1 x 2", ta=True, mul_args=Tags(17, 15), add_args=[Qu(26, 15)])
    _run(oracle, e55, e20, Qu(40, 20), 64, 48, 128, kernel="mfma_i8_limb", reason="x 2 x 1", lda=70, ldc=80, mul_args=Tags(27, 26), add_args=[Qu(35, 26)])
    _run(oracle, e20, e55, Qu(40, 11, True, RND.POS_INF), 48, 64, 64, kernel="mfma_i8_limb", reason="x 1 x 2", ldb=70, mul_args=Tags(27, 26),
         add_args=[Qu(33, 26)])
    _run(oracle, u28, u28, Qu(32, 28, False), 96, 40, 16, kernel="mfma_i8_limb", reason="x 2 x 2", mul_args=Tags(28, 28), add_args=[Qu(32, 28, False)])


def test_composite_through_the_resident_entry_points(oracle):
    """fill on the device (the generator the oracle restates), execute, unpack: the same bytes as the host-buffer call"""
    M, N, K = 256, 128, 70000
    d = lower(E88Z, E88Z, Qu(35, 16), M, N, K, mul_args=Tags(17, 16), add_args=[Qu(35, 16)])
    with capi.Context(0) as ctx:
        p = capi.Plan(ctx, d)
        assert b"2 k-chunk(s)" in p.info.reason
        b = p.info.packed_bytes
        pa, pb, pc = ctx.alloc(b[0]), ctx.alloc(b[1]), ctx.alloc(b[2])
        hc = ctx.alloc(M * N * p.info.host_elem_bytes[2])
        p.fill(capi.OPERAND_A, 1, 0, pa)
        p.fill(capi.OPERAND_B, 2, 0, pb)
        p.execute(pc, pa, pb)
        p.unpack_c(pc, hc)
        got = np.zeros(M * N, np.int64)
        ctx.d2h(got, hc)
        # a second execute must give the same result (the running sums start from the first chunk again)
        p.execute(pc, pa, pb)
        p.unpack_c(pc, hc)
        again = np.zeros(M * N, np.int64)
        ctx.d2h(again, hc)
        p.close()
        for x in (pa, pb, pc, hc):
            ctx.free(x)
    A = oracle.fill(E88Z, M * K, 1)
    B = oracle.fill(E88Z, K * N, 2)
    assert np.array_equal(got, again)
    rows = (0, 16)
    exp = oracle.gemm(d, A, B, Qu(35, 16), rows=rows, nthreads=8)
    assert np.array_equal(got.reshape(N, M)[:, :16], exp.reshape(N, M)[:, :16])
    # every packed C element equals what the one-shot host-buffer call returns
    full = capi.run(d, np.zeros(M * N, np.int64), A, B)
    assert np.array_equal(got, full)


def test_forced_tree_agrees(oracle):
    """the same descriptors through the exact tree kernel (what round 2 ran): identical results"""
    M, N, K = 48, 40, 512
    d = lower(E1212, E1212, Qu(20, 6, True, RND.CONV, SAT.SMGN), M, N, K, mul_args=Tags(25, 24), add_args=[Qu(35, 24)])
    A = oracle.fill(E1212, M * K, 5)
    B = oracle.fill(E1212, K * N, 6)
    a = capi.run(d, np.zeros(M * N, np.int32), A, B)
    b = capi.run(d, np.zeros(M * N, np.int32), A, B, flags=capi.OPT_FORCE_TREE)
    assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_FORCE_TREE).kernel] == "tree_i64"
    assert np.array_equal(a, b)
