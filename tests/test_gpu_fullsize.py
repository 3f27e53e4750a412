"""BASELINE.json configurations at FULL size on the GPU (run with -m gpu).  The CPU restatement cannot
evaluate 4096^3 .. 16384x16384x4096 in test time, so these use
  * sampled row/column blocks of the oracle (oracle/qoracle.c evaluates any sub-block), and
  * size-independent properties of the domain: additivity of the linear class in A, invariance of the
    result under the kernel choice (MFMA limbs vs exact tree), row-shard consistency.
Operands are generated on the device by qgemul_fill_packed; the host copies used for the oracle come
from the oracle's own implementation of the same counter-based generator."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, Tags, TFComplexMul, lower

pytestmark = pytest.mark.gpu

E43 = Qu(4, 3)
E88Z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)


def run_resident(d, flags=0, seeds=(1, 2), dist=0):
    """fill -> execute -> unpack on the device, returns the host-layout C (column-major, flattened)."""
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d, flags)
        info = plan.info
        pb = info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        nbytes = d.M * d.N * info.host_elem_bytes[2]
        dC = ctx.alloc(nbytes)
        plan.fill(capi.OPERAND_A, seeds[0], dist, pA)
        plan.fill(capi.OPERAND_B, seeds[1], dist, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, dC)
        out = np.zeros(nbytes, np.uint8)
        ctx.d2h(out, dC)
        for p in (pA, pB, pC, dC):
            ctx.free(p)
        kernel = capi.KERNEL_NAMES[info.kernel]
        plan.close()
    return out, kernel


def check_block(oracle, d, ea, eb, ec, got_bytes, rows, cols, dist=0, seeds=(1, 2)):
    M, N, K = d.M, d.N, d.K
    A = oracle.fill(ea, M * K, seeds[0], dist)
    B = oracle.fill(eb, K * N, seeds[1], dist)
    cdt = oracle.host_dtype(ec)
    got = got_bytes.view(cdt)
    exp = np.zeros(M * N, dtype=cdt)
    oracle.gemm(d, A, B, ec, rows=rows, cols=cols, nthreads=16, out=exp)
    g2, e2 = got.reshape(N, M), exp.reshape(N, M)
    sl = (slice(cols[0], cols[1]), slice(rows[0], rows[1]))
    if cdt.names:
        for n in cdt.names:
            assert np.array_equal(g2[n][sl], e2[n][sl])
    else:
        assert np.array_equal(g2[sl], e2[sl])


def test_config3_4096_linear_limb_mfma(oracle):
    """Configuration 3 operands (4096^3, int<8,8>) in the linear class: 3x3 int8 limbs on MFMA."""
    ec = Qu(23, 8)
    d = lower(E88Z, E88Z, ec, 4096, 4096, 4096, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
    got, kernel = run_resident(d)
    assert kernel == "mfma_i8_limb"
    check_block(oracle, d, E88Z, E88Z, ec, got, rows=(1000, 1024), cols=(0, 512))
    check_block(oracle, d, E88Z, E88Z, ec, got, rows=(4090, 4096), cols=(3584, 4096))
    # kernel-choice invariance on the whole matrix: the exact tree kernel must give the same C
    got_t, kernel_t = run_resident(d, flags=capi.OPT_FORCE_TREE)
    assert kernel_t == "tree_i64" or kernel_t == "tree_i32"
    assert np.array_equal(got, got_t)


def test_config3_4096_tree_default_tags(oracle):
    """Configuration 3 as literally configured: default tags -> per-product and per-node quantisation."""
    d = lower(E88Z, E88Z, E88Z, 4096, 4096, 4096)
    got, kernel = run_resident(d, dist=1)
    assert kernel == "tree_i32"
    check_block(oracle, d, E88Z, E88Z, E88Z, got, rows=(2040, 2056), cols=(100, 356), dist=1)
    c = got.view(np.int32)
    assert 0.02 < float(np.mean(c != 0)) < 1.0  # not everything was zeroed by SAT::ZERO


def test_config2_1024_tree_and_linear(oracle):
    d = lower(E43, E43, Qu(16, 3), 1024, 1024, 1024)
    got, kernel = run_resident(d, dist=1)
    assert kernel == "tree_i32"
    check_block(oracle, d, E43, E43, Qu(16, 3), got, rows=(0, 1024), cols=(500, 532), dist=1)


def test_config4_16384_rowshards(oracle):
    """Configuration 4 (16384 x 16384 x 4096, int<4,3>, AddArgs<Qu<21,6>>): sampled blocks vs the oracle, and EVERY one of the
    eight 2048-row shards computed on its own (what each of 8 GPUs does) equals the same rows of the full product."""
    ec = Qu(16, 3)
    M = N = 16384
    K = 4096
    kw = dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
    d = lower(E43, E43, ec, M, N, K, **kw)
    got, kernel = run_resident(d)
    assert kernel == "mfma_i8"
    check_block(oracle, d, E43, E43, ec, got, rows=(5000, 5016), cols=(0, 2048))
    check_block(oracle, d, E43, E43, ec, got, rows=(16380, 16384), cols=(14336, 16384))
    full = got.view(np.int32).reshape(N, M)
    A = oracle.fill(E43, M * K, 1)
    B = oracle.fill(E43, K * N, 2)
    ds = lower(E43, E43, ec, 2048, N, K, **kw)
    out = np.zeros(2048 * N, np.int32)
    for shard in range(8):          # rows [2048 shard, 2048 (shard + 1)) — packed directly from the host view of those rows
        capi.run(ds, out, A[2048 * shard:], B, lda=M)
        assert np.array_equal(out.reshape(N, 2048), full[:, 2048 * shard:2048 * (shard + 1)]), shard


def test_config4_narrow_c_as_configured(oracle):
    """Configuration 4 with ITS OWN C type (int<4,3>: the 1-byte packed container that travels in the gather): the whole
    16384 x 16384 product on the two-group kernel and on the lock-step kernel agree, blocks agree with the oracle, and the
    result is not all saturation (half-range operands)."""
    M = N = 16384
    K = 4096
    d = lower(E43, E43, E43, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
    got, kernel = run_resident(d, dist=1)
    assert kernel == "mfma_i8"
    check_block(oracle, d, E43, E43, E43, got, rows=(0, 8), cols=(0, 1024), dist=1)
    check_block(oracle, d, E43, E43, E43, got, rows=(8191, 8195), cols=(8000, 9024), dist=1)
    check_block(oracle, d, E43, E43, E43, got, rows=(16376, 16384), cols=(15360, 16384), dist=1)
    got_l, _ = run_resident(d, flags=capi.OPT_LOCKSTEP_TILES, dist=1)
    assert np.array_equal(got, got_l)
    c = got.view(np.int32)
    assert float(np.mean((c == E43.raw_max) | (c == E43.raw_min))) < 0.9


R63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
I63N = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
C5 = Qcomplex(R63, I63N)


@pytest.mark.parametrize("cname", ["narrow", "wide"])
def test_config5_2048_complex_tf_full_size(oracle, cname):
    """Configuration 5 at FULL size, M = N = K = 2048 (the 2048-row grid, not a band): the configuration's own narrow C
    (Qcomplex<int<6,3>,int<6,-3>>) and a wide C that does not saturate; bands at the first rows, in the middle and at the last
    rows against the oracle; and the whole matrix from the fixed-mode kernel variant equals the run-time-mode variant."""
    ec = C5 if cname == "narrow" else Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
    d = lower(C5, C5, ec, 2048, 2048, 2048, mul_args=TFComplexMul())
    got, kernel = run_resident(d, dist=1)
    assert kernel == "tree_cplx_i32"
    for rows in ((0, 8), (1000, 1008), (2040, 2048)):
        check_block(oracle, d, C5, C5, ec, got, rows=rows, cols=(0, 2048), dist=1)
    got_rt, kernel_rt = run_resident(d, flags=capi.OPT_RUNTIME_MODES, dist=1)
    assert kernel_rt == "tree_cplx_i32"
    assert np.array_equal(got, got_rt)
    if cname == "wide":
        v = got.view(oracle.host_dtype(ec))
        assert 0.5 < float(np.mean(v["re"] != 0))      # a real comparison, not zeros


def test_linear_class_additivity(oracle):
    """Size-independent property of the linear class: with a C wide enough not to saturate,
    C(A1 + A2, B) = C(A1, B) + C(A2, B) exactly (no intermediate rounding exists in this class)."""
    M = N = K = 2048
    ea = Qu(4, 2)           # 7-bit operands so that A1 + A2 stays inside int<4,3>-sized storage
    ec = Qu(24, 5)
    kw = dict(mul_args=Tags(10, 5), add_args=[Qu(23, 5)])
    d = lower(ea, E43, ec, M, N, K, **kw)
    A1 = oracle.fill(ea, M * K, 11)
    A2 = oracle.fill(ea, M * K, 12)
    B = oracle.fill(E43, K * N, 13)
    c1 = capi.run(d, np.zeros(M * N, np.int32), A1, B).astype(np.int64)
    c2 = capi.run(d, np.zeros(M * N, np.int32), A2, B).astype(np.int64)
    ds = lower(Qu(5, 2), E43, ec, M, N, K, **kw)
    c12 = capi.run(ds, np.zeros(M * N, np.int32), A1 + A2, B).astype(np.int64)
    assert np.array_equal(c12, c1 + c2)
