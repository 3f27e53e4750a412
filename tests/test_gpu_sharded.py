"""qgemul_run_sharded (include/qgemul.h): ONE process, the rows of C in bands over a list of devices, packed bands moved to the
first device with peer copies, one unpack into the host-layout C.  The test box has one GPU, so the list names device 0 two or
three times: two / three contexts, plans and streams on one card exercise the partition, the per-device state, the event
ordering between streams and the reassembly (the peer copy degenerates to a device-to-device copy).  Checked against the
reference-generated golden GEMMs where they are big enough to split, and against the oracle / the one-device call elsewhere."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, Tags, TFComplexMul, desc_from_dict, lower

pytestmark = pytest.mark.gpu

E43 = Qu(4, 3)
E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_golden_gemms_through_the_sharded_entry(oracle, ndev):
    """Every reference-generated real GEMM: bands of 256 rows mean that most of them live in the first band and the other
    contexts get nothing — the degenerate partitions are part of the contract."""
    n = 0
    for j in G.gemm_cases("real") + G.gemm_cases("cplx"):
        d = desc_from_dict(j)
        if capi.classify_status(d)[0] != capi.QG_OK:
            continue
        A, B = G.case_inputs(j, oracle)
        exp = G.case_expected(j, oracle)
        out = np.zeros_like(exp)
        capi.run_sharded(d, out, A, B, [0] * ndev)
        assert out.tobytes() == exp.tobytes(), j["name"]
        n += 1
    assert n > 80


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("M,N,K,ndev", [(1000, 70, 96, 2), (700, 33, 200, 3), (2048, 256, 128, 3), (520, 40, 64, 2)])
def test_bands_against_the_oracle(oracle, M, N, K, ndev, ta):
    """Ragged M (the last band is short), transposed and plain A, padded leading dimensions; tree class and linear class."""
    rng = np.random.default_rng(M + N)
    Q78, U8 = Qu(7, 8), Qu(8, 0, False)   # (centred operands: every band packs its own rows of A — and their row sums)
    for ea, ec, kw in ((E88, E88, {}), (E43, Qu(16, 3), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)])),
                       (Q78, Qu(20, 8), dict(mul_args=Tags(15, 16), add_args=[Qu(27, 16)])), (U8, Qu(24, 0, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(24, 0, False)]))):
        d = lower(ea, ea, ec, M, N, K, transposed_a=ta, **kw)
        lda = (K if ta else M) + 3
        ldc = M + 5
        A = np.zeros(lda * (M if ta else K), np.int32)
        Av = rng.integers(ea.raw_min, ea.raw_max + 1, (M if ta else K, K if ta else M), dtype=np.int32)
        A.reshape(-1, lda)[:, :(K if ta else M)] = Av
        B = rng.integers(ea.raw_min, ea.raw_max + 1, K * N, dtype=np.int32)
        out = np.full(ldc * N, 77, np.int32)
        capi.run_sharded(d, out, A, B, [0] * ndev, lda=lda, ldc=ldc)
        exp = oracle.gemm(d, np.ascontiguousarray(Av).reshape(-1), B, ec, nthreads=8)
        o2 = out.reshape(N, ldc)
        assert np.array_equal(o2[:, :M], exp.reshape(N, M))
        assert (o2[:, M:] == 77).all()           # padding between the caller's columns is left alone
        one = np.zeros(M * N, np.int32)
        capi.run(d, one, np.ascontiguousarray(Av).reshape(-1), B)
        assert np.array_equal(one, exp)


def test_complex_bands(oracle):
    r, i = Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r, i)
    M, N, K = 600, 24, 64
    d = lower(c5, c5, c5, M, N, K, mul_args=TFComplexMul())
    A = oracle.fill(c5, M * K, 1, 1)
    B = oracle.fill(c5, K * N, 2, 1)
    out = np.zeros(M * N, dtype=oracle.host_dtype(c5))
    capi.run_sharded(d, out, A, B, [0, 0, 0])
    assert out.tobytes() == oracle.gemm(d, A, B, c5, nthreads=8).tobytes()


def test_all_devices_flag_and_cache_reuse(oracle):
    """QG_OPT_ALL_DEVICES through plain qgemul_run (what Qgemul<...> passes when QgemulRunFlags() asks for it), called in a
    loop with alternating shapes: the per-device caches must follow."""
    rng = np.random.default_rng(5)
    for M in (300, 1100, 300, 513):
        d = lower(E43, E43, Qu(16, 3), M, 50, 128, mul_args=Tags(9, 6), add_args=[Qu(19, 6)])
        A = rng.integers(E43.raw_min, E43.raw_max + 1, M * 128, dtype=np.int32)
        B = rng.integers(E43.raw_min, E43.raw_max + 1, 128 * 50, dtype=np.int32)
        out = capi.run(d, np.zeros(M * 50, np.int32), A, B, flags=capi.OPT_ALL_DEVICES)
        assert np.array_equal(out, oracle.gemm(d, A, B, Qu(16, 3), nthreads=4))
    capi.run_release()


def test_bad_device_lists():
    d = lower(E43, E43, E43, 8, 8, 8)
    z = np.zeros(64, np.int32)
    for devs in ([], [99], [0] * 17):
        with pytest.raises(capi.QgemulError) as ei:
            capi.run_sharded(d, z.copy(), z, z, devs)
        assert ei.value.status == capi.QG_EINVAL
