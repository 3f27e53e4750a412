"""Quantise-on-load: double matrices handed to the engine are converted on the device exactly as
Qu_s(double) does (QuBLAS.h:2387-2393) and packed in the same pass.  Checked against the oracle's from_double, which is
pinned by the reference's own construction tables (tests/test_from_double.py).  -m gpu."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, WRP, Tags, TFComplexMul, lower

pytestmark = pytest.mark.gpu


def quantize(oracle, e: Qu, x: np.ndarray) -> np.ndarray:
    L = oracle.lib()
    return np.array([L.qoracle_from_double(float(v), e.c()) for v in x], dtype=np.int64)


def awkward_doubles(rng, n, scale):
    x = rng.standard_normal(n) * scale
    x[rng.integers(0, n, n // 8)] = np.round(x[rng.integers(0, n, n // 8)] * 8) / 8 + 0.0625   # exact ties
    x[rng.integers(0, n, 16)] = [0.0, -0.0, 1e-320, -1e-320, 1e300, -1e300, np.inf, -np.inf, np.nan, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 1e-9]
    return x


@pytest.mark.parametrize("ea,eb", [
    (Qu(4, 3), Qu(4, 3)),
    (Qu(8, 8, True, RND.CONV, SAT.ZERO), Qu(6, 2, False, RND.POS_INF, SAT.SMGN)),
    (Qu(6, -3, True, TRN.SMGN, WRP.TCPL), Qu(3, 5, True, RND.INF, SAT.TCPL)),
    (Qu(5, 4, True, RND.ZERO, WRP.TCPL), Qu(5, 4, True, RND.NEG_INF, SAT.ZERO)),
])
@pytest.mark.parametrize("ta", [False, True])
def test_pack_f64_then_gemm(oracle, ea, eb, ta):
    rng = np.random.default_rng(7)
    M, N, K = 70, 45, 96
    Ad = awkward_doubles(rng, M * K, 6.0)
    Bd = awkward_doubles(rng, K * N, 6.0)
    ec = Qu(20, 6)
    d = lower(ea, eb, ec, M, N, K, transposed_a=ta)
    Ai = quantize(oracle, ea, Ad).astype(oracle.host_dtype(ea))
    Bi = quantize(oracle, eb, Bd).astype(oracle.host_dtype(eb))
    exp = oracle.gemm(d, Ai, Bi, ec)
    conv = RND.CONV in (ea.QuMode, eb.QuMode)
    with capi.Context() as ctx:
        # an RND::CONV element type: the arithmetic conversion only on request (QG_OPT_ARITHMETIC_CONV); by default the
        # pack is refused, because the reference's double construction under CONV is an artefact the engine does not imitate
        plan = capi.Plan(ctx, d, capi.OPT_ARITHMETIC_CONV if conv else 0)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        dA, dB, dC = ctx.alloc(Ad.nbytes), ctx.alloc(Bd.nbytes), ctx.alloc(M * N * 4)
        ctx.h2d(dA, Ad)
        ctx.h2d(dB, Bd)
        if conv:
            gated = capi.Plan(ctx, d)
            with pytest.raises(capi.QgemulError) as ei:
                gated.pack_f64(capi.OPERAND_A if ea.QuMode == RND.CONV else capi.OPERAND_B, dA if ea.QuMode == RND.CONV else dB, pA)
            assert ei.value.status == capi.QG_EUNSUPPORTED
            gated.close()
        plan.pack_f64(capi.OPERAND_A, dA, pA)
        plan.pack_f64(capi.OPERAND_B, dB, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, dC)
        got = np.zeros(M * N, np.int32)
        ctx.d2h(got, dC)
        for p in (pA, pB, pC, dA, dB, dC):
            ctx.free(p)
        plan.close()
    assert np.array_equal(got, exp)


def test_pack_f64_complex_and_limbs(oracle):
    rng = np.random.default_rng(11)
    r, i = Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r, i)
    wide = Qcomplex(Qu(18, 6), Qu(18, 6))
    M, N, K = 40, 33, 64
    Ad = awkward_doubles(rng, 2 * M * K, 20.0)   # {re, im} pairs
    Bd = awkward_doubles(rng, 2 * K * N, 20.0)
    d = lower(c5, c5, wide, M, N, K, mul_args=TFComplexMul())
    A = np.zeros(M * K, dtype=oracle.host_dtype(c5))
    B = np.zeros(K * N, dtype=oracle.host_dtype(c5))
    A["re"], A["im"] = quantize(oracle, r, Ad[0::2]), quantize(oracle, i, Ad[1::2])
    B["re"], B["im"] = quantize(oracle, r, Bd[0::2]), quantize(oracle, i, Bd[1::2])
    exp = oracle.gemm(d, A, B, wide)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        dA, dB, dC = ctx.alloc(Ad.nbytes), ctx.alloc(Bd.nbytes), ctx.alloc(exp.nbytes)
        ctx.h2d(dA, Ad); ctx.h2d(dB, Bd)
        plan.pack_f64(capi.OPERAND_A, dA, pA)
        plan.pack_f64(capi.OPERAND_B, dB, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, dC)
        got = np.zeros(M * N, dtype=exp.dtype)
        ctx.d2h(got, dC)
        plan.close()
    assert np.array_equal(got["re"], exp["re"]) and np.array_equal(got["im"], exp["im"])
    # linear class: int<8,8> doubles -> 3 int8 limbs straight from the doubles
    e = Qu(8, 8)
    ec = Qu(23, 8)
    M, N, K = 128, 128, 128
    Ad = awkward_doubles(rng, M * K, 60.0)
    Bd = awkward_doubles(rng, K * N, 60.0)
    d = lower(e, e, ec, M, N, K, mul_args=Tags(17, 16), add_args=[Qu(27, 16)])
    exp = oracle.gemm(d, quantize(oracle, e, Ad).astype(np.int32), quantize(oracle, e, Bd).astype(np.int32), ec)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d)
        assert capi.KERNEL_NAMES[plan.info.kernel] == "mfma_i8_limb"
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        dA, dB, dC = ctx.alloc(Ad.nbytes), ctx.alloc(Bd.nbytes), ctx.alloc(M * N * 4)
        ctx.h2d(dA, Ad); ctx.h2d(dB, Bd)
        plan.pack_f64(capi.OPERAND_A, dA, pA)
        plan.pack_f64(capi.OPERAND_B, dB, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, dC)
        got = np.zeros(M * N, np.int32)
        ctx.d2h(got, dC)
        plan.close()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("e,M,K,pad", [
    (Qu(8, 8, True, RND.POS_INF, SAT.ZERO), 300, 200, 0),     # 3 limbs
    (Qu(8, 8, True, TRN.SMGN, WRP.TCPL), 257, 130, 3),        # odd leading dimension: scalar loads
    (Qu(4, 3, True, RND.INF, SAT.SMGN), 515, 1000, 2),        # 1 limb, ragged K
    (Qu(7, 7, False, RND.ZERO, SAT.TCPL), 128, 256, 4),       # 2 limbs, unsigned
    (Qu(4, 3), 4096, 256, 0),                                 # the 256-row / 128-byte k-tile layout
])
def test_fast_pack_f64_writes_the_bytes_of_the_generic_kernel(e, M, K, pad, ta):
    """k_pack_limb32<F64> (both axis orders, 16-byte and scalar loads) against k_pack's quantise-on-load path
    (QG_OPT_GENERIC_LAYOUT): the same packed bytes for awkward doubles (ties, subnormals, huge values, NaN, infinities)."""
    rng = np.random.default_rng(M + K)
    ld = (K if ta else M) + pad
    cols = M if ta else K
    x = np.zeros(ld * cols)
    x.reshape(cols, ld)[:, :(K if ta else M)] = awkward_doubles(rng, M * K, 2.0 ** (e.intBits - 1)).reshape(cols, -1)
    N = 4096 if M == 4096 else 64
    d = lower(e, e, Qu(20, 6), M, N, K, mul_args=Tags(e.intBits * 2 + 1, e.fracBits * 2), add_args=[Qu(e.intBits * 2 + 13, e.fracBits * 2)], transposed_a=ta)
    outs = []
    with capi.Context() as ctx:
        dX = ctx.alloc(x.nbytes)
        ctx.h2d(dX, x)
        for flags in (0, capi.OPT_GENERIC_LAYOUT):
            plan = capi.Plan(ctx, d, flags)
            assert capi.KERNEL_NAMES[plan.info.kernel].startswith("mfma_i8")
            nb = int(plan.info.packed_bytes[0])
            pA = ctx.alloc(nb)
            ctx.h2d(pA, np.full(nb, 0xee, np.uint8))
            plan.pack_f64(capi.OPERAND_A, dX, pA, ld)
            ctx.sync()
            buf = np.zeros(nb, np.uint8)
            ctx.d2h(buf, pA)
            if plan.info.limbs[0] > 1:      # the plane mask is the OR of the trailer's 64 words
                tr = buf[-256:].view(np.uint32)
                m = np.bitwise_or.reduce(tr)
                tr[:] = 0
                tr[0] = m
            outs.append(buf)
            ctx.free(pA)
            plan.close()
        ctx.free(dX)
    assert np.array_equal(outs[0], outs[1])
    assert np.count_nonzero(outs[0]) > 0.3 * M * K
