"""GPU parity of the BitStream export through the C-ABI: every golden string of the reference, fed
through a GEMM with the identity matrix whose result IS the fixture's tensor; a 1024^2 result against the oracle; the
packed-bit form; a plan with an epilogue exports D."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import Ew, Qu, RND, SAT, TRN, Tags, lower, lower_epilogue

pytestmark = pytest.mark.gpu

ONE = Qu(1, 0, False)
CASES = G.bitstream_cases()


def _resident_gemm(ctx, d, A, B, epilogue=None):
    plan = capi.Plan(ctx, d, epilogue=epilogue)
    dA, dB = ctx.alloc(A.nbytes), ctx.alloc(B.nbytes)
    ctx.h2d(dA, A); ctx.h2d(dB, B)
    pA, pB, pC = (ctx.alloc(int(plan.info.packed_bytes[i])) for i in range(3))
    plan.pack(capi.OPERAND_A, dA, pA); plan.pack(capi.OPERAND_B, dB, pB)
    return plan, pA, pB, pC


def _export(ctx, plan, pC, tc, ec, fmt=capi.BITS_ASCII):
    nb = plan.bitstream_bytes(fmt)
    dev = ctx.alloc(nb)
    plan.export_bitstream(pC, dev, tc, ec, fmt)
    out = np.zeros(nb, dtype=np.uint8)
    ctx.d2h(out, dev)
    ctx.free(dev)
    return out.tobytes()


@pytest.mark.parametrize("j", CASES, ids=lambda j: j["name"])
def test_golden_strings_through_identity_gemm(oracle, j):
    f = Qu.from_tuple(j["fmt"])
    M, N = j["rows"], j["n"] // j["rows"]
    d = lower(f, ONE, f, M, N, N, mul_args=f, add_args=[f])      # C = X * I, every node exact
    X = np.asarray(j["X"], dtype=np.int64).astype(oracle.host_dtype(f))
    I = np.eye(N, dtype=np.int32).reshape(-1)
    with capi.Context() as ctx:
        plan, pA, pB, pC = _resident_gemm(ctx, d, X, I)
        plan.execute(pC, pA, pB)
        got = _export(ctx, plan, pC, j["tensor_chunk"], j["elem_chunk"])
        assert got.decode() == j["bits"]
        packed = _export(ctx, plan, pC, j["tensor_chunk"], j["elem_chunk"], capi.BITS_PACKED)
        bits = np.unpackbits(np.frombuffer(packed, dtype=np.uint8))[:len(j["bits"])]
        assert "".join("01"[b] for b in bits) == j["bits"]
        w = f.intBits + f.fracBits + int(f.isSigned)
        if w % 4:
            with pytest.raises(capi.QgemulError):
                _export(ctx, plan, pC, 0, 4)
        plan.close()


def test_bench_shaped_result_against_oracle(oracle):
    e88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    cq = Qu(23, 8)
    M = N = 1024
    K = 128
    d = lower(e88, e88, cq, M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
    A, B = oracle.fill(e88, M * K, 1, 0), oracle.fill(e88, K * N, 2, 0)
    exp_c = oracle.gemm(d, A, B, cq, nthreads=8).astype(np.int64)
    with capi.Context() as ctx:
        plan, pA, pB, pC = _resident_gemm(ctx, d, A, B)
        plan.execute(pC, pA, pB)
        for tc, ec in ((0, 0), (1024, 8), (1, 32), (4, 1)):
            assert _export(ctx, plan, pC, tc, ec) == oracle.bitstream(cq, exp_c, tc, ec), (tc, ec)
        packed = _export(ctx, plan, pC, 1024, 8, capi.BITS_PACKED)
        ref = oracle.bitstream(cq, exp_c, 1024, 8)
        assert np.array_equal(np.unpackbits(np.frombuffer(packed, dtype=np.uint8))[:len(ref)], np.frombuffer(ref, dtype=np.uint8) - ord("0"))
        plan.close()


def test_plan_with_epilogue_exports_d(oracle):
    e43 = Qu(4, 3)
    cq, dq = Qu(15, 8), Qu(6, 2, False)
    M, N, K = 96, 40, 64
    d = lower(e43, e43, cq, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
    ep = lower_epilogue(cq, [Ew("add", Qu(3, 4), scalar=True)], dq)
    A, B = oracle.fill(e43, M * K, 1, 0), oracle.fill(e43, K * N, 2, 0)
    exp_d = oracle.eltwise(ep, cq, oracle.gemm(d, A, B, cq).astype(np.int64), [np.array([5])])
    with capi.Context() as ctx:
        plan, pA, pB, pD = _resident_gemm(ctx, d, A, B, epilogue=ep)
        plan.execute_ep(pD, pA, pB, plan.ep_args(scalars=[5]))
        assert plan.bitstream_bytes() == M * N * 8            # unsigned int<6,2>: 8 characters, no sign bit
        assert _export(ctx, plan, pD, 2, 4) == oracle.bitstream(dq, exp_d, 2, 4)
        plan.close()


# ---- complex tensors: "(re-bits, im-bits)" per element (QuBLAS.h:2553-2556) ----
CCASES = G.cplx_bitstream_cases()


def _binary_only(text: str) -> str:
    return "".join(ch for ch in text if ch in "01")


@pytest.mark.parametrize("j", CCASES, ids=lambda j: j["name"])
def test_complex_golden_strings_through_identity_gemm(oracle, j):
    """C = X * (1 + 0i) with every product sub-operation in C's own part formats reproduces the fixture's tensor."""
    from qublas_amd.desc import BasicComplexMul, Qcomplex
    f = Qcomplex(Qu.from_tuple(j["fmt"][0]), Qu.from_tuple(j["fmt"][1]))
    cone = Qcomplex(ONE, ONE)
    M, N = j["rows"], j["n"] // j["rows"]
    d = lower(f, cone, f, M, N, N, add_args=[f],
              mul_args=BasicComplexMul(acT=f.real, bdT=f.imag, adT=f.real, bcT=f.imag, acbdT=f.real, adbcT=f.imag))
    X = np.zeros(M * N, dtype=oracle.host_dtype(f))
    X["re"], X["im"] = j["Xre"], j["Xim"]
    I = np.zeros(N * N, dtype=oracle.host_dtype(cone))
    I["re"] = np.eye(N, dtype=np.int32).reshape(-1)
    with capi.Context() as ctx:
        plan, pA, pB, pC = _resident_gemm(ctx, d, X, I)
        plan.execute(pC, pA, pB)
        got = _export(ctx, plan, pC, j["tensor_chunk"], j["elem_chunk"])
        assert got.decode() == j["bits"]
        # packed: the binary characters of the same stream
        want = _binary_only(j["bits"])
        packed = _export(ctx, plan, pC, j["tensor_chunk"], j["elem_chunk"], capi.BITS_PACKED)
        assert len(packed) == ((len(want) + 7) // 8 + 3) // 4 * 4
        bits = np.unpackbits(np.frombuffer(packed, dtype=np.uint8))
        assert "".join("01"[b] for b in bits[:len(want)]) == want and not bits[len(want):].any()
        w = len(j["bits"]) // j["n"]
        if w % 7:
            with pytest.raises(capi.QgemulError):
                _export(ctx, plan, pC, 0, 7)
        plan.close()


def test_complex_result_and_complex_chain_against_oracle(oracle):
    """configuration 5's element type through TFComplexMul: the exported string of C, and of D after a complex chain."""
    from qublas_amd.desc import EwC, Qcomplex, TFComplexMul, lower_epilogue_cplx
    c5 = Qcomplex(Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL))
    dq = Qcomplex(Qu(10, 4, True, RND.CONV, SAT.SMGN), Qu(20, 12))          # 15 + 33 + 4 = 52 characters
    M, N, K = 96, 40, 64
    d = lower(c5, c5, c5, M, N, K, mul_args=TFComplexMul())
    A, B = oracle.fill(c5, M * K, 1, 0), oracle.fill(c5, K * N, 2, 0)
    Cx = oracle.gemm(d, A, B, c5, nthreads=8)
    cre, cim = Cx["re"].astype(np.int64), Cx["im"].astype(np.int64)
    with capi.Context() as ctx:
        plan, pA, pB, pC = _resident_gemm(ctx, d, A, B)
        plan.execute(pC, pA, pB)
        assert plan.bitstream_bytes() == M * N * 18
        for tc, ec in ((0, 0), (96, 9), (1, 1), (4, 6)):
            assert _export(ctx, plan, pC, tc, ec) == oracle.bitstream_cplx(c5, cre, cim, tc, ec), (tc, ec)
        plan.close()
        epc = lower_epilogue_cplx(c5, [EwC("mul", Qu(2, 2), scalar=True), EwC("add", Qcomplex(Qu(5, 4), Qu(3, 2)), scalar=True)], dq)
        dre, dim = oracle.eltwise_cplx(epc, c5, cre, cim, [np.array([-3]), np.array([100])], [np.array([-3]), np.array([-9])])
        plan, pA, pB, pD = _resident_gemm(ctx, d, A, B, epilogue=epc)
        plan.execute_ep(pD, pA, pB, plan.ep_args(scalars=[-3, 100], scalars_im=[-3, -9]))
        assert plan.bitstream_bytes() == M * N * 52
        for tc, ec in ((0, 0), (2, 13), (40, 4)):
            assert _export(ctx, plan, pD, tc, ec) == oracle.bitstream_cplx(dq, dre, dim, tc, ec), (tc, ec)
        want = _binary_only(oracle.bitstream_cplx(dq, dre, dim, 2, 13).decode())
        packed = _export(ctx, plan, pD, 2, 13, capi.BITS_PACKED)
        bits = np.unpackbits(np.frombuffer(packed, dtype=np.uint8))[:len(want)]
        assert np.array_equal(bits, np.frombuffer(want.encode(), dtype=np.uint8) - ord("0"))
        plan.close()
