"""GPU parity of element-wise operators after a COMPLEX Qgemul (include/qgemul.h, qgemul_epilogue_cplx), through the C-ABI:
  * every golden vector of the reference's lazy tensor operators on complex tensors (tests/golden/ref_cplx_eltwise_*),
    fed through a K = 1 complex GEMM whose result IS the fixture's X tensor;
  * complex GEMMs (exact tree kernels, the fixed-mode 32-bit kernel, the stacked MFMA linear class) + chains against
    oracle GEMM + oracle chains, by the one-shot host entry and by the resident-data entry points."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import (BasicComplexMul, EwC, Qcomplex, Qu, RND, SAT, TRN, WRP, Tags, TFComplexMul, host_layout, lower,
                             lower_epilogue_cplx)

pytestmark = pytest.mark.gpu

CASES = G.cplx_eltwise_cases()
ONE = Qu(1, 0, False)
CONE = Qcomplex(ONE, ONE)


def host_elems(oracle, e, re, im=None):
    """host-layout array of element type e from raw values (structured {re, im} for a complex type)"""
    out = np.zeros(len(re), dtype=oracle.host_dtype(e))
    if isinstance(e, Qcomplex):
        out["re"], out["im"] = re, im
    else:
        out[:] = re
    return out


def run_epc(oracle, d, epc, A, B, E, dq, flags=0, ldc=0):
    out = np.zeros((ldc or d.M) * d.N, dtype=oracle.host_dtype(dq))
    return capi.run_ep(d, epc, out, A, B, E, flags=flags, ldc=ldc)


def operand_type(s):
    e = [Qu.from_tuple(t) for t in s["e"]]
    return Qcomplex(e[0], e[1]) if s["e_complex"] else e[0]


@pytest.mark.parametrize("j", CASES, ids=lambda j: j["name"])
def test_golden_vectors_through_identity_gemm(oracle, j):
    """C = X * (1 + 0i) with K = 1: re = a*1 - b*0 and im = a*0 + b*1 in C's own part formats reproduce the fixture's
    tensor exactly, so D must equal the reference's D."""
    epc, c, _, _ = G.cplx_eltwise_epilogue(j)
    n = j["n"]
    d = lower(c, CONE, c, n, 1, 1, mul_args=BasicComplexMul(acT=c.real, bdT=c.imag, adT=c.real, bcT=c.imag, acbdT=c.real, adbcT=c.imag))
    A = host_elems(oracle, c, j["Xre"], j["Xim"])
    B = host_elems(oracle, CONE, [1], [0])
    dq = Qcomplex(Qu.from_tuple(j["d"][0]), Qu.from_tuple(j["d"][1]))
    Eh = [host_elems(oracle, operand_type(s), s["Ere"], s["Eim"]) for s in j["stages"]]
    got = run_epc(oracle, d, epc, A, B, Eh, dq)
    assert np.array_equal(got["re"].astype(np.int64), np.asarray(j["Dre"], dtype=np.int64)), j["name"]
    assert np.array_equal(got["im"].astype(np.int64), np.asarray(j["Dim"], dtype=np.int64)), j["name"]
    again = run_epc(oracle, d, epc, A, B, Eh, dq, flags=capi.OPT_GENERIC_TREE)
    assert np.array_equal(again["re"], got["re"]) and np.array_equal(again["im"], got["im"])


R63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
R6N3 = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
C5 = Qcomplex(R63, R6N3)
R54, R32, S22 = Qu(5, 4), Qu(3, 2), Qu(2, 2)
CB = Qcomplex(R54, R32)
CD = Qcomplex(Qu(10, 4, True, RND.CONV, SAT.SMGN), Qu(8, 2, True, TRN.TCPL, SAT.ZERO))
CQ = Qcomplex(Qu(7, 3, True, RND.ZERO, WRP.TCPL), Qu(9, 1, True, TRN.SMGN, SAT.SMGN))
CM = Qcomplex(Qu(20, 6), Qu(30, 10))                    # {int32 re; int64 im}
WIDE = Qcomplex(Qu(18, 6, True, RND.POS_INF), Qu(18, 6, True, RND.POS_INF))
CHAINS = {
    "cbias": ([EwC("add", CB)], CD),
    "scale_cbias": ([EwC("mul", S22, real_tags=Tags(20, 6), imag_tags=Tags(20, 6), scalar=True, into=WIDE), EwC("add", CB)], CQ),
    "real_tensor_minus": ([EwC("sub", R54, x_first=False)], CD),
    "real_bias_then_csub_scalar_then_scale": ([EwC("add", R54, into=WIDE), EwC("sub", CB, x_first=False, scalar=True, into=CD),
                                               EwC("mul", R32, imag_tags=Qu(12, 3), x_first=False)], CM),
    "four_stages": ([EwC("add", CB, tags=Tags(FullPrec=True)), EwC("sub", R32, scalar=True), EwC("mul", S22), EwC("add", CM, scalar=True)], CM),
    "convert_only": ([], CQ),
}
BL = BasicComplexMul(acT=Qu(14, 6), bdT=Qu(14, -6), adT=Qu(14, 0), bcT=Qu(14, 0), acbdT=Qu(15, 6), adbcT=Qu(15, 0))
GEMMS = {
    # name: (C type, M, N, K, lowering keywords, expected kernel)
    "tf_fixed_modes": (C5, 70, 33, 256, dict(mul_args=TFComplexMul()), "tree_cplx_i32"),
    "basic_default": (WIDE, 40, 24, 64, dict(), None),
    "linear_mfma": (WIDE, 200, 130, 512, dict(mul_args=BL, add_args=[Qcomplex(Qu(30, 6), Qu(30, 0))]), "mfma_cplx"),
}


def _operands(oracle, stages, n, seed0=90):
    """host arrays for the call and the per-part value lists for the oracle"""
    Eh, Ere, Eim = [], [], []
    for k, st in enumerate(stages):
        m = 1 if st.scalar else n
        h = oracle.fill(st.e, m, seed0 + k, 0)
        Eh.append(h)
        if isinstance(st.e, Qcomplex):
            Ere.append(h["re"].astype(np.int64))
            Eim.append(h["im"].astype(np.int64))
        else:
            Ere.append(h.astype(np.int64))
            # the imaginary parts' stage: the real operand (Qmul), nothing (carried over), or the zero of its type (real - complex)
            Eim.append(h.astype(np.int64) if st.op == "mul" else np.zeros(1, dtype=np.int64))
    return Eh, Ere, Eim


@pytest.mark.parametrize("chain", sorted(CHAINS))
@pytest.mark.parametrize("gemm", sorted(GEMMS))
def test_complex_gemm_plus_chain_vs_oracle(oracle, gemm, chain):
    ec, M, N, K, kw, kern = GEMMS[gemm]
    stages, dq = CHAINS[chain]
    d = lower(C5, C5, ec, M, N, K, **kw)
    epc = lower_epilogue_cplx(ec, stages, dq)
    st, info = capi.classify_ep_status(d, epc)
    assert st == capi.QG_OK, info.reason
    if kern:
        assert capi.KERNEL_NAMES[info.kernel] == kern
    A, B = oracle.fill(C5, M * K, 1, 0), oracle.fill(C5, K * N, 2, 0)
    Eh, Ere, Eim = _operands(oracle, stages, M * N)
    got = run_epc(oracle, d, epc, A, B, Eh, dq)
    Cx = oracle.gemm(d, A, B, ec, nthreads=8)
    exp_re, exp_im = oracle.eltwise_cplx(epc, ec, Cx["re"].astype(np.int64), Cx["im"].astype(np.int64), Ere, Eim)
    assert np.array_equal(got["re"].astype(np.int64), exp_re)
    assert np.array_equal(got["im"].astype(np.int64), exp_im)
    assert len(np.unique(got["re"])) > 8 and len(np.unique(got["im"])) > 8   # not hidden by saturation
    # a padded destination (ldc > M) keeps its padding
    ldc = M + 5
    pad = run_epc(oracle, d, epc, A, B, Eh, dq, ldc=ldc)
    assert np.array_equal(pad.reshape(N, ldc)[:, :M].reshape(-1), got)


def test_resident_entry_points(oracle):
    """qgemul_plan_create_epc / qgemul_pack_e / qgemul_execute_ep / qgemul_unpack_c on device-resident data; the plan never
    runs the chain inside the GEMM kernel."""
    ec, M, N, K, kw, _ = GEMMS["tf_fixed_modes"]
    stages, dq = CHAINS["real_bias_then_csub_scalar_then_scale"]
    d = lower(C5, C5, ec, M, N, K, **kw)
    epc = lower_epilogue_cplx(ec, stages, dq)
    A, B = oracle.fill(C5, M * K, 3, 0), oracle.fill(C5, K * N, 4, 0)
    Eh, Ere, Eim = _operands(oracle, stages, M * N, seed0=40)
    Cx = oracle.gemm(d, A, B, ec, nthreads=8)
    exp_re, exp_im = oracle.eltwise_cplx(epc, ec, Cx["re"].astype(np.int64), Cx["im"].astype(np.int64), Ere, Eim)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d, epilogue=epc)
        assert not plan.fuses_epilogue()
        dA, dB = ctx.alloc(A.nbytes), ctx.alloc(B.nbytes)
        ctx.h2d(dA, A); ctx.h2d(dB, B)
        pA, pB, pD = (ctx.alloc(int(plan.info.packed_bytes[i])) for i in range(3))
        plan.pack(capi.OPERAND_A, dA, pA); plan.pack(capi.OPERAND_B, dB, pB)
        packed, sc_re, sc_im = [], [], []
        for k, st in enumerate(stages):
            if st.scalar:
                assert plan.packed_e_bytes(k) == 0
                packed.append(0)
                sc_re.append(int(Ere[k][0])); sc_im.append(int(Eim[k][0]))
                continue
            nb = plan.packed_e_bytes(k)
            assert nb > 0
            dE, pE = ctx.alloc(Eh[k].nbytes), ctx.alloc(nb)
            ctx.h2d(dE, Eh[k])
            plan.pack_e(k, dE, pE)
            packed.append(pE); sc_re.append(0); sc_im.append(0)
        plan.execute_ep(pD, pA, pB, plan.ep_args(packed=packed, scalars=sc_re, scalars_im=sc_im))
        size, _, _ = host_layout(dq)
        assert plan.info.host_elem_bytes[2] == size
        dD = ctx.alloc(M * N * size)
        plan.unpack_c(pD, dD)
        out = np.zeros(M * N, dtype=oracle.host_dtype(dq))
        ctx.d2h(out, dD)
        plan.close()
    assert np.array_equal(out["re"].astype(np.int64), exp_re) and np.array_equal(out["im"].astype(np.int64), exp_im)


def test_repeated_one_shot_calls_rebuild_the_plan_when_the_chain_changes(oracle):
    ec, M, N, K, kw, _ = GEMMS["basic_default"]
    d = lower(C5, C5, ec, M, N, K, **kw)
    A, B = oracle.fill(C5, M * K, 5, 0), oracle.fill(C5, K * N, 6, 0)
    Cx = oracle.gemm(d, A, B, ec, nthreads=8)
    for name in ("cbias", "real_tensor_minus", "cbias"):
        stages, dq = CHAINS[name]
        epc = lower_epilogue_cplx(ec, stages, dq)
        Eh, Ere, Eim = _operands(oracle, stages, M * N)
        got = run_epc(oracle, d, epc, A, B, Eh, dq)
        exp_re, exp_im = oracle.eltwise_cplx(epc, ec, Cx["re"].astype(np.int64), Cx["im"].astype(np.int64), Ere, Eim)
        assert np.array_equal(got["re"].astype(np.int64), exp_re) and np.array_equal(got["im"].astype(np.int64), exp_im), name
