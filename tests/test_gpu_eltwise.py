"""GPU parity of the fused element-wise epilogue, through the C-ABI (qgemul_run_ep and the
resident-data entry points):
  * every golden vector of the reference's lazy tensor operators (tests/golden/ref_eltwise_*), fed through a K = 1
    GEMM whose result IS the fixture's X tensor — on the MFMA kernels the chain runs fused in their epilogue, on the
    tree kernels as the linear pass after them;
  * real GEMMs + chains against oracle GEMM + oracle chain, as the pass after the kernel (default) and fused into
    the MFMA epilogue (QG_OPT_FUSED_EPILOGUE), which must agree bit for bit."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import Ew, Qu, RND, SAT, TRN, WRP, Tags, lower, lower_epilogue

pytestmark = pytest.mark.gpu

CASES = G.eltwise_cases()
ONE = Qu(1, 0, False)


def host_array(vals, q: Qu):
    return np.asarray(vals, dtype=np.int64).astype(np.int32 if q.storage_bits <= 32 else np.int64)


def run_ep(d, ep, A, B, E, dq: Qu, flags=0, ldc=0):
    out = np.zeros((ldc or d.M) * d.N, dtype=np.int32 if dq.storage_bits <= 32 else np.int64)
    return capi.run_ep(d, ep, out, A, B, E, flags=flags, ldc=ldc)


@pytest.mark.parametrize("j", CASES, ids=lambda j: j["name"])
def test_golden_vectors_through_identity_gemm(j):
    """C = X * 1 with K = 1 reproduces the fixture's tensor exactly, so D must equal the reference's D."""
    ep, c, E = G.eltwise_epilogue(j)
    n = j["n"]
    d = lower(c, ONE, c, n, 1, 1, mul_args=c)
    info = capi.classify_ep_status(d, ep)[1]
    A = host_array(j["X"], c)
    B = host_array([1], ONE)
    dq = Qu.from_tuple(j["d"])
    Eh = [host_array(e, Qu.from_tuple(s["e"])) for e, s in zip(E, j["stages"])]
    got = run_ep(d, ep, A, B, Eh, dq)
    assert np.array_equal(got.astype(np.int64), np.asarray(j["D"], dtype=np.int64)), (j["name"], capi.KERNEL_NAMES[info.kernel])
    # fused into the MFMA kernel's epilogue where a variant exists, and on the exact tree kernel + the pass
    assert np.array_equal(run_ep(d, ep, A, B, Eh, dq, flags=capi.OPT_FUSED_EPILOGUE), got)
    assert np.array_equal(run_ep(d, ep, A, B, Eh, dq, flags=capi.OPT_FORCE_TREE), got)


def test_golden_cases_cover_fused_and_unfused_kernels():
    kinds = set()
    for j in CASES:
        ep, c, _ = G.eltwise_epilogue(j)
        kinds.add(capi.KERNEL_NAMES[capi.classify_ep_status(lower(c, ONE, c, j["n"], 1, 1, mul_args=c), ep)[1].kernel])
    assert {"mfma_i8", "mfma_i8_limb"} <= kinds and (kinds & {"tree_i64", "tree_i32", "gemv_i32", "gemv_i64"})   # (MFMA and exact-tree kernels)


E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E43 = Qu(4, 3)
C238, B106, S34 = Qu(23, 8), Qu(10, 6), Qu(3, 4)
CHAINS = {
    "bias": ([Ew("add", B106)], Qu(12, 4, True, RND.CONV, SAT.SMGN)),
    "scale_bias": ([Ew("mul", S34, Tags(24, 8), scalar=True, into=Qu(24, 8)), Ew("add", B106)], Qu(16, 6, True, RND.ZERO, SAT.TCPL)),
    "sub_from_scalar_wrap": ([Ew("sub", S34, x_first=False, scalar=True)], Qu(6, 2, True, RND.INF, WRP.TCPL)),
    "mul_tensor_fullprec_then_narrow": ([Ew("mul", S34, Tags(FullPrec=True), into=Qu(20, 6, True, RND.CONV, SAT.ZERO)),
                                         Ew("add", B106, Tags(isSigned=False)), Ew("sub", B106)], Qu(40, 8)),
    "convert_only": ([], Qu(9, 3, True, RND.NEG_INF, SAT.SMGN)),
}


def _operands(oracle, stages, n, seed0=70):
    Eo, Eh = [], []
    for k, st in enumerate(stages):
        m = 1 if st.scalar else n
        h = oracle.fill(st.e, m, seed0 + k, 0)
        Eh.append(h)
        Eo.append(h.astype(np.int64))
    return Eo, Eh


C158 = Qu(15, 8)
FUSED_SEEN = set()


def _plan_fuses(d, ep, flags=0):
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d, flags=flags, epilogue=ep)
        f = plan.fuses_epilogue()
        plan.close()
    return f


@pytest.mark.parametrize("chain", sorted(CHAINS))
@pytest.mark.parametrize("cw", ["narrowC", "wideC"])
@pytest.mark.parametrize("cfg", ["limb", "i8", "i8_big", "tree"])
def test_gemm_plus_chain_vs_oracle(oracle, cfg, cw, chain):
    """narrowC: a 24-bit C keeps most chains within 32-bit arithmetic, which the MFMA kernels fuse; wideC: a 32-bit C
    forces 64-bit arithmetic, which runs as the stand-alone pass.  Either way the result is the oracle's."""
    stages, dq = CHAINS[chain]
    cq = C158 if cw == "narrowC" else C238
    if cfg == "limb":      # 3x3 int8 limbs, the bench workload's operand formats
        ea, ec, M, N, K, kw, kern = E88, cq, 200, 136, 192, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), "mfma_i8_limb"
    elif cfg == "i8":      # single limb, 128x128 tiles (32x32x32 MFMA)
        ea, ec, M, N, K, kw, kern = E43, cq, 130, 260, 128, dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), "mfma_i8"
    elif cfg == "i8_big":  # single limb, 256x256 tiles (16x16x64 MFMA)
        ea, ec, M, N, K, kw, kern = E43, cq, 4096, 4096, 64, dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), "mfma_i8"
    else:                  # default tags: exact tree kernel + stand-alone pass
        if cw == "wideC":
            pytest.skip("one C format is enough for the tree kernel")
        ea, ec, M, N, K, kw, kern = E88, E88, 96, 80, 128, dict(), "tree_i32"
    d = lower(ea, ea, ec, M, N, K, **kw)
    ep = lower_epilogue(ec, stages, dq)
    st, info = capi.classify_ep_status(d, ep)
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == kern, info.reason
    fused = _plan_fuses(d, ep, capi.OPT_FUSED_EPILOGUE)  # inside the MFMA kernel's epilogue wherever a variant exists
    assert _plan_fuses(d, ep) == (fused and cfg == "limb")   # default: the limb kernel only
    assert not _plan_fuses(d, ep, capi.OPT_UNFUSED_EPILOGUE)
    assert not (fused and (cfg == "tree" or cw == "wideC" and stages))
    if fused:
        FUSED_SEEN.add((cfg, chain))
    A = oracle.fill(ea, M * K, 1, 1 if cfg != "tree" else 0)
    B = oracle.fill(ea, K * N, 2, 1 if cfg != "tree" else 0)
    Eo, Eh = _operands(oracle, stages, M * N)
    got = run_ep(d, ep, A, B, Eh, dq)
    if cfg == "i8_big":
        # the oracle GEMM on a band of rows only; the rest through fused == unfused below
        rows, cols = (1000, 1100), (0, N)
        Cx = oracle.gemm(d, A, B, ec, rows=rows, cols=cols, nthreads=8).astype(np.int64).reshape(N, M)[:, rows[0]:rows[1]]
        Es = [e if e.size == 1 else e.reshape(N, M)[:, rows[0]:rows[1]].reshape(-1) for e in Eo]
        exp = oracle.eltwise(ep, ec, Cx.reshape(-1), Es)
        assert np.array_equal(got.reshape(N, M)[:, rows[0]:rows[1]].reshape(-1).astype(np.int64), exp)
    else:
        Cx = oracle.gemm(d, A, B, ec, nthreads=8).astype(np.int64)
        exp = oracle.eltwise(ep, ec, Cx, Eo)
        assert np.array_equal(got.astype(np.int64), exp)
    if cfg != "tree":
        # the same MFMA kernel with the chain inside its epilogue (when eligible), and the exact tree kernel + the pass
        assert np.array_equal(run_ep(d, ep, A, B, Eh, dq, flags=capi.OPT_FUSED_EPILOGUE), got)
        assert np.array_equal(run_ep(d, ep, A, B, Eh, dq, flags=capi.OPT_UNFUSED_EPILOGUE), got)
        if cfg != "i8_big":
            assert np.array_equal(run_ep(d, ep, A, B, Eh, dq, flags=capi.OPT_FORCE_TREE), got)
    assert len(np.unique(got)) > 8   # the comparison is not hidden by saturation


def test_fused_path_was_exercised_on_every_mfma_kernel():
    """(runs after the parametrised test above) the three kernels with a fused epilogue all took it for some chain"""
    assert {c for c, _ in FUSED_SEEN} == {"limb", "i8", "i8_big"}, sorted(FUSED_SEEN)
    assert {("limb", "bias"), ("i8", "bias"), ("i8_big", "bias")} <= FUSED_SEEN, sorted(FUSED_SEEN)


def test_run_ep_keeps_column_padding(oracle):
    stages, dq = CHAINS["bias"]
    M, N, K, ldc = 70, 9, 64, 75
    d = lower(E43, E43, C238, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
    ep = lower_epilogue(C238, stages, dq)
    A, B = oracle.fill(E43, M * K, 1, 0), oracle.fill(E43, K * N, 2, 0)
    Eo, Eh = _operands(oracle, stages, M * N)
    out = np.full(ldc * N, -777, dtype=np.int32)
    capi.run_ep(d, ep, out, A, B, Eh, ldc=ldc)
    exp = oracle.eltwise(ep, C238, oracle.gemm(d, A, B, C238).astype(np.int64), Eo).reshape(N, M)
    o2 = out.reshape(N, ldc)
    assert np.array_equal(o2[:, :M].astype(np.int64), exp) and (o2[:, M:] == -777).all()


def test_resident_api_and_error_paths(oracle):
    """plan_create_ep / pack_e / execute_ep on resident buffers; a plan with an epilogue refuses qgemul_execute."""
    stages, dq = CHAINS["scale_bias"]
    M = N = K = 256
    d = lower(E88, E88, C238, M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
    ep = lower_epilogue(C238, stages, dq)
    A, B = oracle.fill(E88, M * K, 1, 1), oracle.fill(E88, K * N, 2, 1)
    Eo, Eh = _operands(oracle, stages, M * N)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d, epilogue=ep)
        assert plan.packed_e_bytes(0) == 0 and plan.packed_e_bytes(1) == plan.info.packed_bytes[2]
        dA, dB = ctx.alloc(A.nbytes), ctx.alloc(B.nbytes)
        ctx.h2d(dA, A); ctx.h2d(dB, B)
        pA, pB, pD = (ctx.alloc(int(plan.info.packed_bytes[i])) for i in range(3))
        plan.pack(capi.OPERAND_A, dA, pA); plan.pack(capi.OPERAND_B, dB, pB)
        dE = ctx.alloc(Eh[1].nbytes); ctx.h2d(dE, Eh[1])
        pE = ctx.alloc(plan.packed_e_bytes(1)); plan.pack_e(1, dE, pE)
        args = plan.ep_args(packed=[0, pE], scalars=[int(Eo[0][0]), 0])
        with pytest.raises(capi.QgemulError):
            plan.execute(pD, pA, pB)
        with pytest.raises(capi.QgemulError):
            plan.execute_ep(pD, pA, pB, plan.ep_args(packed=[0, 0], scalars=[1, 0]))   # tensor operand missing
        plan.execute_ep(pD, pA, pB, args)
        out = np.zeros(M * N, dtype=np.int32)
        dD = ctx.alloc(out.nbytes)
        plan.unpack_c(pD, dD)
        ctx.d2h(out, dD)
        ms = plan.time_execute_ep(pD, pA, pB, args, 1, 3)
        assert ms > 0
        plan.close()
    exp = oracle.eltwise(ep, C238, oracle.gemm(d, A, B, C238, nthreads=8).astype(np.int64), Eo)
    assert np.array_equal(out.astype(np.int64), exp)
