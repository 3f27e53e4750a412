"""The C++ drop-in boundary end to end on the GPU (run with -m gpu):
  * oracle/_ref/ref_binding_demo — the REFERENCE header's own types + include/qgemul_reference_binding.hpp,
    built in the build container (the header cannot travel), executed here;
  * tests/binding/amd_header_run.cpp — the standalone include/QuBLAS_amd.h, compiled here with clang++ -std=c++23.
Both call Qgemul<...>(C, A, B) exactly as the README does and must print the matrices the reference's own
primitives produced (tests/golden)."""
import json
import os
import subprocess

import pytest

import golden_io as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _golden():
    return {j["name"]: j for j in G.gemm_cases("real")}


def _check(lines):
    gold = _golden()
    seen = 0
    for l in lines:
        r = json.loads(l)
        assert "error" not in r, r
        if r["name"] in gold:
            assert r["C"] == gold[r["name"]]["C"], r["name"]
            seen += 1
    return seen


def test_reference_header_binding_runs_on_gpu():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_binding_demo")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_binding_demo was not built (needs the reference header, build container only)")
    out = subprocess.check_output([exe], text=True)
    lines = out.strip().splitlines()
    assert _check([l for l in lines if '"epilogue' not in l]) == 4
    # the Then* front-ends on the reference's own types: D of the golden element-wise record, bit for bit
    ep = [json.loads(l) for l in lines if '"epilogue"' in l]
    gold = {j["name"]: j for j in G.eltwise_cases()}
    assert len(ep) == 1 and ep[0]["D"] == gold[ep[0]["epilogue"]]["D"]
    # ... and after a complex Qgemul (complex tensors, realT<> tag, a real operand whose imaginary part is carried over)
    epc = [json.loads(l) for l in lines if '"epilogue_cplx"' in l]
    cgold = {j["name"]: j for j in G.cplx_eltwise_cases()}
    assert len(epc) == 1
    assert epc[0]["Dre"] == cgold[epc[0]["epilogue_cplx"]]["Dre"] and epc[0]["Dim"] == cgold[epc[0]["epilogue_cplx"]]["Dim"]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs AMD clang (C++23)")
def test_standalone_header_runs_on_gpu(tmp_path):
    exe = tmp_path / "amd_header_run"
    lib = os.path.join(ROOT, "qublas_amd")
    subprocess.check_call([CLANG, "-std=c++23", "-O1", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "binding", "amd_header_run.cpp"), "-o", str(exe), "-L" + lib, "-lqugemm",
                           "-Wl,-rpath," + lib])
    out = subprocess.check_output([str(exe)], text=True)
    lines = out.strip().splitlines()
    assert _check(lines) == 4
    q = [json.loads(l) for l in lines if '"qreduce"' in l]
    assert q and q[0]["C"] == [80]
    # Qreduce on signed SAT::SMGN elements with the raw minimum present: the header's lowering against the Python one (which
    # the reference's own tables pin, tests/test_reduce.py)
    import numpy as np
    from oracle import qoracle as oracle
    from qublas_amd.desc import Qu, SAT, TRN, lower_reduce, reduce_result_type
    sm = Qu(3, 4, True, TRN.TCPL, SAT.SMGN)
    w = np.array([(i * 37) % 256 - 128 for i in range(16)], dtype=np.int32)
    w[0] = w[8] = -128
    exp = []
    for levels in ([Qu(4, 6, True, TRN.TCPL, SAT.ZERO)], None):
        ec = reduce_result_type(sm, levels or [], 16)
        exp.append(int(oracle.gemm(lower_reduce(sm, 1, 16, levels), w, np.ones(16, dtype=np.int32), ec)[0]))
    qs = [json.loads(l) for l in lines if '"qreduce_smgn"' in l]
    assert qs and qs[0]["C"] == exp
    # the variadic overload Qreduce<L...>(q1, q2, ...) on the reference tables' inputs of seed 1 (tests/golden/ref_scalar_10)
    var = {t["name"]: t for t in G.scalar_tables(10) if t["x"][:2] == [17, 120]}
    names = ["var4_readme", "var3_default", "var5_nar_wide", "var6_nar_wide", "var7_nar_wide", "var7_nar", "var5_list"]
    qv = [json.loads(l) for l in lines if '"qreduce_variadic"' in l]
    assert qv and qv[0]["C"] == [var[n]["y"] for n in names] and qv[0]["F"] == [var[n]["fr"][1] for n in names]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs AMD clang (C++23)")
def test_standalone_header_all_devices_flag(tmp_path):
    """The same README-style program with `QgemulRunFlags() |= QG_OPT_ALL_DEVICES`: every Qgemul<...> goes through
    qgemul_run_sharded over all visible devices (one here) and must print the same matrices."""
    exe = tmp_path / "amd_header_run_all"
    lib = os.path.join(ROOT, "qublas_amd")
    subprocess.check_call([CLANG, "-std=c++23", "-O1", "-w", "-DQUBLAS_TEST_ALL_DEVICES", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "binding", "amd_header_run.cpp"), "-o", str(exe), "-L" + lib, "-lqugemm",
                           "-Wl,-rpath," + lib])
    out = subprocess.check_output([str(exe)], text=True)
    assert _check(out.strip().splitlines()) == 4


def _check_complex_chain(r, oracle):
    """the complex chain of tests/binding/amd_header_ep_run.cpp against oracle GEMM + the Python lowering's part-wise chains"""
    import numpy as np
    from qublas_amd.desc import EwC, Qcomplex, Qu, RND, SAT, Tags, TFComplexMul, lower, lower_epilogue_cplx
    c5 = Qcomplex(Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL))
    cw = Qcomplex(Qu(18, 6), Qu(18, 6))
    cbias = Qcomplex(Qu(5, 4), Qu(3, 2))
    cdst = Qcomplex(Qu(10, 4, True, RND.CONV, SAT.SMGN), Qu(20, 12))
    M, N, K = r["M"], r["N"], r["K"]
    assert r["elem_bytes"] == 16
    d = lower(c5, c5, cw, M, N, K, mul_args=TFComplexMul())
    epc = lower_epilogue_cplx(cw, [EwC("mul", Qu(2, 2), real_tags=Tags(18, 6), imag_tags=Tags(18, 6), scalar=True, into=cw),
                                   EwC("add", cbias, imag_tags=Tags(intBits=19)), EwC("sub", Qu(5, 4), x_first=False)], cdst)
    A = np.zeros(M * K, dtype=oracle.host_dtype(c5)); B = np.zeros(K * N, dtype=oracle.host_dtype(c5))
    i = np.arange(M * K); A["re"], A["im"] = (i * 37) % 1024 - 512, (i * 11 + 3) % 16 - 8
    i = np.arange(K * N); B["re"], B["im"] = (i * 53 + 1) % 1024 - 512, (i * 7) % 16 - 8
    i = np.arange(M * N)
    bias_re, bias_im, off = (i * 29) % 1024 - 512, (i * 13) % 64 - 32, (i * 41) % 1024 - 512
    C = oracle.gemm(d, A, B, cw)
    zero = np.zeros(1, dtype=np.int64)
    exp_re, exp_im = oracle.eltwise_cplx(epc, cw, C["re"].astype(np.int64), C["im"].astype(np.int64),
                                         [np.array([-7]), bias_re, off], [np.array([-7]), bias_im, zero])
    assert r["Dre"] == exp_re.tolist() and r["Dim"] == exp_im.tolist()
    assert len(set(r["Dre"])) > 8 and len(set(r["Dim"])) > 8


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs AMD clang (C++23)")
def test_standalone_header_elementwise_chain_on_gpu(tmp_path, oracle):
    """Qgemul<…, QgemulResult<CT>>(D, A, B, ThenMul<…>(s), ThenAdd<>(Bias), ThenRsub<…>(off)) through QuBLAS_amd.h:
    the C++ lowering must resolve the formats the Python mirror resolves, and D must be the oracle's."""
    import numpy as np
    from qublas_amd.desc import Ew, Qu, RND, SAT, TRN, Tags, lower, lower_epilogue
    exe = tmp_path / "amd_header_ep_run"
    lib = os.path.join(ROOT, "qublas_amd")
    subprocess.check_call([CLANG, "-std=c++23", "-O1", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "binding", "amd_header_ep_run.cpp"), "-o", str(exe), "-L" + lib, "-lqugemm",
                           "-Wl,-rpath," + lib])
    lines = [json.loads(l) for l in subprocess.check_output([str(exe)], text=True).strip().splitlines()]
    assert all("error" not in l for l in lines), lines
    r = next(l for l in lines if l["name"] == "scale_bias_rsub")
    _check_complex_chain(next(l for l in lines if l["name"] == "cplx_scale_cbias_rsub"), oracle)
    e88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    ct, t1, bt, st = Qu(15, 8), Qu(16, 8), Qu(10, 6), Qu(3, 3)
    dt = Qu(12, 4, True, RND.CONV, SAT.SMGN)
    M, N, K = r["M"], r["N"], r["K"]
    d = lower(e88, e88, ct, M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
    ep = lower_epilogue(ct, [Ew("mul", st, Tags(16, 8), scalar=True, into=t1), Ew("add", bt),
                             Ew("sub", st, Tags(QuMode=RND.CONV), x_first=False, scalar=True)], dt)
    tup = lambda f: [f.I, f.F, f.S, f.Q, f.O]
    assert r["n_stages"] == 3 and r["d"] == tup(ep.d)
    for k in range(3):
        assert r[f"r{k}"] == tup(ep.stage[k].r) and r[f"op{k}"] == [ep.stage[k].op, ep.stage[k].x_first, ep.stage[k].e_scalar]
        if k < 2:
            assert r[f"t{k}"] == tup(ep.stage[k].t)
    i = np.arange(M * K, dtype=np.uint64)
    A = ((i * np.uint64(2654435761)) % np.uint64(8192)).astype(np.int64) - 4096
    i = np.arange(K * N, dtype=np.uint64)
    B = ((i * np.uint64(40503) + np.uint64(7)) % np.uint64(8192)).astype(np.int64) - 4096
    i = np.arange(M * N, dtype=np.uint64)
    bias = ((i * np.uint64(97)) % np.uint64(131072)).astype(np.int64) - 65536
    C = oracle.gemm(d, A.astype(np.int32), B.astype(np.int32), ct).astype(np.int64)
    exp = oracle.eltwise(ep, ct, C, [np.array([13]), bias, np.array([-20])])
    assert r["D"] == exp.tolist()
    assert len(set(r["D"])) > 20
