"""Descriptor lowering: the Python mirror (qublas_amd/desc.py) and the two C++ headers
(include/qgemul_reference_binding.hpp on the real reference header, include/QuBLAS_amd.h standalone)
must produce exactly the formats the reference's own types report for the same tags
(tests/golden/ref_gemm_*.jsonl.gz, printed by oracle/ref_driver.hpp).  CPU only; the C++ probes
need AMD clang (C++23) and, for the binding, the reference header (build container)."""
import json
import os
import shutil
import subprocess

import pytest

import golden_io as G
from qublas_amd.desc import (BasicComplexMul, Qcomplex, Qu, RND, SAT, TRN, WRP, Tags, TFComplexMul, desc_to_dict,
                             lower)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
KEYS = ("M", "N", "K", "transA", "is_complex", "cmul", "a", "b", "c", "mul", "n_levels", "level_add", "level")

e88z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
e43 = Qu(4, 3)
u44 = Qu(4, 4, False)
n63 = Qu(6, -3)
w16 = Qu(16, 3)
t1 = Qu(6, 5, True, RND.CONV, SAT.SMGN)
t2 = Qu(8, 4, True, RND.ZERO, SAT.TCPL)
t3 = Qu(9, 2, True, TRN.SMGN, SAT.ZERO)
pm = Qu(5, 4, True, RND.INF, SAT.TCPL)
type1 = Qu(6, 3, True, TRN.TCPL, SAT.ZERO)
type2 = Qu(6, -3)
r55 = Qu(5, 5)
c55 = Qcomplex(r55, r55)
rw = Qu(18, 6, True, RND.POS_INF, SAT.TCPL)
cw = Qcomplex(rw, rw)
r63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
i63n = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
c5 = Qcomplex(r63, i63n)
tA = Qu(9, 4, True, RND.CONV, SAT.SMGN)
tB = Qu(7, 2, True, TRN.SMGN, SAT.ZERO)
tC = Qu(10, 5, True, RND.ZERO, WRP.TCPL)
tD = Qu(8, 3, True, RND.INF, SAT.TCPL)
TFmix = TFComplexMul(abT=tA, cdT=tD, abcT=tC, cdbT=tB, badT=tA, ABT=tD, BCT=tC)
l1 = Qcomplex(Qu(12, 4, True, RND.CONV, SAT.SMGN), Qu(11, 6, True, TRN.SMGN, SAT.ZERO))
l2 = Qcomplex(Qu(16, 2, True, RND.ZERO), Qu(16, 3, True, RND.INF, WRP.TCPL))

# golden case name -> the tags it was generated with (oracle/ref_cases_*.cpp)
PY_CASES = {
    "c1_nn_classT": dict(A=e88z, B=e88z, C=e88z, mul_args=e88z, add_args=[e88z]),
    "c1_tn_classT": dict(A=e88z, B=e88z, C=e88z, mul_args=e88z, add_args=[e88z], transposed_a=True),
    "c1_nn_default": dict(A=e88z, B=e88z, C=e88z),
    "c1_nn_classL": dict(A=e88z, B=e88z, C=e88z, mul_args=Tags(17, 16), add_args=[Qu(29, 16)]),
    "e43_L_33x17x128_full_wideC": dict(A=e43, B=e43, C=w16, mul_args=Tags(9, 6), add_args=[Qu(19, 6)]),
    "e43_L_tn_4x4x4096_full": dict(A=e43, B=e43, C=e43, mul_args=Tags(9, 6), add_args=[Qu(21, 6)], transposed_a=True),
    "u44_default_8x8x64_full": dict(A=u44, B=u44, C=u44),
    "u44_L_8x8x64_full": dict(A=u44, B=u44, C=Qu(14, 8, False), mul_args=Tags(8, 8), add_args=[Qu(14, 8, False)]),
    "n63_fullprec_tn_8x8x64_full": dict(A=n63, B=n63, C=Qu(16, -3), mul_args=Tags(FullPrec=True), add_args=[Qu(20, -6)],
                                        transposed_a=True),
    "mixed_e88z_e43_default_8x8x64_small": dict(A=e88z, B=e43, C=w16),
    "mixed_e43_u44_default_8x8x64_full": dict(A=e43, B=u44, C=w16),
    "e43_levels3_8x8x64_full": dict(A=e43, B=e43, C=w16, mul_args=pm, add_args=[t1, t2, t3]),
    "e43_levels2_8x8x16_full": dict(A=e43, B=e43, C=e43, mul_args=Tags(fracBits=2, QuMode=RND.POS_INF), add_args=[t2, t1]),
    "e88z_rnd_levels1_8x8x32_small": dict(A=e88z, B=e88z, C=e88z, mul_args=Tags(intBits=10, QuMode=RND.NEG_INF, OfMode=SAT.SMGN),
                                          add_args=[Qu(12, 6, True, RND.POS_INF)]),
    "readme_list_tn_4x4x4": dict(A=type1, B=type1, C=type1, mul_args=type1, add_args=[type1, type2], transposed_a=True),
    "e43_K5": dict(A=e43, B=e43, C=w16, add_args=[t1, t2]),
    "e43_K1": dict(A=e43, B=e43, C=w16, add_args=[t1, t2]),
    "e43_K37_default": dict(A=e43, B=e43, C=w16),
    "c5_basic_default_8x8x64_full": dict(A=c5, B=c5, C=c5),
    "c5_tf_default_8x8x64_full": dict(A=c5, B=c5, C=c5, mul_args=TFComplexMul()),
    "c55_tf_mixedtags_8x8x16_full": dict(A=c55, B=c55, C=cw, mul_args=TFmix),
    "c55_c5_tf_mixedtags_tn_8x8x16_small": dict(A=c55, B=c5, C=cw, mul_args=TFmix, transposed_a=True),
    "c55_basic_mixedtags_8x8x16_full": dict(A=c55, B=c55, C=cw,
                                            mul_args=BasicComplexMul(acT=tA, bdT=tB, adT=tC, bcT=tD, acbdT=tC, adbcT=tA)),
    "c55_basic_loosetags_4x4x8_full": dict(A=c55, B=c55, C=cw,
                                           mul_args=BasicComplexMul(bdT=tB, loose=Tags(intBits=12, OfMode=SAT.ZERO))),
    "c55_tf_levels2_8x8x32_full": dict(A=c55, B=c55, C=cw, mul_args=TFComplexMul(), add_args=[l1, l2]),
    "c55_basic_levels1_K7_small": dict(A=c55, B=c55, C=cw, add_args=[l1]),
    "tf_quirk_baT_1x1x1": dict(A=Qcomplex(Qu(14, 6), Qu(14, 6)), B=Qcomplex(Qu(14, 6), Qu(14, 6)), C=Qcomplex(Qu(14, 6), Qu(14, 6)),
                               mul_args=TFComplexMul(*([Qu(14, 6)] * 8))),
}


def golden_by_name():
    return {j["name"]: j for j in G.gemm_cases("real") + G.gemm_cases("cplx")}


@pytest.mark.parametrize("name", sorted(PY_CASES))
def test_python_lowering_matches_reference_types(name):
    j = golden_by_name()[name]
    kw = dict(PY_CASES[name])
    d = lower(kw.pop("A"), kw.pop("B"), kw.pop("C"), j["M"], j["N"], j["K"], **kw)
    got = desc_to_dict(d)
    for k in KEYS:
        assert got[k] == j[k], (name, k)


def _probe(src, extra_inc, tmp_path):
    exe = tmp_path / "probe"
    cmd = [CLANG, "-std=c++23", "-O0", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "binding")]
    cmd += extra_inc + [os.path.join(ROOT, "tests", "binding", src), "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.check_output([str(exe)], text=True)
    return [json.loads(l) for l in out.splitlines() if l.strip()]


@pytest.mark.skipif(not (os.path.exists(CLANG) and os.path.isdir("/root/reference/include")),
                    reason="needs AMD clang and the reference header (build container)")
def test_reference_binding_lowering(tmp_path):
    """include/qgemul_reference_binding.hpp compiled against the REAL reference header."""
    gold = golden_by_name()
    recs = _probe("ref_binding_probe.cpp", ["-I/root/reference/include"], tmp_path)
    eps = [r for r in recs if "epilogue" in r]
    epcs = [r for r in recs if "epilogue_cplx" in r]
    recs = [r for r in recs if "epilogue" not in r and "epilogue_cplx" not in r]
    assert len(recs) >= 12
    for r in recs:
        j = gold[r["name"]]
        for k in KEYS:
            assert r[k] == j[k], (r["name"], k)
    _check_epilogues(eps)
    _check_cplx_epilogues(epcs)


def _check_epilogues(eps):
    """the Then* front-ends must resolve the operators of the golden element-wise cases to the formats the reference's
    own tensor operators reported (tests/golden/ref_eltwise_*)"""
    import golden_io as G
    gold = {j["name"]: j for j in G.eltwise_cases()}
    assert len(eps) == 3
    for r in eps:
        j = gold[r["epilogue"]]
        assert r["d"] == j["d"] and len(r["stages"]) == len(j["stages"])
        for k, (a, b) in enumerate(zip(r["stages"], j["stages"])):
            for key in ("op", "x_first", "scalar", "e", "r"):
                assert a[key] == b[key], (r["epilogue"], k, key)
            if k + 1 < len(j["stages"]):
                assert a["t"] == b["t"], (r["epilogue"], k)


def _check_cplx_epilogues(eps):
    """... and after a complex Qgemul: the two part-wise chains the headers lower to must be the ones the golden records of
    the reference's complex tensor operators imply (tests/golden_io.py, cplx_eltwise_epilogue)"""
    import golden_io as G
    gold = {j["name"]: j for j in G.cplx_eltwise_cases()}
    assert len(eps) == 6
    tup = lambda f: [f.I, f.F, f.S, f.Q, f.O]
    for r in eps:
        exp, _, _, _ = G.cplx_eltwise_epilogue(gold[r["epilogue_cplx"]])
        n = exp.part[0].n_stages
        assert r["n"] == [n, n] and r["d"] == [tup(exp.part[0].d), tup(exp.part[1].d)]
        for k, st in enumerate(r["stages"]):
            assert st["e_complex"] == exp.e_complex[k]
            for p in range(2):
                a, b = st["parts"][p], exp.part[p].stage[k]
                assert (a["op"], a["scalar"]) == (b.op, b.e_scalar), (r["epilogue_cplx"], k, p)
                if b.op != 4:
                    assert a["x_first"] == b.x_first and a["e"] == tup(b.e), (r["epilogue_cplx"], k, p)
                assert a["r"] == tup(b.r), (r["epilogue_cplx"], k, p)
                if k + 1 < n:
                    assert a["t"] == tup(b.t), (r["epilogue_cplx"], k, p)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs AMD clang (C++23)")
def test_standalone_header_lowering(tmp_path):
    """include/QuBLAS_amd.h (own tag API, no reference header) lowers the same tags to the same formats."""
    gold = golden_by_name()
    recs = _probe("amd_header_probe.cpp", [], tmp_path)
    eps = [r for r in recs if "epilogue" in r]
    epcs = [r for r in recs if "epilogue_cplx" in r]
    recs = [r for r in recs if "epilogue" not in r and "epilogue_cplx" not in r]
    assert len(recs) >= 12
    for r in recs:
        j = gold[r["name"]]
        for k in KEYS:
            assert r[k] == j[k], (r["name"], k)
    _check_epilogues(eps)
    _check_cplx_epilogues(epcs)
