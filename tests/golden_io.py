"""Readers for tests/golden/*.  Fixtures are data only (formats, raw integers, seeds)."""
import glob
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _records(path):
    with gzip.open(path, "rt") as f:
        txt = f.read()
    dec = json.JSONDecoder()
    i, n = 0, len(txt)
    while i < n:
        while i < n and txt[i].isspace():
            i += 1
        if i >= n:
            break
        obj, i = dec.raw_decode(txt, i)
        yield obj


def gemm_cases(kind="real"):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, f"ref_gemm_{kind}_*.jsonl.gz"))):
        out.extend(_records(p))
    return out


def scalar_tables(part=None):
    pat = f"ref_scalar_{part}.jsonl.gz" if part is not None else "ref_scalar_*.jsonl.gz"
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, pat))):
        out.extend(_records(p))
    return out


def rounding_kat():
    with open(os.path.join(GOLD, "ref_rounding_kat.json")) as f:
        return json.load(f)["cases"]


def case_elems(j):
    """(A elem, B elem, C elem) of a golden GEMM record as qublas_amd.desc types."""
    from qublas_amd.desc import Qcomplex, Qu
    mk = lambda p: Qcomplex(Qu.from_tuple(p[0]), Qu.from_tuple(p[1])) if j["is_complex"] else Qu.from_tuple(p[0])
    return mk(j["a"]), mk(j["b"]), mk(j["c"])


def case_inputs(j, qoracle):
    """Host-layout A and B arrays of a golden GEMM record (explicit values or generator seeds)."""
    ea, eb, _ = case_elems(j)
    M, N, K = j["M"], j["N"], j["K"]
    inp = j["inputs"]
    if "seedA" in inp:
        return qoracle.fill(ea, M * K, inp["seedA"], inp["dist"]), qoracle.fill(eb, K * N, inp["seedB"], inp["dist"])

    def explicit(vals, e, n):
        arr = np.zeros(n, dtype=qoracle.host_dtype(e))
        v = np.asarray(vals, dtype=np.int64)
        if j["is_complex"]:
            arr["re"], arr["im"] = v[0::2], v[1::2]
        else:
            arr[:] = v
        return arr
    return explicit(inp["A"], ea, M * K), explicit(inp["B"], eb, K * N)


def case_expected(j, qoracle):
    _, _, ec = case_elems(j)
    n = j["M"] * j["N"]
    arr = np.zeros(n, dtype=qoracle.host_dtype(ec))
    if not j["is_complex"] and arr.dtype == qoracle.WIDE:   # values beyond 64 bits: JSON big integers -> two words
        return qoracle.to_host(j["C"], ec)
    v = np.asarray(j["C"], dtype=np.int64)
    if j["is_complex"]:
        arr["re"], arr["im"] = v[0::2], v[1::2]
    else:
        arr[:] = v
    return arr


def wide_tables(part=None):
    pat = f"ref_wide_{part}.jsonl.gz" if part is not None else "ref_wide_*.jsonl.gz"
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, pat))):
        out.extend(_records(p))
    return out


def eltwise_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "ref_eltwise_*.jsonl.gz"))):
        out.extend(_records(p))
    return out


def eltwise_epilogue(j):
    """(qgemul_epilogue, C format, [operand arrays]) of a golden element-wise record."""
    from qublas_amd.desc import Qu, qgemul_epilogue
    ep = qgemul_epilogue()
    ep.n_stages = len(j["stages"])
    ep.d = Qu.from_tuple(j["d"]).c()
    E = []
    for k, s in enumerate(j["stages"]):
        st = ep.stage[k]
        st.op, st.x_first, st.e_scalar = s["op"], s["x_first"], s["scalar"]
        st.e = Qu.from_tuple(s["e"]).c()
        st.r = Qu.from_tuple(s["r"]).c()
        st.t = Qu.from_tuple(s["t"]).c()
        E.append(np.asarray(s["E"], dtype=np.int64))
    return ep, Qu.from_tuple(j["c"]), E


def cplx_eltwise_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "ref_cplx_eltwise_*.jsonl.gz"))):
        out.extend(_records(p))
    return out


def cplx_eltwise_epilogue(j):
    """(qgemul_epilogue_cplx, C type, operand values of the real-part chain, of the imaginary-part chain) of a golden
    record of oracle/ref_cases_cplx_eltwise.cpp.  The record holds what the reference did: the operator, the operand and
    the two result formats; which stage each PART runs follows include/qgemul.h's table (complex operand or Qmul: both
    parts the operator; real operand of Qadd / Qsub: the imaginary part is carried over, or — real - complex — taken
    from the zero of the operand's type)."""
    from qublas_amd.desc import EW_ADD, EW_MUL, EW_PASS, EW_SUB, Qcomplex, Qu, qgemul_epilogue_cplx
    epc = qgemul_epilogue_cplx()
    E = ([], [])
    for p in range(2):
        epc.part[p].n_stages = len(j["stages"])
        epc.part[p].d = Qu.from_tuple(j["d"][p]).c()
    for k, s in enumerate(j["stages"]):
        epc.e_complex[k] = s["e_complex"]
        for p in range(2):
            st = epc.part[p].stage[k]
            st.op, st.x_first, st.e_scalar = s["op"], s["x_first"], s["scalar"]
            st.e = Qu.from_tuple(s["e"][p]).c()
            st.r = Qu.from_tuple(s["r"][p]).c()
            st.t = Qu.from_tuple(s["t"][p]).c()
            vals = np.asarray(s["Ere"] if (p == 0 or not s["e_complex"]) else s["Eim"], dtype=np.int64)
            if p == 1 and not s["e_complex"] and s["op"] != EW_MUL:
                if s["op"] == EW_ADD or s["x_first"]:
                    st.op, st.e_scalar = EW_PASS, 1
                else:
                    st.e_scalar, vals = 1, np.zeros(1, dtype=np.int64)
            E[p].append(vals)
    c = Qcomplex(Qu.from_tuple(j["c"][0]), Qu.from_tuple(j["c"][1]))
    return epc, c, E[0], E[1]


def bitstream_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "ref_bitstream_*.jsonl.gz"))):
        out.extend(_records(p))
    return out


def cplx_bitstream_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "ref_cplx_bitstream_*.jsonl.gz"))):
        out.extend(_records(p))
    return out

