"""Values beyond 64 bits (the reference's multi-word ArbiInt<N > 64>, QuBLAS.h:566-912): products, sums, level formats and C of
up to 120 bits.  Round 2 refused every descriptor with an intermediate beyond 62 bits.

tests/golden/ref_wide_*.jsonl.gz were produced by the REAL reference header (oracle/ref_cases_wide.cpp): converting constructor
across the 64-bit boundary (all 7 QuModes x 4 OfModes), Qmul / Qadd / Qsub with wide results, Qreduce with wide levels, Qgemul
compositions on Q15.16 words with exact (linear class) and quantising (tree class) wide types.  They pin
  * the 128-bit arithmetic of oracle/qoracle.c — including the one place where the reference's multi-word code is NOT the
    arithmetic definition yet is what every user gets: comparing a multi-word value with one-word bounds (operator<=>,
    QuBLAS.h:1781-1793) reads the low word as a signed number, so a value in [2^63, 2^64) or [-2^64, -2^63) is neither above
    the maximum nor below the minimum of a one-word target and comes out as its low word; the tables hold values in those bands;
  * the list of combinations the engine REFUSES because the reference computes a width artefact there (`artefact` below) or
    cannot even compile them (the "not_compiled" records: signed WRP::TCPL between multi-word types of different widths).
CPU tests: oracle against the tables, planner verdicts.  GPU tests (-m gpu): the 128-bit kernels against the tables and the oracle."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import (BasicComplexMul, Qcomplex, Qu, RND, SAT, TRN, WRP, Tags, TFComplexMul, desc_from_dict, lower, lower_reduce,
                             reduce_result_type)

TABLES = G.wide_tables()


def sb(f: Qu) -> int:
    return f.storage_bits


def round_width(n, d, q):
    if d <= 0:
        return n - d
    w = max(n - d, 1)
    return w if q in (TRN.TCPL, TRN.SMGN) else w + 1


def artefact(kind, fa: Qu, fb, fr: Qu) -> str:
    """Why the reference's result for this operation is a width artefact ('' = it is the arithmetic definition, up to the
    multi-word comparison the oracle restates).  The planner refuses exactly these (qg_plan.cpp: fmt_ok, through, do_addsub)."""
    if kind == "cvt":
        n, d = sb(fa), fa.fracBits - fr.fracBits
    elif kind == "mul":
        n, d = sb(fa) + sb(fb), fa.fracBits + fb.fracBits - fr.fracBits
    else:
        fm = max(fa.fracBits, fb.fracBits)
        na, nb = sb(fa) + fm - fa.fracBits, sb(fb) + fm - fb.fracBits
        n, d = max(na, nb) + 1, fm - fr.fracBits
        if kind == "sub" and na > 64 and nb <= 64:
            return "operator-(multi-word, one-word)"
    if sb(fr) == 65:
        return "ArbiInt<65>::maximum() is -1"
    if d in (32, 64) and fr.QuMode <= RND.CONV:
        return "RND over a 32 / 64-bit shift (allOnes artefact)"
    if d > 0 and fr.QuMode == RND.CONV and n > 64:
        return "RND::CONV of a multi-word value"
    return ""


def test_the_reference_cannot_compile_some_combinations():
    """... and only signed WRP::TCPL between multi-word types of different widths (oracle/ref_cases_wide_probe.log)"""
    nc = [t for t in TABLES if t.get("kind") == "not_compiled"]
    assert len(nc) == 18
    for t in nc:
        assert "WRP::TCPL" in t["what"] or "lv3" in t["what"], t["what"]


def test_oracle_matches_wide_conversions(oracle):
    n = n_art = 0
    for t in TABLES:
        if t.get("kind") != "cvt":
            continue
        f, to = Qu.from_tuple(t["from"]), Qu.from_tuple(t["to"])
        why = artefact("cvt", f, None, to)
        got = [oracle.convert_w(x, f, to) for x in t["x"]]
        if why:
            n_art += got != t["y"]          # (documented: these are the ones that differ)
            continue
        assert got == t["y"], (t["from"], t["to"], [(x, y, g) for x, y, g in zip(t["x"], t["y"], got) if y != g][:3])
        n += len(got)
    assert n > 14000 and n_art > 60


def test_oracle_matches_wide_mul_add_sub(oracle):
    n = n_art = 0
    for t in TABLES:
        k = t.get("kind")
        if k not in ("mul", "add", "sub"):
            continue
        fa, fb, fr = (Qu.from_tuple(t[m]) for m in ("fa", "fb", "fr"))
        got = [oracle.mul_w(a, fa, b, fb, fr) if k == "mul" else oracle.add_w(a, fa, b, fb, fr, k == "sub") for a, b in t["xy"]]
        if artefact(k, fa, fb, fr):
            n_art += got != t["y"]
            continue
        assert got == t["y"], (k, t["tags"], t["fa"], t["fb"], t["fr"])
        n += len(got)
    assert n >= 14 * 56 and n_art == 4


def _reduce_tables():
    return [t for t in TABLES if t.get("kind") == "reduce"]


def _reduce_refused(levels):
    # RND::CONV of the 65-bit sum of two 64-bit elements, or a level type of exactly 65 storage bits: artefacts, refused
    return any(l.QuMode == RND.CONV or l.storage_bits == 65 for l in levels)


def _gemm_cases():
    return [t for t in TABLES if "M" in t and "name" in t]


def _reduce_inputs(oracle, t):
    fin = Qu.from_tuple(t["fin"])
    levels = [Qu.from_tuple(x) for x in t["levels"]]
    L = oracle.lib()
    n, rows = t["len"], len(t["seeds"])
    A = np.zeros(n * rows, dtype=oracle.host_dtype(fin))
    for r, seed in enumerate(t["seeds"]):
        A[r * n:(r + 1) * n] = [L.qoracle_synth(fin.c(), seed, t["dist"], i, 0) for i in range(n)]
    return fin, levels, reduce_result_type(fin, levels, n), A, rows


@pytest.mark.parametrize("t", _reduce_tables(), ids=lambda t: t["name"])
def test_oracle_matches_wide_reduce(oracle, t):
    fin, levels, ec, A, rows = _reduce_inputs(oracle, t)
    for (_, fr) in t["y"]:
        assert list(ec.as_tuple()) == fr
    d = lower_reduce(fin, rows, t["len"], levels)
    exp = [y for (y, _) in t["y"]]
    if _reduce_refused(levels):
        assert capi.classify_status(d)[0] == capi.QG_EUNSUPPORTED
        return
    out = oracle.gemm(d, A, np.ones(t["len"], np.int32), ec)
    assert oracle.from_host(out) == exp


def _case_verdict(j):
    """golden GEMM cases the planner must refuse (the reference's result is an artefact there)"""
    return "convC" in j["name"]        # C with RND::CONV from the 76-bit root


@pytest.mark.parametrize("j", _gemm_cases(), ids=lambda j: j["name"])
def test_oracle_and_planner_on_wide_gemm_goldens(oracle, j):
    d = desc_from_dict(j)
    st, info = capi.classify_status(d)
    if _case_verdict(j):
        assert st == capi.QG_EUNSUPPORTED and b"RND::CONV" in info.reason
        return
    assert st == capi.QG_OK, info.reason
    assert info.max_bits > 64
    kn = capi.KERNEL_NAMES[info.kernel]
    assert kn == ("mfma_i8_limb" if "_L_" in j["name"] else "tree_i128"), (kn, info.reason)
    A, B = G.case_inputs(j, oracle)
    _, _, ec = G.case_elems(j)
    got = oracle.gemm(d, A, B, ec)
    assert got.tobytes() == G.case_expected(j, oracle).tobytes()


def test_planner_verdicts_on_wide_descriptors():
    q = Qu(15, 16)
    ok = lambda d: capi.classify_status(d)
    # the "accumulate Q15.16 exactly" call of VERDICT r2: linear class, composite MFMA plan with the 128-bit combine
    st, info = ok(lower(q, q, Qu(43, 32), 256, 256, 4096, mul_args=Tags(31, 32), add_args=[Qu(43, 32)]))
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "mfma_i8_limb" and b"128-bit combine" in info.reason and list(info.limbs) == [4, 4]   # (a full-range 32-bit word needs five PLAIN balanced base-256 digits — 127 * (256^4 - 1) / 255 < 2^31 - 1 — and four once it is stored centred, x - 0x808080)
    assert info.host_elem_bytes[2] == 16 and info.packed_bytes[2] == 256 * 256 * 16
    # int<16,15> linear (What's missing 1)
    e = Qu(16, 15)
    st, info = ok(lower(e, e, Qu(45, 30), 64, 64, 4096, mul_args=Tags(33, 30), add_args=[Qu(45, 30)]))
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "mfma_i8_limb", info.reason
    # refused: a 65-bit format anywhere, RND::CONV of a wide value, Qsub(wide, one-word), operands wider than a word
    assert ok(lower(q, q, Qu(32, 32), 8, 8, 8, mul_args=Tags(31, 32), add_args=[Qu(43, 32)]))[0] == capi.QG_EUNSUPPORTED
    assert ok(lower(q, q, q, 8, 8, 8, mul_args=Tags(31, 32), add_args=[Qu(32, 32)]))[0] == capi.QG_EUNSUPPORTED
    assert ok(lower(q, q, Qu(15, 16, True, RND.CONV), 8, 8, 8, mul_args=Tags(31, 32), add_args=[Qu(43, 32)]))[0] == capi.QG_EUNSUPPORTED
    assert ok(lower(Qu(40, 30), q, q, 8, 8, 8))[0] == capi.QG_EUNSUPPORTED
    c = Qcomplex(q, q)
    wide, narrow = Qu(43, 32), Qu(31, 32)
    st, info = ok(lower(c, c, Qcomplex(wide, wide), 8, 8, 8, mul_args=BasicComplexMul(acT=wide, bdT=narrow, adT=wide, bcT=wide, acbdT=wide, adbcT=wide)))
    assert st == capi.QG_EUNSUPPORTED and b"Qsub" in info.reason
    # complex with wide sub-operation formats everywhere: the 128-bit tree kernel
    st, info = ok(lower(c, c, Qcomplex(wide, wide), 8, 8, 8, mul_args=BasicComplexMul(acT=wide, bdT=wide, adT=wide, bcT=wide, acbdT=wide, adbcT=wide),
                        add_args=[Qcomplex(Qu(46, 32), Qu(46, 32))]))
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i128", info.reason


# ------------------------------------------------------------------------------------------------------------------ GPU
def _run(d, A, B, ec, oracle, **kw):
    out = np.zeros(d.M * d.N, dtype=oracle.host_dtype(ec))
    return capi.run(d, out, A, B, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("j", [j for j in _gemm_cases() if not _case_verdict(j)], ids=lambda j: j["name"])
def test_gpu_wide_gemm_goldens(oracle, j):
    d = desc_from_dict(j)
    A, B = G.case_inputs(j, oracle)
    _, _, ec = G.case_elems(j)
    exp = G.case_expected(j, oracle)
    assert _run(d, A, B, ec, oracle).tobytes() == exp.tobytes()
    # the linear cases also through the 128-bit tree kernel
    if "_L_" in j["name"]:
        assert capi.KERNEL_NAMES[capi.classify(d, capi.OPT_FORCE_TREE).kernel] == "tree_i128"
        assert _run(d, A, B, ec, oracle, flags=capi.OPT_FORCE_TREE).tobytes() == exp.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("t", [t for t in _reduce_tables() if not _reduce_refused([Qu.from_tuple(x) for x in t["levels"]])], ids=lambda t: t["name"])
def test_gpu_wide_reduce(oracle, t):
    fin, levels, ec, A, rows = _reduce_inputs(oracle, t)
    d = lower_reduce(fin, rows, t["len"], levels)
    out = capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, np.ones(t["len"], np.int32))
    assert oracle.from_host(out) == [y for (y, _) in t["y"]]


Q = Qu(15, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("ec", [Qu(43, 32), Q, Qu(15, 16, True, RND.POS_INF, SAT.ZERO), Qu(20, 10, False, TRN.SMGN, SAT.SMGN), Qu(50, 40), Qu(8, 8, True, RND.INF, WRP.TCPL),
                                Qu(20, 32), Qu(25, 30, True, RND.NEG_INF, SAT.ZERO), Qu(30, 31, False, TRN.TCPL, SAT.SMGN)],
                         ids=["wideC", "q1516C", "posinf_zero", "unsigned_smgn", "q5040C", "inf_wrap", "band_q2032", "band_zero", "band_unsigned"])
def test_gpu_q1516_exact_accumulation(oracle, ec):
    """Q15.16 x Q15.16, MulArgs<intBits<31>, fracBits<32>>, AddArgs<Qu<43,32>>, K = 4096: linear class beyond 64 bits, four 2 x 2-limb
    MFMA launches and the 128-bit combine.  With full-range words about 7 % of the 76-bit sums lie in [2^63, 2^64) or
    [-2^64, -2^63), where the reference's comparison with one-word bounds misfires: a one-word C type that keeps (nearly) all
    fraction bits sees it (the band_* cases: counted); Q15.16 itself drops 16 bits first and never gets there."""
    M, N, K = 96, 80, 4096
    d = lower(Q, Q, ec, M, N, K, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])
    info = capi.classify(d)
    assert capi.KERNEL_NAMES[info.kernel] == "mfma_i8_limb" and b"128-bit" in info.reason
    A = oracle.fill(Q, M * K, 3)
    B = oracle.fill(Q, K * N, 4)
    got = _run(d, A, B, ec, oracle)
    exp = oracle.gemm(d, A, B, ec, nthreads=8)
    assert got.tobytes() == exp.tobytes()
    if ec.fracBits >= 30 and ec.storage_bits <= 64:
        # the same descriptor with C = the exact sums tells which outputs were in the band
        dw = lower(Q, Q, Qu(43, 32), M, N, K, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])
        sums = oracle.from_host(oracle.gemm(dw, A, B, Qu(43, 32), nthreads=8))
        sh = 32 - ec.fracBits
        band = sum(1 for v in sums if 2**63 <= (v >> sh) < 2**64 or -2**64 <= (v >> sh) < -2**63)
        assert band > 50, band
        assert info.host_elem_bytes[2] == 8 and info.packed_bytes[2] >= M * N * 8   # (such values come out as their low WORD: 8-byte containers)
    # the tree kernel agrees
    assert _run(d, A, B, ec, oracle, flags=capi.OPT_FORCE_TREE).tobytes() == exp.tobytes()


@pytest.mark.gpu
def test_gpu_wide_tree_class_vs_oracle(oracle):
    cases = [
        # quantising wide levels; product kept exact
        (Q, Q, Qu(43, 32), dict(mul_args=Tags(31, 32), add_args=[Qu(35, 30, True, RND.POS_INF), Qu(38, 28, True, RND.ZERO, SAT.ZERO)]), 40, 24, 300),
        # product rounded from 64 bits, saturating 33.32 levels (default modes), narrow C
        (Q, Q, Q, dict(mul_args=Tags(31, 32), add_args=[Qu(33, 32)]), 33, 17, 1000),
        # unsigned words, FullPrec product (66-bit type), wide level
        (Qu(16, 16, False), Qu(16, 16, False), Qu(40, 32, False), dict(mul_args=Tags(34, 32), add_args=[Qu(38, 30, False, RND.INF)]), 20, 12, 64),
        # 40-bit x 32-bit operands
        (Qu(20, 19), Q, Qu(50, 20, True, RND.NEG_INF, SAT.SMGN), dict(mul_args=Tags(36, 35), add_args=[Qu(47, 30, True, TRN.SMGN)]), 24, 20, 128),
        # signed wrap into a one-word level from a wide sum
        (Q, Q, Qu(43, 32), dict(mul_args=Tags(31, 32), add_args=[Qu(30, 32, True, TRN.TCPL, WRP.TCPL), Qu(43, 32)]), 16, 16, 64),
    ]
    for ea, eb, ec, kw, M, N, K in cases:
        for ta in (False, True):
            d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == "tree_i128", (str(ec), info.reason)
            A = oracle.fill(ea, M * K, 7)
            B = oracle.fill(eb, K * N, 8)
            assert _run(d, A, B, ec, oracle).tobytes() == oracle.gemm(d, A, B, ec, nthreads=8).tobytes(), (str(ec), ta)


@pytest.mark.gpu
def test_gpu_wide_complex(oracle):
    c = Qcomplex(Q, Q)
    w = Qu(43, 32)
    cw = Qcomplex(Qu(46, 32), Qu(46, 32))
    for mul in (BasicComplexMul(acT=w, bdT=w, adT=w, bcT=w, acbdT=w, adbcT=w),
                TFComplexMul(abT=Qu(16, 16), cdT=Qu(16, 16), abcT=w, cdbT=w, badT=w, ABT=w, BCT=w)):
        for ec in (cw, c):
            d = lower(c, c, ec, 20, 12, 100, mul_args=mul, add_args=[cw])
            st, info = capi.classify_status(d)
            assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i128", info.reason
            A = oracle.fill(c, 20 * 100, 11)
            B = oracle.fill(c, 100 * 12, 12)
            got = _run(d, A, B, ec, oracle)
            assert got.tobytes() == oracle.gemm(d, A, B, ec, nthreads=8).tobytes()


@pytest.mark.gpu
def test_gpu_wide_sharded_and_leading_dimensions(oracle):
    """wide C through the row-sharded entry (bands on one card) and with a padded ldc: 16-byte host elements"""
    M, N, K = 600, 40, 256
    ec = Qu(43, 32)
    d = lower(Q, Q, ec, M, N, K, mul_args=Tags(31, 32), add_args=[ec])
    A = oracle.fill(Q, M * K, 5)
    B = oracle.fill(Q, K * N, 6)
    exp = oracle.gemm(d, A, B, ec, nthreads=8)
    got = capi.run_sharded(d, np.zeros(M * N, dtype=oracle.host_dtype(ec)), A, B, [0, 0, 0])
    assert got.tobytes() == exp.tobytes()
    ldc = M + 5
    out = np.zeros(ldc * N, dtype=oracle.host_dtype(ec))
    out["lo"] = 77
    got = capi.run(d, out.copy(), A, B, ldc=ldc)
    e2 = oracle.gemm(d, A, B, ec, ldc=ldc, nthreads=8, out=out.copy())
    assert got.tobytes() == e2.tobytes()
