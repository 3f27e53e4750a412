"""The CPU restatement (oracle/qoracle.c) against everything the reference pins:
 - the 40 known answers of the reference's own rounding tests (tests/golden/ref_rounding_kat.json)
 - scalar truth tables and GEMM results produced by the real reference header in the build
   container (tests/golden/ref_scalar_*.jsonl.gz, ref_gemm_*.jsonl.gz; oracle/gen_golden.py).
Bit-exact: integer work.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import golden_io as G
from qublas_amd.desc import Qu, desc_from_dict, qfmt


def test_reference_rounding_known_answers(oracle):
    L = oracle.lib()
    cases = G.rounding_kat()
    assert len(cases) == 40
    for c in cases:
        src, dst = Qu.from_tuple(c["src"]), Qu.from_tuple(c["dst"])
        raw = c["value"] * 2.0 ** src.fracBits
        assert raw == int(raw)
        raw = int(raw)
        got = L.qoracle_convert128(raw >> 64, raw & (2**64 - 1), src.c(), dst.c())
        assert got / 2.0 ** dst.fracBits == c["expect"], c["test"]


def width_artefact(src: Qu, dst: Qu) -> bool:
    """A rounding shift of exactly 32 (or 64) bits with an RND mode: the reference builds its
    masks from ArbiInt<32>::allOnes()/allZeros(), which return -1/0 instead of 2^32-1 / ~(2^32-1)
    (QuBLAS.h:361-377, :388-402), so its result there is a width artefact, not rounding.  The
    engine rejects such descriptors (QG_EUNSUPPORTED); the tables document that it is only them."""
    return (src.fracBits - dst.fracBits) in (32, 64) and dst.QuMode <= 4


def test_tcpl_sat_stub_tables(oracle):
    """tests/golden/ref_scalar_9: WRP::TCPL_SAT<N> targets — the reference's intConvert stub returns its input and the target's
    storage word (int32_t / int64_t) keeps what fits"""
    L = oracle.lib()
    n = 0
    for t in G.scalar_tables(9):
        f, to = Qu.from_tuple(t["from"]).c(), Qu.from_tuple(t["to"]).c()
        assert to.O == 4
        for x, y in zip(range(t["lo"], t["hi"] + 1, t["step"]), t["y"]):
            assert L.qoracle_convert(x, f, to) == y, (t["from"], t["to"], x)
            n += 1
    assert n > 3000


@pytest.mark.parametrize("part", [0, 1, 2])
def test_convert_truth_tables(oracle, part):
    L = oracle.lib()
    n = 0
    for t in G.scalar_tables(part):
        assert t["kind"] == "cvt"
        if width_artefact(Qu.from_tuple(t["from"]), Qu.from_tuple(t["to"])):
            continue
        f, to = Qu.from_tuple(t["from"]).c(), Qu.from_tuple(t["to"]).c()
        xs = range(t["lo"], t["hi"] + 1, t["step"])
        assert len(xs) == len(t["y"])
        for x, y in zip(xs, t["y"]):
            assert L.qoracle_convert(x, f, to) == y, (t["from"], t["to"], x)
            n += 1
    assert n > 10000


def test_wrap_into_32_bits_is_a_reference_artefact_only_for_unsigned(oracle):
    """tests/golden/ref_scalar_6: WRP::TCPL targets around 32 bits, produced by the reference.  Unsigned targets of EXACTLY 32
    value bits come back unwrapped (the mask is ArbiInt<32>::allOnes() = -1, QuBLAS.h:361-377, :2328-2331): the engine rejects
    such conversions (qg_plan.cpp, tests/test_cabi_cpu.py::test_rejections) and the oracle keeps the arithmetic definition.
    Every other width, and the signed 32-storage-bit target, agree with the oracle value for value."""
    L = oracle.lib()
    seen_artefact = seen_plain = 0
    for t in G.scalar_tables(6):
        src, dst = Qu.from_tuple(t["from"]), Qu.from_tuple(t["to"])
        xs = range(t["lo"], t["hi"] + 1, t["step"])
        assert len(xs) == len(t["y"])
        artefact = (not dst.isSigned) and dst.intBits + dst.fracBits == 32
        for x, y in zip(xs, t["y"]):
            got = L.qoracle_convert(x, src.c(), dst.c())
            if artefact:
                assert y == x << (dst.fracBits - src.fracBits)            # the reference stores the value unwrapped
                assert got == y % (1 << 32)                               # the oracle wraps
                seen_artefact += got != y
            else:
                assert got == y, (t["from"], t["to"], x)
                seen_plain += 1
    assert seen_artefact > 100 and seen_plain > 1000


def test_reference_artefacts_flag_reproduces_the_unwrapped_32_bit_results(oracle):
    """QG_DESC_REFERENCE_ARTEFACTS (include/qgemul.h): with the flag, C of an unsigned WRP::TCPL format with exactly 32 value bits
    is what the reference stores — the value unwrapped (tests/golden/ref_scalar_6, produced by the reference).  Run as a K = 1
    Qgemul (x * 1 in x's own format, then the conversion into C) through the oracle; the planner admits the descriptor with the
    flag and refuses it without; every other table of the file is untouched by the flag."""
    from qublas_amd import capi
    from qublas_amd.desc import ONE, lower
    import numpy as np
    seen = other = 0
    for t in G.scalar_tables(6):
        src, dst = Qu.from_tuple(t["from"]), Qu.from_tuple(t["to"])
        xs = np.arange(t["lo"], t["hi"] + 1, t["step"], dtype=np.int64)
        artefact = (not dst.isSigned) and dst.intBits + dst.fracBits == 32
        d = lower(src, ONE, dst, len(xs), 1, 1, mul_args=src, reference_artefacts=True)
        st, _ = capi.classify_status(d)
        if st != capi.QG_OK:
            assert not artefact, (t["from"], t["to"])
            continue
        A = xs.astype(oracle.host_dtype(src))
        got = oracle.gemm(d, A, np.ones(1, np.int32), dst)
        assert [int(v) for v in got] == t["y"], (t["from"], t["to"])
        if artefact:
            st0, info0 = capi.classify_status(lower(src, ONE, dst, len(xs), 1, 1, mul_args=src))
            assert st0 == capi.QG_EUNSUPPORTED or max(t["y"]) < 1 << 32, info0.reason
            seen += 1
        else:
            other += 1
    assert seen >= 3 and other >= 3


def test_mul_add_truth_tables(oracle):
    L = oracle.lib()
    kinds = set()
    for t in G.scalar_tables(3):
        fa, fb, fr = (Qu.from_tuple(t[k]) for k in ("fa", "fb", "fr"))
        it = iter(t["y"])
        for a in range(fa.raw_min, fa.raw_max + 1):
            for b in range(fb.raw_min, fb.raw_max + 1):
                y = next(it)
                if t["kind"] == "mul":
                    got = L.qoracle_mul(a, fa.c(), b, fb.c(), fr.c())
                else:
                    got = L.qoracle_add(a, fa.c(), b, fb.c(), fr.c(), int(t["kind"] == "sub"))
                assert got == y, (t["kind"], t["tags"], a, b)
        kinds.add(t["kind"])
    assert kinds == {"mul", "add", "sub"}


def test_reduce_truth_tables(oracle):
    L = oracle.lib()
    for t in G.scalar_tables(4):
        fin = Qu.from_tuple(t["fin"])
        levels = [Qu.from_tuple(x) for x in t["levels"]]
        lv = (qfmt * max(1, len(levels)))(*[x.c() for x in levels])
        for seed, (y, fr) in zip(t["seeds"], t["y"]):
            v = np.array([L.qoracle_synth(fin.c(), seed, t["dist"], i, 0) for i in range(t["len"])], dtype=np.int64)
            got = L.qoracle_reduce(v.ctypes.data_as(C.POINTER(C.c_int64)), t["len"], fin.c(), lv, len(levels))
            assert got == y, (t["name"], seed)


def _check_gemm(oracle, j):
    d = desc_from_dict(j)
    A, B = G.case_inputs(j, oracle)
    _, _, ec = G.case_elems(j)
    got = oracle.gemm(d, A, B, ec)
    exp = G.case_expected(j, oracle)
    assert got.tobytes() == exp.tobytes(), j["name"]


@pytest.mark.parametrize("j", G.gemm_cases("real"), ids=lambda j: j["name"])
def test_gemm_real_golden(oracle, j):
    _check_gemm(oracle, j)


@pytest.mark.parametrize("j", G.gemm_cases("cplx"), ids=lambda j: j["name"])
def test_gemm_complex_golden(oracle, j):
    _check_gemm(oracle, j)


def test_config1_known_answer(oracle):
    """SURVEY.md §8-a known answer for BASELINE.json configuration 1 (README shapes)."""
    by = {j["name"]: j for j in G.gemm_cases("real")}
    assert by["c1_nn_classT"]["C"] == [23040, 25600, 28160, 30720, 51712, 58368, 65024, 0, 0, 19968, 24832, 29696,
                                       21248, 28160, 35072, 0]
    assert by["c1_tn_classT"]["C"] == [7680, 17920, 28160, 38400, 17920, 44544, 0, 0, 28160, 0, 46336, 0, 38400, 0,
                                       0, 57600]
