"""Seeded random descriptors on the GPU against the CPU restatement (run with -m gpu).

The oracle is pinned by the reference's golden vectors (tests/test_oracle_golden.py); this sweep then
pushes several hundred random combinations of element formats (signed/unsigned, negative fracBits),
QgemulMulArgs tag subsets (incl. FullPrec and full types), per-level type lists with mixed modes,
reduction lengths (powers of two and not), shapes, A orientation, input distributions and complex
multipliers with per-sub-op tags through every kernel the planner can choose.  Bit-exact."""
import random

import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, Tags, TFComplexMul, lower

pytestmark = pytest.mark.gpu

KS = [1, 2, 3, 5, 8, 16, 31, 32, 33, 64, 100, 128, 256, 512]


def rand_qu(rng, max_w=12, allow_neg_frac=True):
    F = rng.randint(-3 if allow_neg_frac else 0, 9)
    I = rng.randint(max(0, -F), max(1, max_w - max(F, 0)))
    return Qu(I, F, rng.random() < 0.75, rng.randint(0, 6), rng.randint(0, 3))


def rand_tags(rng, base: Qu):
    if rng.random() < 0.3:
        return None
    if rng.random() < 0.3:
        return rand_qu(rng, 16)
    t = {}
    if rng.random() < 0.5:
        t["intBits"] = rng.randint(max(0, -base.fracBits), 14)
    if rng.random() < 0.5:
        t["fracBits"] = rng.randint(max(-3, -t.get("intBits", 3)), 10)
    if rng.random() < 0.3:
        t["isSigned"] = rng.random() < 0.7
    if rng.random() < 0.5:
        t["QuMode"] = rng.randint(0, 6)
    if rng.random() < 0.5:
        t["OfMode"] = rng.randint(0, 3)
    if rng.random() < 0.2:
        t["FullPrec"] = True
    return Tags(**t)


def fields_equal(a, b):
    if a.dtype.names:
        return all(np.array_equal(a[n], b[n]) for n in a.dtype.names)
    return np.array_equal(a, b)


def run_case(oracle, rng, ea, eb, ec, kw, kernels_seen):
    M, N, K = rng.randint(1, 70), rng.randint(1, 70), rng.choice(KS)
    dist = rng.randint(0, 1)
    try:
        d = lower(ea, eb, ec, M, N, K, transposed_a=rng.random() < 0.5, **kw)
    except ValueError:
        return False
    flags = rng.choice([0, 0, 0, capi.OPT_FORCE_TREE, capi.OPT_GENERIC_TREE, capi.OPT_RUNTIME_MODES])
    st, info = capi.classify_status(d, flags)
    if st != capi.QG_OK:
        return False
    A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
    B = oracle.fill(eb, K * N, rng.randint(1, 1 << 30), dist)
    got = capi.run(d, np.zeros(M * N, dtype=oracle.host_dtype(ec)), A, B, flags=flags)
    exp = oracle.gemm(d, A, B, ec, nthreads=4)
    assert fields_equal(got, exp), (ea, eb, ec, kw, M, N, K, dist, flags, capi.KERNEL_NAMES[info.kernel])
    kernels_seen.add(capi.KERNEL_NAMES[info.kernel])
    return True


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_real(oracle, seed):
    rng = random.Random(1000 + seed)
    seen, n = set(), 0
    for _ in range(60):
        ea, eb = rand_qu(rng), rand_qu(rng)
        if rng.random() < 0.4:
            eb = ea
        kind = rng.random()
        if kind < 0.35:      # linear-class shaped tags: exact product, wide accumulators
            pf = Qu(ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits, ea.isSigned or eb.isSigned)
            kw = dict(mul_args=pf, add_args=[Qu(pf.intBits + 10, pf.fracBits, pf.isSigned)])
        else:
            lv = [rand_qu(rng, 14) for _ in range(rng.randint(0, 3))]
            kw = dict(mul_args=rand_tags(rng, ea), add_args=lv or None)
        ec = rand_qu(rng, 16)
        n += run_case(oracle, rng, ea, eb, ec, kw, seen)
    assert n >= 30
    assert len(seen) >= 2, seen


def rand_cplx(rng):
    return Qcomplex(rand_qu(rng, 9), rand_qu(rng, 9))


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_complex(oracle, seed):
    rng = random.Random(5000 + seed)
    seen, n = set(), 0
    for _ in range(40):
        ea = rand_cplx(rng)
        eb = ea if rng.random() < 0.5 else rand_cplx(rng)
        ec = Qcomplex(rand_qu(rng, 16), rand_qu(rng, 16))
        def sub():
            return rand_qu(rng, 14) if rng.random() < 0.5 else None
        if rng.random() < 0.5:
            m = TFComplexMul(abT=sub(), cdT=sub(), baT=sub(), abcT=sub(), cdbT=sub(), badT=sub(), ABT=sub(), BCT=sub())
        else:
            m = BasicComplexMul(acT=sub(), bdT=sub(), adT=sub(), bcT=sub(), acbdT=sub(), adbcT=sub())
        if rng.random() < 0.2:
            m = None
        lv = [Qcomplex(rand_qu(rng, 14), rand_qu(rng, 14)) for _ in range(rng.randint(0, 2))]
        n += run_case(oracle, rng, ea, eb, ec, dict(mul_args=m, add_args=lv or None), seen)
    assert n >= 20
    assert seen <= {"tree_cplx", "tree_cplx_i32"} and seen
