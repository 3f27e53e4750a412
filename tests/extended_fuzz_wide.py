#!/usr/bin/env python3
"""Opt-in long fuzz run of round 3's new plans on an MI355X (not collected by pytest), against the oracle:
  * composite linear plans: operands of 1 ... 8 int8 limbs in every group shape, reduction lengths around and beyond the k-chunk
    bound, ragged shapes, leading dimensions, both A orientations, random C modes; exact sums within 62 bits (64-bit combine) and
    beyond (128-bit combine), C of 1 ... 16 bytes;
  * wide tree class (`tree_i128`): random products / level lists whose values pass 64 bits, every QuMode / OfMode the planner
    admits there, real and complex (Basic / TF with random sub-operation tags), one-word and two-word C;
  * WRP::TCPL_SAT as C's overflow mode.
Descriptors the planner refuses (artefacts of the reference, combinations it cannot compile, > 120 bits) are skipped and counted.
usage: python tests/extended_fuzz_wide.py [cases] [seed]"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qoracle as oracle  # noqa: E402
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, Tags, TFComplexMul, lower  # noqa: E402

EDGES = [1, 2, 31, 33, 64, 65, 127, 129, 200, 257]


def rand_fmt(rng, bits, signed=None, modes=True):
    """a format of `bits` value bits with a random split into int / frac and (optionally) random modes"""
    f = rng.randint(-2, max(0, bits - 1))
    s = (rng.random() < 0.8) if signed is None else signed
    return Qu(bits - f, f, s, rng.randint(0, 6) if modes else 5, rng.choice([0, 0, 1, 2, 3]) if modes else 0)


def shrink(M, N, K, budget):
    while M * N * K > budget:
        if M >= N and M > 1:
            M = max(1, M // 2)
        elif N > 1:
            N = max(1, N // 2)
        else:
            K = max(1, K // 2)
    return M, N, K


def linear_case(rng):
    wa = rng.choice([7, 8, 12, 15, 16, 17, 22, 23, 24, 25, 30, 31, 32, 33, 39, 40, 47, 55, 61])   # (value bits; signed formats of 16 / 24 / 32 / 40 / 48 bits and unsigned ones are stored centred)
    wb = rng.choice([7, 7, 8, 12, 15, 16, 22, 23, 24, 28, 31, 36, 45])
    if rng.random() < 0.5:
        wa, wb = wb, wa
    ea, eb = rand_fmt(rng, wa, modes=False), rand_fmt(rng, wb, modes=False)
    M, N = rng.choice(EDGES), rng.choice(EDGES)
    K = rng.choice([1, 7, 64, 100, 1000, 4096, 43519, 43521, 50000, 65280, 90000, 130816, 131000, 140000])
    budget = 3e7 if wa + wb <= 48 else 1.2e7
    M, N, K = shrink(M, N, K, budget)
    pf = Qu(ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits, ea.isSigned or eb.isSigned)
    lev = max(1, (K - 1).bit_length())
    acc = Qu(pf.intBits + lev, pf.fracBits, pf.isSigned)
    if rng.random() < 0.5:
        ec = rand_fmt(rng, rng.randint(4, 30))
    elif rng.random() < 0.5:
        ec = rand_fmt(rng, rng.randint(33, 61))
    else:
        ec = rand_fmt(rng, rng.randint(66, 110))
    if rng.random() < 0.1:
        ec = Qu(ec.intBits, ec.fracBits, ec.isSigned, ec.QuMode, 4)          # WRP::TCPL_SAT
    return ea, eb, ec, M, N, K, dict(mul_args=pf, add_args=[acc])


def tree_case(rng):
    wa, wb = rng.choice([(31, 31), (31, 16), (24, 40), (32, 32), (36, 20), (31, 31), (45, 17)])
    ea, eb = rand_fmt(rng, wa, modes=False), rand_fmt(rng, wb, modes=False)
    M, N = rng.choice([1, 3, 17, 33]), rng.choice([1, 2, 9, 20])
    K = rng.choice([1, 2, 5, 16, 37, 64, 100, 300])
    fp = ea.fracBits + eb.fracBits
    prod = Qu(rng.randint(wa + wb - fp - 6, wa + wb - fp + 2), fp - rng.choice([0, 0, 1, 3, 9]), True, rng.randint(0, 6), rng.choice([0, 0, 1, 2]))
    levels = []
    for _ in range(rng.randint(1, 3)):
        w = rng.randint(50, 100)
        f = prod.fracBits - rng.choice([0, 0, 2, 5]) + rng.choice([0, 0, 0, 4])
        levels.append(Qu(w - f, f, True, rng.randint(0, 6), rng.choice([0, 0, 1, 2, 3])))
    ec = rand_fmt(rng, rng.choice([20, 31, 45, 61, 70, 90, 110]))
    return ea, eb, ec, M, N, K, dict(mul_args=prod, add_args=levels)


def cplx_case(rng):
    q = Qu(15, 16)
    c = Qcomplex(q, rand_fmt(rng, 31, True, modes=False))
    w = lambda: Qu(rng.randint(40, 60), rng.randint(28, 34), True, rng.choice([0, 1, 2, 3, 5, 6]), rng.choice([0, 0, 1, 2]))
    if rng.random() < 0.5:
        mul = BasicComplexMul(acT=w(), bdT=w(), adT=w(), bcT=w(), acbdT=w(), adbcT=w())
    else:
        mul = TFComplexMul(abT=Qu(16, 16), cdT=Qu(17, 16), abcT=w(), cdbT=w(), badT=w(), ABT=w(), BCT=w())
    lv = Qcomplex(w(), w())
    ec = Qcomplex(rand_fmt(rng, rng.choice([31, 50, 80])), rand_fmt(rng, rng.choice([20, 61, 100])))
    return c, c, ec, rng.choice([1, 5, 20]), rng.choice([1, 4, 12]), rng.choice([1, 3, 8, 33, 100]), dict(mul_args=mul, add_args=[lv])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    oracle.lib()
    ran, refused, kinds = 0, 0, {}
    for it in range(cases):
        kind = rng.choice(["linear", "linear", "tree", "cplx"])
        ea, eb, ec, M, N, K, kw = {"linear": linear_case, "tree": tree_case, "cplx": cplx_case}[kind](rng)
        ta = rng.random() < 0.5
        try:
            d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
        except ValueError:
            continue
        st, info = capi.classify_status(d, 0)
        if st != capi.QG_OK:
            refused += 1
            continue
        lda = (K if ta else M) + rng.choice([0, 0, 3])
        ldb = K + rng.choice([0, 0, 5])
        ldc = M + rng.choice([0, 0, 7])
        dist = rng.choice([0, 0, 1])
        A = oracle.fill(ea, lda * (M if ta else K), rng.randint(1, 1 << 30), dist)
        B = oracle.fill(eb, ldb * N, rng.randint(1, 1 << 30), dist)
        out = np.zeros(ldc * N, dtype=oracle.host_dtype(ec))
        out.view(np.uint8)[:] = 0x5a
        exp = out.copy()
        capi.run(d, out, A, B, lda=lda, ldb=ldb, ldc=ldc)
        oracle.gemm(d, A, B, ec, lda=lda, ldb=ldb, ldc=ldc, out=exp, nthreads=8)
        k = capi.KERNEL_NAMES[info.kernel]
        tag = f"{kind}:{k}:{info.limbs[0]}x{info.limbs[1]}" + (":wide" if info.max_bits > 62 else "")
        cplx = out.dtype.names is not None and "lo" not in out.dtype.names
        same = all(out[n].tobytes() == exp[n].tobytes() for n in out.dtype.names) if cplx else out.tobytes() == exp.tobytes()   # (a struct's padding bytes belong to nobody)
        if not same:
            diffs = []
            for part in (out.dtype.names if out.dtype.names and "lo" not in out.dtype.names else [None]):
                a, b = (out[part], exp[part]) if part else (out, exp)
                ga, gb = oracle.from_host(a), oracle.from_host(b)
                diffs += [(part, i, x, y) for i, (x, y) in enumerate(zip(ga, gb)) if x != y][:4]
            print(json.dumps({"diffs (part, index, engine, oracle)": diffs}), flush=True)
            print(json.dumps({"mismatch": it, "kind": tag, "M": M, "N": N, "K": K, "ta": ta, "ld": [lda, ldb, ldc], "dist": dist,
                              "a": str(ea), "b": str(eb), "c": str(ec), "kw": str(kw), "reason": info.reason.decode()}), flush=True)
            sys.exit(1)
        if kind == "linear" and rng.random() < 0.3:   # centred operands against the plain balanced limbs: the same bytes
            bal = np.zeros_like(out)
            bal.view(np.uint8)[:] = 0x5a
            capi.run(d, bal, A, B, lda=lda, ldb=ldb, ldc=ldc, flags=capi.OPT_BALANCED_LIMBS)
            if bal.tobytes() != out.tobytes():
                print(json.dumps({"mismatch_vs_balanced_limbs": it, "kind": tag, "M": M, "N": N, "K": K, "a": str(ea), "b": str(eb), "c": str(ec)}), flush=True)
                sys.exit(1)
        kinds[tag] = kinds.get(tag, 0) + 1
        ran += 1
    print(json.dumps({"wide_and_composite_run": ran, "refused_by_planner": refused, "kinds": kinds, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
