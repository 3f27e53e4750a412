#!/usr/bin/env python3
"""Opt-in fuzz of the complex tree kernel's FIXED-mode step forms on an MI355X (not collected by pytest): complex descriptors
whose every step rounds by "add a constant, shift right" (TRN::TCPL, RND::POS_INF, RND::NEG_INF) and overflows by one clamp
(SAT::TCPL, SAT::SMGN) — random part formats (negative fracBits included), Basic / TF with
random sub-operation tags, 0..2 level types, any K >= 17 — so that the planner picks the compact branch-free steps, or the
table-driven fixed steps where a condition of the compact form fails (left shifts at tree nodes, per-level formats).  Each
case: GPU against the oracle, and against the same plan with run-time modes (QG_OPT_RUNTIME_MODES).
With a third argument "all" the formats draw from every QuMode and OfMode (the compact form's rounding / overflow kinds).
usage: python tests/extended_fuzz_cplx_fixed.py [cases] [seed] [all]"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import qoracle as oracle  # noqa: E402
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, RND, SAT, TRN, WRP, Tags, TFComplexMul, lower  # noqa: E402

ALL_MODES = len(sys.argv) > 3 and sys.argv[3] == "all"


def rq(rng, bits):
    """a real format with `bits` value bits split at random, RND::POS_INF + SAT::TCPL"""
    i = rng.randint(0, bits)
    f = bits - i
    if rng.random() < 0.25:
        shift = rng.randint(1, 3)
        i, f = i + shift, f - shift          # negative or smaller fracBits at the same width
    # modes the compact steps cover: "add a constant, shift right" roundings and one-clamp overflows (mixed modes between
    # two operands merge to the reference's defaults TRN::TCPL / SAT::TCPL, which are among them)
    if ALL_MODES:   # (argv[3] == "all") ... and the modes the compact steps cover as rounding / overflow KINDS
        return Qu(i, f, rng.random() < 0.85, rng.choice([RND.POS_INF, TRN.TCPL, RND.NEG_INF, RND.ZERO, RND.INF, RND.CONV, TRN.SMGN]),
                  rng.choice([SAT.TCPL, SAT.SMGN, SAT.ZERO, WRP.TCPL]))
    return Qu(i, f, rng.random() < 0.85, rng.choice([RND.POS_INF, RND.POS_INF, TRN.TCPL, TRN.TCPL, RND.NEG_INF]), rng.choice([SAT.TCPL, SAT.TCPL, SAT.SMGN]))


def rtag(rng, like):
    """loose tags that keep the fixed modes: widths only, or nothing"""
    r = rng.random()
    if r < 0.4:
        return None
    if r < 0.7:
        return Tags(intBits=like.intBits + rng.randint(0, 6))
    if r < 0.9:
        return Tags(intBits=like.intBits + rng.randint(0, 6), fracBits=like.fracBits + rng.randint(-2, 3))
    return rq(rng, min(20, like.intBits + like.fracBits + rng.randint(0, 6)) if like.intBits + like.fracBits > 0 else 6)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
    oracle.lib()
    ran = skipped = 0
    forms, kernels = {}, {}
    for it in range(cases):
        ea = Qcomplex(rq(rng, rng.choice([5, 7, 9, 11])), rq(rng, rng.choice([5, 7, 9, 11])))
        eb = ea if rng.random() < 0.5 else Qcomplex(rq(rng, rng.choice([5, 8, 10])), rq(rng, rng.choice([5, 8, 10])))
        if rng.random() < 0.5:
            mul = TFComplexMul(abT=rtag(rng, ea.real), cdT=rtag(rng, eb.real), abcT=rtag(rng, ea.real), badT=rtag(rng, eb.real),
                               cdbT=rtag(rng, ea.imag), ABT=rtag(rng, ea.real), BCT=rtag(rng, ea.imag), loose=rtag(rng, ea.real))
        else:
            mul = BasicComplexMul(acT=rtag(rng, ea.real), bdT=rtag(rng, ea.imag), adT=rtag(rng, ea.real), bcT=rtag(rng, ea.imag),
                                  acbdT=rtag(rng, ea.real), adbcT=rtag(rng, ea.imag), loose=rtag(rng, ea.real))
        levels = [Qcomplex(rq(rng, rng.choice([10, 14, 18, 22])), rq(rng, rng.choice([10, 14, 18, 22]))) for _ in range(rng.choice([0, 0, 1, 1, 2]))]
        if rng.random() < 0.2:   # one format for every value of the k loop: the "one clamp for the whole loop" form, or just short of it
            u = rq(rng, rng.choice([6, 9, 10, 13, 16]))
            if rng.random() < 0.5:   # narrow formats and operands: the packed 16-bit form needs the common format's bits + the products' shifts <= 16
                u = rq(rng, rng.choice([5, 6, 7, 8, 9]))
                ea = Qcomplex(rq(rng, rng.choice([3, 4, 5, 6])), rq(rng, rng.choice([3, 4, 5, 6])))
                eb = ea if rng.random() < 0.5 else Qcomplex(rq(rng, rng.choice([3, 4, 5, 6])), rq(rng, rng.choice([3, 4, 5, 6])))
            v = u if rng.random() < 0.8 else rq(rng, 12)
            mul = TFComplexMul(abT=rtag(rng, ea.real), cdT=rtag(rng, eb.real), abcT=u, cdbT=u, badT=u, ABT=u, BCT=v)
            less = lambda: Qu(u.intBits, u.fracBits - rng.choice([0, 0, 1, 2, 6]), u.isSigned, u.QuMode, u.OfMode)   # same integer bits, fewer fraction bits: its own mask in the justified forms
            if rng.random() < 0.3:
                mul = TFComplexMul(abT=rtag(rng, ea.real), cdT=rtag(rng, eb.real), abcT=less(), cdbT=less(), badT=less(), ABT=u, BCT=v)
            if rng.random() < 0.5:
                mul = BasicComplexMul(acT=less(), bdT=less(), adT=less(), bcT=less(), acbdT=u, adbcT=v)
            levels = [Qcomplex(u, u) for _ in range(rng.choice([0, 1, 1, 2]))]
        ec = Qcomplex(rq(rng, rng.choice([7, 12, 18])), rq(rng, rng.choice([7, 12, 18])))
        M, N = rng.randint(1, 120), rng.randint(1, 120)
        K = rng.choice([17, 32, 33, 64, 100, 128, 250, 256, 512, 1000])
        try:
            d = lower(ea, eb, ec, M, N, K, mul_args=mul, add_args=levels or None, transposed_a=rng.random() < 0.5)
        except ValueError:
            skipped += 1
            continue
        st, info = capi.classify_status(d)
        if st != capi.QG_OK:
            skipped += 1
            continue
        k = capi.KERNEL_NAMES[info.kernel]
        kernels[k] = kernels.get(k, 0) + 1
        form = info.reason.decode().split("steps: ")[-1] if k == "tree_cplx_i32" else k
        forms[form] = forms.get(form, 0) + 1
        dist = rng.randint(0, 1)
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
        B = oracle.fill(eb, K * N, rng.randint(1, 1 << 30), dist)
        out = np.zeros(M * N, dtype=oracle.host_dtype(ec))
        capi.run(d, out, A, B)
        exp = oracle.gemm(d, A, B, ec, nthreads=8)
        rt = np.zeros(M * N, dtype=oracle.host_dtype(ec))
        capi.run(d, rt, A, B, flags=capi.OPT_RUNTIME_MODES)
        ok = all(np.array_equal(out[p], exp[p]) and np.array_equal(rt[p], exp[p]) for p in ("re", "im"))
        if not ok:
            print(json.dumps({"mismatch": it, "kernel": k, "form": form, "M": M, "N": N, "K": K, "a": str(ea), "b": str(eb), "c": str(ec),
                              "mul": str(mul), "levels": str(levels)}), flush=True)
            sys.exit(1)
        ran += 1
    print(json.dumps({"complex_fixed_mode_cases": ran, "skipped_unsupported": skipped, "kernels": kernels, "step_forms": forms, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
