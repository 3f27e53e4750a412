#!/usr/bin/env python3
"""Opt-in long fuzz run on an MI355X (not collected by pytest): thousands of seeded random descriptors through the C-ABI
against the CPU restatement, with wider formats, arbitrary reduction lengths and more one-column shapes than the sweep of
tests/test_gpu_fuzz.py.  usage: python tests/extended_fuzz.py [cases] [seed]; prints a JSON summary, exits 1 on a mismatch."""
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
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, Tags, TFComplexMul, lower  # noqa: E402
from test_gpu_fuzz import fields_equal, rand_qu, rand_tags  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260)
    oracle.lib()
    seen, ran, skipped = {}, 0, 0
    for it in range(cases):
        cx = rng.random() < 0.15
        wide = rng.random() < 0.3
        words = rng.random() < 0.12          # operands of up to 32 storage bits: the unrounded product uses all of an int64
        mw = rng.choice([26, 28, 30, 31]) if words else 22 if wide else 12
        if cx:
            pw = rng.choice([24, 28, 31]) if words else 9
            ea = Qcomplex(rand_qu(rng, pw), rand_qu(rng, pw))
            eb = ea if rng.random() < 0.5 else Qcomplex(rand_qu(rng, pw), rand_qu(rng, pw))
            ec = Qcomplex(rand_qu(rng, 16), rand_qu(rng, 16))
            sub = lambda: rand_qu(rng, 14) if rng.random() < 0.5 else None  # noqa: E731
            m = (TFComplexMul(abT=sub(), cdT=sub(), baT=sub(), abcT=sub(), cdbT=sub(), badT=sub(), ABT=sub(), BCT=sub()) if rng.random() < 0.5
                 else BasicComplexMul(acT=sub(), bdT=sub(), adT=sub(), bcT=sub(), acbdT=sub(), adbcT=sub()))
            lv = [Qcomplex(rand_qu(rng, 14), rand_qu(rng, 14)) for _ in range(rng.randint(0, 2))]
            kw = dict(mul_args=m if rng.random() < 0.8 else None, add_args=lv or None)
        else:
            ea, eb = rand_qu(rng, mw), rand_qu(rng, mw)
            if rng.random() < 0.4:
                eb = ea
            if rng.random() < 0.35:
                pf = Qu(ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits, ea.isSigned or eb.isSigned)
                kw = dict(mul_args=pf, add_args=[Qu(pf.intBits + 12, pf.fracBits, pf.isSigned)])
            else:
                lv = [rand_qu(rng, 40 if (wide or words) else 14) for _ in range(rng.randint(0, 3))]
                kw = dict(mul_args=rand_tags(rng, ea), add_args=lv or None)
            ec = rand_qu(rng, rng.choice([30, 36, 44]) if wide else rng.choice([16, 16, 34]))
        M = rng.randint(1, 150)
        N = 1 if rng.random() < 0.2 else rng.randint(1, 150)
        K = rng.choice([rng.randint(1, 40), rng.randint(1, 600), rng.randint(1, 5000)]) if not cx else rng.randint(1, 300)
        dist = rng.randint(0, 1)
        try:
            d = lower(ea, eb, ec, M, N, K, transposed_a=rng.random() < 0.5, **kw)
        except ValueError:
            skipped += 1
            continue
        flags = rng.choice([0, 0, 0, 0, capi.OPT_FORCE_TREE, capi.OPT_GENERIC_TREE, capi.OPT_RUNTIME_MODES])
        st, info = capi.classify_status(d, flags)
        if st != capi.QG_OK:
            skipped += 1
            continue
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
        B = oracle.fill(eb, K * N, rng.randint(1, 1 << 30), dist)
        got = capi.run(d, np.zeros(M * N, dtype=oracle.host_dtype(ec)), A, B, flags=flags)
        exp = oracle.gemm(d, A, B, ec, nthreads=8)
        kname = capi.KERNEL_NAMES[info.kernel]
        if not fields_equal(got, exp):
            print(json.dumps({"mismatch": it, "kernel": kname, "M": M, "N": N, "K": K, "flags": flags, "dist": dist,
                              "a": str(ea), "b": str(eb), "c": str(ec), "kw": str(kw)}), flush=True)
            sys.exit(1)
        seen[kname] = seen.get(kname, 0) + 1
        ran += 1
        if it % 250 == 0:
            print(json.dumps({"progress": it, "ran": ran}), flush=True)
    print(json.dumps({"cases_run": ran, "skipped_unsupported": skipped, "kernels": seen, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
