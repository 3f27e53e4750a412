#!/usr/bin/env python3
"""Opt-in long fuzz run of SHAPES on an MI355X (not collected by pytest): linear-class descriptors (1, 2 and 3 int8 limbs,
mixed limb counts) with sizes around the tile boundaries, long reductions up to the int32-accumulator bound, random
leading dimensions and A orientation, small-range operands (plane masks), against the oracle.
usage: python tests/extended_fuzz_shapes.py [cases] [seed]"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qoracle as oracle  # noqa: E402
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, lower  # noqa: E402

EDGES = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300, 511, 513]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    oracle.lib()
    ran, kernels = 0, {}
    for it in range(cases):
        wa, wb = rng.choice([(7, 7), (14, 14), (16, 16), (7, 16), (16, 7), (14, 7), (12, 16), (22, 22), (22, 7), (11, 11), (10, 11), (9, 9), (11, 8)])
        fa, fb = rng.randint(-2, 6), rng.randint(-2, 6)
        ea = Qu(wa - max(fa, 0) if fa >= 0 else wa, fa, rng.random() < 0.8) if wa - max(fa, 0) >= 0 else Qu(wa, 0)
        eb = Qu(wb - max(fb, 0) if fb >= 0 else wb, fb, rng.random() < 0.8) if wb - max(fb, 0) >= 0 else Qu(wb, 0)
        M, N = rng.choice(EDGES), rng.choice(EDGES)
        K = rng.choice([1, 5, 63, 64, 65, 100, 128, 1000, 4096, 20000, 43000, 131071])
        while M * N * K > 6e7:
            if M >= N and M > 1:
                M = max(1, M // 2)
            elif N > 1:
                N = max(1, N // 2)
            else:
                K //= 2
        pf = Qu(ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits, ea.isSigned or eb.isSigned)
        acc = Qu(pf.intBits + 18, pf.fracBits, pf.isSigned)
        ec = Qu(rng.randint(4, 30), rng.randint(-2, 10), rng.random() < 0.8, rng.randint(0, 6), rng.randint(0, 3))
        ta = rng.random() < 0.5
        try:
            d = lower(ea, eb, ec, M, N, K, mul_args=pf, add_args=[acc], transposed_a=ta)
        except ValueError:
            continue
        st, info = capi.classify_status(d, 0)
        if st != capi.QG_OK:
            continue
        lda = (K if ta else M) + rng.choice([0, 0, 3, 17])
        ldb = K + rng.choice([0, 0, 5])
        ldc = M + rng.choice([0, 0, 7])
        dist = rng.choice([0, 0, 1])
        A = oracle.fill(ea, lda * (M if ta else K), rng.randint(1, 1 << 30), dist)
        B = oracle.fill(eb, ldb * N, rng.randint(1, 1 << 30), dist)
        out = np.full(ldc * N, -99, dtype=oracle.host_dtype(ec))
        exp = out.copy()
        capi.run(d, out, A, B, lda=lda, ldb=ldb, ldc=ldc)
        oracle.gemm(d, A, B, ec, lda=lda, ldb=ldb, ldc=ldc, out=exp, nthreads=8)
        k = capi.KERNEL_NAMES[info.kernel]
        if not np.array_equal(out, exp):
            print(json.dumps({"mismatch": it, "kernel": k, "M": M, "N": N, "K": K, "ta": ta, "ld": [lda, ldb, ldc], "dist": dist,
                              "a": str(ea), "b": str(eb), "c": str(ec), "limbs": [info.limbs[0], info.limbs[1]]}), flush=True)
            sys.exit(1)
        kernels[f"{k}:{info.limbs[0]}x{info.limbs[1]}"] = kernels.get(f"{k}:{info.limbs[0]}x{info.limbs[1]}", 0) + 1
        ran += 1
    print(json.dumps({"shapes_run": ran, "kernels": kernels, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
