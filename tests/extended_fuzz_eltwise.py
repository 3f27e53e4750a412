#!/usr/bin/env python3
"""Opt-in long fuzz run of the element-wise epilogue on an MI355X (not collected by pytest): random chains of 0..4 lazy
tensor operators with random tags, operand formats, scalar / tensor operands, intermediate tensor types and placement
flags, behind random linear-class and tree-class GEMMs, against oracle GEMM + oracle chain.
usage: python tests/extended_fuzz_eltwise.py [cases] [seed]"""
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
from qublas_amd.desc import Ew, Qu, Tags, lower, lower_epilogue  # noqa: E402
from test_gpu_fuzz import rand_qu, rand_tags  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    oracle.lib()
    ran = skipped = fused = 0
    kernels = {}
    for it in range(cases):
        ea = rand_qu(rng, rng.choice([7, 12, 15, 16, 23]))   # (15 / 23 value bits + sign: 16- and 24-bit words, stored centred; 24-bit words fuse the chain on 3 x 3 limbs)
        kind = rng.random()
        if kind < 0.6:
            pf = Qu(2 * ea.intBits + 1, 2 * ea.fracBits, ea.isSigned)
            kw = dict(mul_args=pf, add_args=[Qu(pf.intBits + 12, pf.fracBits, pf.isSigned)])
        else:
            kw = dict(mul_args=rand_tags(rng, ea), add_args=[rand_qu(rng, 14) for _ in range(rng.randint(0, 2))] or None)
        ec = rand_qu(rng, rng.choice([7, 16, 24, 30]))
        M, N, K = rng.randint(1, 200), rng.randint(1, 200), rng.choice([1, 7, 64, 100, 256, 300])
        stages = []
        for _ in range(rng.randint(0, 4)):
            op = rng.choice(["add", "sub", "mul"])
            e = rand_qu(rng, rng.choice([6, 10, 16]))
            stages.append(Ew(op, e, rand_tags(rng, e), x_first=rng.random() < 0.6, scalar=rng.random() < 0.4,
                             into=rand_qu(rng, rng.choice([12, 20, 30])) if rng.random() < 0.5 else None))
        dq = rand_qu(rng, rng.choice([7, 15, 24, 40]))
        try:
            d = lower(ea, ea, ec, M, N, K, transposed_a=rng.random() < 0.5, **kw)
            ep = lower_epilogue(ec, stages, dq)
        except ValueError:
            skipped += 1
            continue
        flags = rng.choice([0, 0, capi.OPT_FUSED_EPILOGUE, capi.OPT_UNFUSED_EPILOGUE, capi.OPT_FORCE_TREE])
        st, info = capi.classify_ep_status(d, ep, flags)
        if st != capi.QG_OK:
            skipped += 1
            continue
        dist = rng.randint(0, 1)
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
        B = oracle.fill(ea, K * N, rng.randint(1, 1 << 30), dist)
        Eo, Eh = [], []
        for s_ in stages:
            h = oracle.fill(s_.e, 1 if s_.scalar else M * N, rng.randint(1, 1 << 30), rng.randint(0, 1))
            Eh.append(h)
            Eo.append(h.astype(np.int64))
        out = np.zeros(M * N, dtype=np.int32 if dq.storage_bits <= 32 else np.int64)
        capi.run_ep(d, ep, out, A, B, Eh, flags=flags)
        exp = oracle.eltwise(ep, ec, oracle.gemm(d, A, B, ec, nthreads=8).astype(np.int64), Eo)
        k = capi.KERNEL_NAMES[info.kernel]
        if not np.array_equal(out.astype(np.int64), exp):
            print(json.dumps({"mismatch": it, "kernel": k, "M": M, "N": N, "K": K, "flags": flags, "a": str(ea), "c": str(ec), "d": str(dq),
                              "kw": str(kw), "stages": str(stages)}), flush=True)
            sys.exit(1)
        kernels[k] = kernels.get(k, 0) + 1
        ran += 1
    print(json.dumps({"chains_run": ran, "skipped_unsupported": skipped, "kernels": kernels, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
