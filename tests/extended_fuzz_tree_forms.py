#!/usr/bin/env python3
"""Opt-in fuzz of the real 32-bit tree kernel's step forms on an MI355X (not collected by pytest): tree-class descriptors
whose roundings are "add a constant, shift right" (TRN::TCPL, RND::POS_INF, RND::NEG_INF) and whose overflows are a clamp
(SAT::TCPL, SAT::SMGN), a range test (SAT::ZERO) or a wrap (WRP::TCPL), with product tags and 0..3 level types of other widths / fracBits, any K >= 17, split and direct
products — so that the planner picks the compact per-level form (fast_mode 3) or one of the one-format forms.  Each case:
GPU against the oracle, and against the same plan with run-time modes (QG_OPT_RUNTIME_MODES).
usage: python tests/extended_fuzz_tree_forms.py [cases] [seed]"""
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
from qublas_amd.desc import ONE, Qu, RND, SAT, TRN, WRP, Tags, lower, lower_reduce, reduce_result_type  # noqa: E402

QM = [TRN.TCPL, TRN.TCPL, RND.POS_INF, RND.NEG_INF, RND.ZERO, RND.INF, RND.CONV, TRN.SMGN]   # (the last four: rounding kinds of the unbiased form)
OM = [SAT.TCPL, SAT.TCPL, SAT.SMGN, SAT.ZERO, SAT.ZERO, WRP.TCPL]


def rq(rng, bits, signed=None):
    i = rng.randint(0, bits)
    f = bits - i
    if rng.random() < 0.2:
        sh = rng.randint(1, 3)
        i, f = i + sh, f - sh
    return Qu(i, f, (rng.random() < 0.85) if signed is None else signed, rng.choice(QM), rng.choice(OM))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2468)
    oracle.lib()
    ran = skipped = 0
    forms = {}
    for it in range(cases):
        ea = rq(rng, rng.choice([4, 7, 8, 12, 16]))
        eb = ea if rng.random() < 0.5 else rq(rng, rng.choice([4, 7, 8, 12, 16]))
        r = rng.random()
        mul = None if r < 0.35 else rq(rng, rng.choice([8, 12, 16, 20])) if r < 0.8 else Tags(intBits=ea.intBits + rng.randint(0, 5), fracBits=ea.fracBits + rng.randint(-3, 3))
        levels = [rq(rng, rng.choice([10, 14, 18, 22, 26])) for _ in range(rng.choice([0, 1, 1, 2, 3]))]
        if rng.random() < 0.2:   # one signed SAT::TCPL format for the product and every level: the left-justified saturating form
            u = rq(rng, rng.choice([3, 5, 7, 9, 11, 12, 14, 15, 15, 16, 20, 23, 23, 25, 27, 30, 31, 31]), True)   # (value bits; 15 + sign: 16-bit words; 31 + sign: 32-bit words)
            if u.intBits + u.fracBits >= 23 and rng.random() < 0.6:     # (32-bit words and justified words: elements of up to 32 bits)
                ea = rq(rng, rng.choice([12, 16, 24, 31]), True)
                eb = ea if rng.random() < 0.5 else rq(rng, rng.choice([7, 16, 31]), True)
            if rng.random() < 0.5:   # narrow operands too: the packed 16-bit form needs format bits + the product's shift <= 16
                ea = rq(rng, rng.choice([3, 4, 5, 7]))
                eb = ea if rng.random() < 0.5 else rq(rng, rng.choice([3, 4, 5, 7]))
            sg = rng.random() < 0.7                 # (unsigned: the uint32 / uint16 counterparts, when the operands are unsigned too)
            u = Qu(u.intBits, u.fracBits, sg, rng.choice([TRN.TCPL, RND.POS_INF, RND.NEG_INF]), SAT.TCPL)
            mul, levels = u, [Qu(u.intBits, u.fracBits, sg, rng.choice(QM), SAT.TCPL) for _ in range(rng.choice([0, 1, 2]))]
            if not sg:
                ea = Qu(ea.intBits, ea.fracBits, False, ea.QuMode, ea.OfMode)
                eb = Qu(eb.intBits, eb.fracBits, False, eb.QuMode, eb.OfMode)
        ec = rq(rng, rng.choice([7, 12, 16, 24]))
        M, N = rng.randint(1, 130), rng.randint(1, 130)
        K = rng.choice([17, 32, 33, 64, 100, 128, 250, 256, 512, 1000, 2048])
        reduce_form = False
        if rng.random() < 0.3:           # one output column: the batched Qreduce / GEMV kernel (K a power of two >= 16)
            N, K = 1, rng.choice([16, 32, 64, 128, 256, 1024, 4096])
            M = rng.randint(1, 700)
            reduce_form = rng.random() < 0.5
        try:
            if reduce_form:              # the Qreduce lowering: B is the 0/1 vector, the product is the element itself
                eb, ec = ONE, reduce_result_type(ea, levels, K)
                d = lower_reduce(ea, M, K, levels or None)
            else:
                d = lower(ea, eb, ec, M, N, K, mul_args=mul, add_args=levels or None, transposed_a=rng.random() < 0.5)
        except ValueError:
            skipped += 1
            continue
        st, info = capi.classify_status(d)
        if st != capi.QG_OK:
            skipped += 1
            continue
        k = capi.KERNEL_NAMES[info.kernel]
        form = (("one column: " if k == "gemv_i32" else "") + info.reason.decode().split("steps: ")[-1]) if k in ("tree_i32", "gemv_i32") else k
        forms[form] = forms.get(form, 0) + 1
        dist = rng.randint(0, 1)
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
        B = np.ones(K, dtype=np.int32) if reduce_form else oracle.fill(eb, K * N, rng.randint(1, 1 << 30), dist)
        out = np.zeros(M * N, dtype=oracle.host_dtype(ec))
        capi.run(d, out, A, B)
        exp = oracle.gemm(d, A, B, ec, nthreads=8)
        ok = np.array_equal(out, exp)
        if ok and k in ("tree_i32", "gemv_i32"):
            rt = np.zeros(M * N, dtype=oracle.host_dtype(ec))
            capi.run(d, rt, A, B, flags=capi.OPT_RUNTIME_MODES)
            ok = np.array_equal(rt, exp)
        if not ok:
            print(json.dumps({"mismatch": it, "kernel": k, "form": form, "M": M, "N": N, "K": K, "a": str(ea), "b": str(eb), "c": str(ec),
                              "mul": str(mul), "levels": str(levels)}), flush=True)
            sys.exit(1)
        ran += 1
    print(json.dumps({"tree_class_cases": ran, "skipped_unsupported": skipped, "kernels_and_step_forms": forms, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
