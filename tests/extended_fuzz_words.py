#!/usr/bin/env python3
"""Opt-in fuzz of the WORD forms of the tree class on an MI355X (not collected by pytest):
one signed format of 24 ... 32 bits for the product and every level — SAT::TCPL (saturating words, justified words) or WRP::TCPL
(wrapping words) — elements of 8 ... 32 bits, truncating and rounding products (the other roundings too: those descriptors must
take the neighbouring kernels and still be right), C of any width up to 32 bits, any K, square shapes, single rows / columns, the
one-column kernel (Qreduce lowering and GEMV).  Each case: GPU against the oracle, and — where a word form runs — against the
64-bit kernels the same descriptor takes with QG_OPT_RUNTIME_MODES.
usage: python tests/extended_fuzz_words.py [cases] [seed] [dry]"""
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
from qublas_amd.desc import ONE, Qu, RND, SAT, TRN, WRP, lower, lower_reduce, reduce_result_type  # noqa: E402

QM_FAST = [TRN.TCPL, TRN.TCPL, RND.POS_INF, RND.NEG_INF]
QM_ALL = QM_FAST + [RND.ZERO, RND.INF, RND.CONV, TRN.SMGN]


def split(rng, bits):
    i = rng.randint(0, bits)
    return i, bits - i


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1357)
    dry = len(sys.argv) > 3 and sys.argv[3] == "dry"      # (no GPU: only which kernels / forms the planner picks)
    oracle.lib()
    ran = skipped = 0
    forms = {}
    for it in range(cases):
        wbits = rng.choice([31, 31, 31, 31, 30, 29, 27, 25, 23, 23])                 # value bits of the word (31 + sign: a 32-bit word)
        wi, wf = split(rng, wbits)
        of = WRP.TCPL if (wbits == 31 and rng.random() < 0.3) else SAT.TCPL
        qm = rng.choice(QM_FAST if rng.random() < 0.85 else QM_ALL)
        word = Qu(wi, wf, True, qm, of)
        level = Qu(wi, wf, True, rng.choice(QM_ALL), of)
        # elements: the word itself, or other widths whose fraction bits sum to at least the word's (a right shift) most of the time
        if rng.random() < 0.5:
            ea = eb = Qu(wi, wf, True, rng.choice(QM_ALL), rng.choice([SAT.TCPL, SAT.SMGN, WRP.TCPL]))
        else:
            ba, bb = rng.choice([8, 12, 16, 24, 31]), rng.choice([7, 15, 16, 24, 31])
            fa = min(ba, max(0, wf - rng.randint(0, 6)))
            fb = rng.randint(0, bb)
            ea = Qu(ba - fa, fa, True, TRN.TCPL, SAT.TCPL)
            eb = Qu(bb - fb, fb, rng.random() < 0.85, TRN.TCPL, SAT.TCPL)
        ci, cf = split(rng, rng.choice([7, 12, 16, 24, 31]))
        ec = word if rng.random() < 0.5 else Qu(ci, cf, True, rng.choice(QM_ALL), rng.choice([SAT.TCPL, SAT.SMGN, SAT.ZERO, WRP.TCPL]))
        M, N = rng.randint(1, 100), rng.randint(1, 100)
        K = rng.choice([1, 2, 17, 32, 33, 64, 100, 250, 256, 512, 1000, 2048, 4096])
        reduce_form = False
        if rng.random() < 0.3:
            N, K = 1, rng.choice([64, 256, 300, 1024, 4096, 5000])
            M = rng.randint(1, 400)
            reduce_form = rng.random() < 0.5
        try:
            if reduce_form:
                ea = Qu(wi, wf, True, qm, of)
                eb, ec = ONE, reduce_result_type(ea, [], K)
                d = lower_reduce(ea, M, K)
            else:
                d = lower(ea, eb, ec, M, N, K, mul_args=word, add_args=[level], transposed_a=rng.random() < 0.5)
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
        if dry:
            ran += 1
            continue
        dist = rng.randint(0, 2)
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist % 2)
        B = np.ones(K, dtype=np.int32) if reduce_form else oracle.fill(eb, K * N, rng.randint(1, 1 << 30), dist % 2)
        if dist == 2:                     # small values: nothing saturates
            A = (A >> rng.randint(4, 12)).astype(A.dtype)
            if not reduce_form:
                B = (B >> rng.randint(4, 12)).astype(B.dtype)
        elif rng.random() < 0.5:          # the extremes are present
            A[: min(2, A.size)] = ea.raw_min
            A[-1] = ea.raw_max
            if not reduce_form:
                B[: min(2, B.size)] = eb.raw_min
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
                              "word": str(word), "level": str(level), "reduce": reduce_form}), flush=True)
            sys.exit(1)
        ran += 1
    print(json.dumps({"word_form_cases": ran, "skipped_unsupported": skipped, "kernels_and_step_forms": forms, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
