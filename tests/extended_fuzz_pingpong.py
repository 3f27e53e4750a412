#!/usr/bin/env python3
"""Opt-in long fuzz run of the two-group MFMA kernels on an MI355X (not collected by pytest).  k_mfma_pp / k_mfma_ppl are chosen
only for problems with at least a tile per CU, far beyond what the CPU oracle evaluates in fuzz time — so every case runs the
WHOLE matrix through the two-group kernel and through the lock-step kernel it replaced (QG_OPT_LOCKSTEP_TILES; different
tiles, packing and pipeline; itself fuzzed against the oracle by extended_fuzz_shapes.py) and compares every byte, then a
random block against the oracle; every third case also goes through qgemul_execute_host_c (the epilogue storing the reference
layout) with a random leading dimension.  Random: shapes around the tile counts, K from one k-tile up, operand formats (1 or 3
limbs), A orientation irrelevant here (resident packed operands), full-range and half-range data (plane masks), every C
container and QuMode x OfMode.
usage: python tests/extended_fuzz_pingpong.py [cases] [seed]"""
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


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    oracle.lib()
    ran = {"pp": 0, "ppl": 0, "host_c": 0, "direct_store": 0}
    with capi.Context() as ctx:
        for it in range(cases):
            limb = rng.random() < 0.5
            if limb:       # 3 x 3 limbs: 17..23 storage bits, 128^2 tiles, >= 256 tiles
                w = rng.choice([16, 16, 17, 20, 22, 14, 13])   # 13, 14: two limbs (k_mfma_ppl22)
                M, N = rng.randint(1921, 2400), rng.randint(1921, 2400)
                K = rng.choice([1, 63, 64, 65, 128, 129, 320, 1000, 1500, 4096])
            else:          # single limb: <= 8 storage bits, 256^2 tiles, >= 256 tiles
                w = rng.choice([7, 7, 6, 4])
                M, N = rng.randint(3841, 4700), rng.randint(3841, 4700)
                K = rng.choice([1, 127, 128, 129, 256, 257, 640, 1000, 2048])
            fa, fb = rng.randint(0, min(6, w)), rng.randint(0, min(6, w))
            ea, eb = Qu(w - fa, fa, True), Qu(w - fb, fb, rng.random() < 0.9)
            pf = Qu(ea.intBits + eb.intBits + 1, fa + fb, True)
            acc = Qu(pf.intBits + 13, pf.fracBits, True)
            ec = Qu(rng.randint(3, 34 if limb else 30), rng.randint(-2, 10), rng.random() < 0.8, rng.randint(0, 6), rng.choice([0, 1, 2, 3]))
            try:
                d = lower(ea, eb, ec, M, N, K, mul_args=pf, add_args=[acc])
            except ValueError:
                continue
            st, info = capi.classify_status(d, 0)
            if st != capi.QG_OK or capi.KERNEL_NAMES[info.kernel] not in ("mfma_i8", "mfma_i8_limb"):
                continue
            dist = rng.choice([0, 0, 1])
            sa, sb = rng.randint(1, 1 << 30), rng.randint(1, 1 << 30)
            outs = []
            eb_c = info.host_elem_bytes[2]
            ldc = M + rng.choice([0, 0, 4, 5])
            hostc = None
            for flags in (0, capi.OPT_LOCKSTEP_TILES):
                plan = capi.Plan(ctx, d, flags)
                pb = plan.info.packed_bytes
                pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
                dC = ctx.alloc(ldc * N * eb_c)
                ctx.h2d(dC, np.full(ldc * N * eb_c, 0x3c, np.uint8))
                plan.fill(capi.OPERAND_A, sa, dist, pA)
                plan.fill(capi.OPERAND_B, sb, dist, pB)
                plan.execute(pC, pA, pB)
                plan.unpack_c(pC, dC, ldc)
                o = np.zeros(ldc * N * eb_c, np.uint8)
                ctx.d2h(o, dC)
                outs.append(o)
                if flags == 0 and it % 3 == 0:
                    ctx.h2d(dC, np.full(ldc * N * eb_c, 0x3c, np.uint8))
                    plan.execute_host_c(dC, pA, pB, ldc)
                    hostc = np.zeros(ldc * N * eb_c, np.uint8)
                    ctx.d2h(hostc, dC)
                    ran["host_c"] += 1
                    ran["direct_store"] += int(plan.stores_host_c)
                for p in (pA, pB, pC, dC):
                    ctx.free(p)
                plan.close()
            rec = dict(it=it, limb=limb, M=M, N=N, K=K, ea=ea.as_tuple(), eb=eb.as_tuple(), ec=ec.as_tuple(), dist=dist, ldc=ldc)
            if not np.array_equal(outs[0], outs[1]):
                print(json.dumps({"mismatch": "two-group vs lock-step", **rec}), flush=True)
                sys.exit(1)
            if hostc is not None and not np.array_equal(hostc, outs[0]):
                print(json.dumps({"mismatch": "execute_host_c vs execute + unpack", **rec}), flush=True)
                sys.exit(1)
            # a random block against the oracle
            r0, c0 = rng.randint(0, M - 4), rng.randint(0, N - 64)
            A = oracle.fill(ea, M * K, sa, dist)
            B = oracle.fill(eb, K * N, sb, dist)
            cdt = oracle.host_dtype(ec)
            exp = np.zeros(M * N, dtype=cdt)
            oracle.gemm(d, A, B, ec, rows=(r0, r0 + 4), cols=(c0, c0 + 64), nthreads=16, out=exp)
            got = outs[0].view(cdt).reshape(N, ldc)[c0:c0 + 64, r0:r0 + 4]
            if not np.array_equal(got, exp.reshape(N, M)[c0:c0 + 64, r0:r0 + 4]):
                print(json.dumps({"mismatch": "oracle block", "r0": r0, "c0": c0, **rec}), flush=True)
                sys.exit(1)
            ran["ppl" if limb else "pp"] += 1
            if it % 20 == 0:
                print(json.dumps({"progress": it, **ran}), flush=True)
    print(json.dumps({"cases": cases, "mismatches": 0, **ran}))


if __name__ == "__main__":
    main()
