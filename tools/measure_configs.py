#!/usr/bin/env python3
"""Kernel time of every BASELINE.json configuration (and the class variants of SURVEY.md §8-d) with
resident, device-generated operands; HIP events on the engine's stream.  Writes one JSON document."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, Tags, TFComplexMul, BasicComplexMul, lower  # noqa: E402

E43 = Qu(4, 3)
E88Z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
R63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
I63N = Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
C5 = Qcomplex(R63, I63N)
PEAK = 5.0e15

CASES = [
    ("c1 4x4x4 int<8,8> tree (README shapes)", lower(E88Z, E88Z, E88Z, 4, 4, 4, mul_args=E88Z, add_args=[E88Z]), 200),
    ("c2 1024^3 int<4,3> linear -> MFMA_I32_I8", lower(E43, E43, E43, 1024, 1024, 1024, mul_args=Tags(9, 6), add_args=[Qu(19, 6)]), 200),
    ("c2 1024^3 int<4,3> default tags -> tree", lower(E43, E43, E43, 1024, 1024, 1024), 20),
    ("c3 4096^3 int<8,8> linear -> 3x3 limb MFMA", lower(E88Z, E88Z, Qu(23, 8), 4096, 4096, 4096, mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 50),
    ("c3 4096^3 int<8,8> default tags -> tree (i32 VALU)", lower(E88Z, E88Z, E88Z, 4096, 4096, 4096), 5),
    ("c4 16384x16384x4096 int<4,3> linear -> MFMA_I32_I8 (one GPU, all rows)", lower(E43, E43, E43, 16384, 16384, 4096, mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 20),
    ("c4 shard 2048x16384x4096 (what one of 8 GPUs computes)", lower(E43, E43, E43, 2048, 16384, 4096, mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 50),
    ("c5 2048^2 (K=2048) Qcomplex<int<6,3>,int<6,-3>> TFComplexMul RND+SAT -> tree (complex)", lower(C5, C5, C5, 2048, 2048, 2048, mul_args=TFComplexMul()), 2),
    ("c5 same, BasicComplexMul", lower(C5, C5, C5, 2048, 2048, 2048, mul_args=BasicComplexMul()), 2),
    ("c5 linear-class variant: BasicComplexMul with exact sub-op types, levels Qcomplex<Qu<30,6>,Qu<30,0>> -> 4 real limb GEMMs on MFMA",
     lower(C5, C5, C5, 2048, 2048, 2048, mul_args=BasicComplexMul(acT=Qu(14, 6), bdT=Qu(14, -6), adT=Qu(14, 0), bcT=Qu(14, 0), acbdT=Qu(15, 6), adbcT=Qu(15, 0)),
           add_args=[Qcomplex(Qu(30, 6), Qu(30, 0))]), 20),
]


def main():
    only = os.environ.get("ONLY")   # substring filter on the configuration name
    global CASES
    if only:
        CASES = [c for c in CASES if only in c[0]]
    out = []
    with capi.Context(0) as ctx:
        for name, d, iters in CASES:
            plan = capi.Plan(ctx, d)
            info = plan.info
            pb = info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            ms = plan.time_execute(pC, pA, pB, 2, iters)
            rec = {"config": name, "M": d.M, "N": d.N, "K": d.K, "kernel": capi.KERNEL_NAMES[info.kernel],
                   "class": "linear" if info.cls == 1 else "tree", "limbs": [info.limbs[0], info.limbs[1]],
                   "kernel_ms": ms, "ops": info.ops, "ops_per_s": info.ops / (ms * 1e-3),
                   "pct_int8_peak": 100.0 * info.ops / (ms * 1e-3) / PEAK, "packed_bytes": list(pb)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()
    if only:
        return
    # the drop-in call with HOST buffers (qgemul_run): H2D + pack + GEMM + unpack + D2H + alloc/free, wall time
    import time
    import numpy as np
    for name, ea, ec, kw, S in (("c3 linear, host buffers through qgemul_run", E88Z, Qu(23, 8), dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 4096),
                                ("c2 linear 1024^3, host buffers through qgemul_run", E43, E43, dict(mul_args=Tags(9, 6), add_args=[Qu(19, 6)]), 1024)):
        d = lower(ea, ea, ec, S, S, S, **kw)
        rng = np.random.default_rng(1)
        A = rng.integers(ea.raw_min, ea.raw_max + 1, S * S, dtype=np.int32)
        B = rng.integers(ea.raw_min, ea.raw_max + 1, S * S, dtype=np.int32)
        Cc = np.zeros(S * S, dtype=np.int32)
        capi.run(d, Cc, A, B)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            capi.run(d, Cc, A, B)
        dt = (time.perf_counter() - t0) / n
        rec = {"config": name, "M": S, "N": S, "K": S, "wall_ms": dt * 1e3, "ops_per_s": 2.0 * S ** 3 / dt,
               "host_bytes_moved": int(A.nbytes + B.nbytes + Cc.nbytes)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
