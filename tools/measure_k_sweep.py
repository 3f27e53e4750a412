#!/usr/bin/env python3
"""Fixed cost per launch of the 3x3 limb kernel: 4096 x 4096 x K for K = 1024 ... 8192, time = a + b*K (a: launch, pipeline fill and
the epilogue of the 4 tile rounds, none of it overlapped with MFMA work of another workgroup: one workgroup per CU).  Needs an MI355X."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, SAT, TRN, Tags, lower  # noqa: E402

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
S = 4096
pts = []
with capi.Context() as ctx:
    for K in (1024, 2048, 4096, 8192):
        d = lower(E88, E88, Qu(23, 8), S, S, K, mul_args=Tags(17, 16), add_args=[Qu(30, 16)])
        plan = capi.Plan(ctx, d)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        plan.fill(capi.OPERAND_A, 1, 0, pA)
        plan.fill(capi.OPERAND_B, 2, 0, pB)
        plan.time_execute(pC, pA, pB, 100, 100)
        ms = min(plan.time_execute(pC, pA, pB, 20, 100) for _ in range(3))
        pts.append((K, ms))
        print(json.dumps({"K": K, "kernel": capi.KERNEL_NAMES[plan.info.kernel], "limbs": list(plan.info.limbs), "kernel_ms": ms}), flush=True)
        for p in (pA, pB, pC):
            ctx.free(p)
        plan.close()
k = np.array([p[0] for p in pts], dtype=float)
t = np.array([p[1] for p in pts])
b, a = np.polyfit(k, t, 1)
print(json.dumps({"fit": "ms = a + b*K", "a_ms": a, "b_ms_per_1024": b * 1024, "fixed_share_at_K4096": a / (a + b * 4096)}))
