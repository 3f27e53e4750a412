#!/usr/bin/env python3
"""Plane-mask dispatch of the limb MFMA kernel: 4096^3 int<8,8> with full-range operands (all 9 limb products) and with
operands that leave the upper limb planes empty.  Needs an MI355X."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, SAT, TRN, Tags, lower  # noqa: E402

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
S = 4096
d = lower(E88, E88, Qu(23, 8), S, S, S, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
with capi.Context() as ctx:
    plan = capi.Plan(ctx, d)
    pb = plan.info.packed_bytes
    pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
    rng = np.random.default_rng(3)
    host = ctx.alloc(S * S * 4)
    cases = (("full range (uniform over 17 bits)", None), ("|x| <= 32639: third limb plane empty", 32639), ("|x| <= 127: one limb plane", 127))
    if "QG_NO_PLANE_MASK" in os.environ:
        cases = cases[:1]   # without the partner kernel only full-range operands are computed correctly
    for label, lim in cases:
        if lim is None:
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
        else:
            for op, buf in ((capi.OPERAND_A, pA), (capi.OPERAND_B, pB)):
                x = rng.integers(-lim, lim + 1, S * S, dtype=np.int64).astype(np.int32)
                ctx.h2d(host, x)
                plan.pack(op, host, buf)
        plan.time_execute(pC, pA, pB, 100, 100)
        ms = min(plan.time_execute(pC, pA, pB, 20, 100) for _ in range(3))
        print(json.dumps({"partner_kernel": "QG_NO_PLANE_MASK" not in os.environ, "operands": label, "kernel_pair_ms": ms, "T_op_per_s": 2.0 * S ** 3 / (ms * 1e-3) / 1e12}), flush=True)
    plan.close()
