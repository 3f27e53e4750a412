#!/usr/bin/env python3
"""Two-digit Karatsuba kernel (operands of 9..12 value+sign bits, here int<6,5>) at 4096^3: QG_KARA32=1 keeps it on the 32x32x32
MFMA shape, QG_NO_KARA=1 runs the four-product 2x2 limb kernel instead.  Needs an MI355X."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

E = Qu(6, 5)
S = 4096
with capi.Context() as ctx:
    d = lower(E, E, Qu(25, 10), S, S, S, mul_args=Tags(13, 10), add_args=[Qu(25, 10)])
    plan = capi.Plan(ctx, d)
    pb = plan.info.packed_bytes
    pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
    plan.fill(capi.OPERAND_A, 1, 0, pA)
    plan.fill(capi.OPERAND_B, 2, 0, pB)
    plan.time_execute(pC, pA, pB, 100, 100)
    ms = min(plan.time_execute(pC, pA, pB, 20, 100) for _ in range(3))
    print(json.dumps({"workload": "4096^3 int<6,5>", "kernel": capi.KERNEL_NAMES[plan.info.kernel], "limbs": list(plan.info.limbs), "kernel_ms": ms,
                      "T_op_per_s": 2.0 * S ** 3 / (ms * 1e-3) / 1e12, "env": {k: v for k, v in os.environ.items() if k.startswith("QG_")}}), flush=True)
    plan.close()
