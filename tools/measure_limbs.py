#!/usr/bin/env python3
"""Kernel time of every limb combination of the linear class at 4096^3 (1 limb: int<4,3>, 2: int<7,7>, 3: int<8,8>),
full-range operands.  Environment switches (QG_LIMB32, QG_LIMB16_ALL) select the MFMA shape.  Needs an MI355X."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

FMT = {1: Qu(4, 3), 2: Qu(7, 7), 3: Qu(8, 8)}
S = 4096
with capi.Context() as ctx:
    for la in (1, 2, 3):
        for lb in (1, 2, 3):
            ea, eb = FMT[la], FMT[lb]
            I, F = ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits
            d = lower(ea, eb, Qu(I + 12, F), S, S, S, mul_args=Tags(I, F), add_args=[Qu(I + 12, F)])
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            plan.time_execute(pC, pA, pB, 50, 50)
            ms = min(plan.time_execute(pC, pA, pB, 10, 50) for _ in range(3))
            print(json.dumps({"limbs": [la, lb], "kernel": capi.KERNEL_NAMES[plan.info.kernel], "info_limbs": list(plan.info.limbs), "kernel_ms": ms,
                              "env": {k: v for k, v in os.environ.items() if k.startswith("QG_")}}), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()
