#!/usr/bin/env python3
"""Centred operands (x - c in balanced int8 limbs, QPackedGeom::offs) against plain balanced limbs (QG_OPT_BALANCED_LIMBS) at full
size: kernel time of the linear class on 8-, 16-, 24- and 32-bit formats.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

U8, Q78, Q1112, Q1516 = Qu(8, 0, False), Qu(7, 8), Qu(11, 12), Qu(15, 16)
CASES = [
    ("uint8 x uint8 4096^3 (exact sums)", lower(U8, U8, Qu(28, 0, False), 4096, 4096, 4096, mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)])),
    ("Q7.8 x Q7.8 (16-bit words) 4096^3", lower(Q78, Q78, Qu(23, 8), 4096, 4096, 4096, mul_args=Tags(15, 16), add_args=[Qu(27, 16)])),
    ("Q11.12 x Q11.12 (24-bit words) 4096^3", lower(Q1112, Q1112, Qu(35, 12), 4096, 4096, 4096, mul_args=Tags(23, 24), add_args=[Qu(35, 24)])),
    ("Q15.16 x Q15.16 (32-bit words), exact 76-bit sums, 2048^3", lower(Q1516, Q1516, Qu(43, 32), 2048, 2048, 2048, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])),
    ("Q15.16 x Q15.16, exact sums, 4096^3", lower(Q1516, Q1516, Qu(43, 32), 4096, 4096, 4096, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])),
]


def main():
    with capi.Context(0) as ctx:
        for name, d in CASES:
            row = {"case": name}
            for tag, fl in (("centred", 0), ("balanced", capi.OPT_BALANCED_LIMBS)):
                plan = capi.Plan(ctx, d, fl)
                pb = plan.info.packed_bytes
                pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
                plan.fill(capi.OPERAND_A, 1, 0, pA)
                plan.fill(capi.OPERAND_B, 2, 0, pB)
                plan.time_execute(pC, pA, pB, 5, 20)
                ms = plan.time_execute(pC, pA, pB, 2, 30)
                row[tag] = {"limbs": list(plan.info.limbs), "ms": round(ms, 4), "Top_s": round(plan.info.ops / ms / 1e9, 1),
                            "reason": plan.info.reason.decode()}
                for p in (pA, pB, pC):
                    ctx.free(p)
                plan.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
