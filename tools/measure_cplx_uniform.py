#!/usr/bin/env python3
"""A/B of the complex tree kernel's "one clamp for the whole loop" form (qg_tree_cplx.hip, MODE 4) against the compact
per-step-record form the same descriptors had before, at BASELINE configuration 5 (2048^3 Qcomplex<int<6,3>, int<6,-3>>
RND::POS_INF + SAT::TCPL, TFComplexMul) and with the reference's default modes.  Run with the diagnostic library:
    QUBLAS_AMD_DIAG=1 python tools/measure_cplx_uniform.py                       # the new form
    QUBLAS_AMD_DIAG=1 QG_NO_PACKED16=1 python tools/measure_cplx_uniform.py        # 32-bit left-justified values instead of packed 16-bit halves
    QUBLAS_AMD_DIAG=1 QG_NO_LEFT_JUSTIFIED=1 python tools/measure_cplx_uniform.py  # v_med3 with the bounds in registers instead of saturating instructions
    QUBLAS_AMD_DIAG=1 QG_NO_UNIFORM_CLAMP=1 python tools/measure_cplx_uniform.py   # the compact per-step-record form
One JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, RND, SAT, TFComplexMul, lower  # noqa: E402

S = 2048
P = lambda i, f: Qu(i, f, True, RND.POS_INF, SAT.TCPL)
C5 = Qcomplex(P(6, 3), P(6, -3))
CD = Qcomplex(Qu(6, 3), Qu(6, -3))
CASES = [("configuration 5 (RND::POS_INF + SAT::TCPL), TFComplexMul", lower(C5, C5, C5, S, S, S, mul_args=TFComplexMul())),
         ("same widths, default modes, TFComplexMul", lower(CD, CD, CD, S, S, S, mul_args=TFComplexMul())),
         ("configuration 5's formats, BasicComplexMul (b d in int<6,-3>: its own mask)", lower(C5, C5, C5, S, S, S, mul_args=BasicComplexMul())),
         ("Qcomplex<int<6,3>, int<6,3>> RND::POS_INF + SAT::TCPL, BasicComplexMul", lower(Qcomplex(P(6, 3), P(6, 3)), Qcomplex(P(6, 3), P(6, 3)), C5, S, S, S, mul_args=BasicComplexMul()))]


def main():
    with capi.Context(0) as ctx:
        for name, d in CASES:
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            plan.time_execute(pC, pA, pB, 2, 5)
            ms = [plan.time_execute(pC, pA, pB, 1, 5) for _ in range(3)]
            print(json.dumps({"case": name, "switches": [k for k in ("QG_NO_UNIFORM_CLAMP", "QG_NO_LEFT_JUSTIFIED", "QG_NO_PACKED16") if os.environ.get(k)], "steps": plan.info.reason.decode().split("steps: ")[-1],
                              "ms": [round(m, 4) for m in ms]}), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()


if __name__ == "__main__":
    main()
