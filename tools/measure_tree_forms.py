#!/usr/bin/env python3
"""Kernel time of the 32-bit tree kernels' step forms at 2048^3 (one format / compact per-level records / run-time modes),
real and complex; one JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, WRP, TFComplexMul, lower  # noqa: E402

E = Qu(8, 8)
EZ = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
S = 2048
CD = Qcomplex(Qu(6, 3), Qu(6, -3))     # configuration 5's widths with the reference's default modes
CZ = Qcomplex(Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3, True, TRN.TCPL, SAT.ZERO))
CC = Qcomplex(Qu(6, 3, True, RND.CONV), Qu(6, -3, True, RND.CONV))
CW = Qcomplex(Qu(6, 3, True, TRN.TCPL, WRP.TCPL), Qu(6, -3, True, TRN.TCPL, WRP.TCPL))
CI = Qcomplex(Qu(6, 3, True, RND.INF), Qu(6, -3, True, RND.INF))
CASES = [
    ("int<8,8> default tags (one format, SAT::TCPL: left-justified; QG_NO_LEFT_JUSTIFIED=1 in the diagnostic library: the v_med3 form)", lower(E, E, E, S, S, S), 0),
    ("same, run-time modes forced", lower(E, E, E, S, S, S), capi.OPT_RUNTIME_MODES),
    ("int<4,3> default tags at 1024^3 (configuration 2 as literally configured: packed 16-bit; QG_NO_PACKED16=1: left-justified 32-bit; QG_NO_LEFT_JUSTIFIED=1: v_med3)",
     lower(Qu(4, 3), Qu(4, 3), Qu(4, 3), 1024, 1024, 1024), 0),
    ("int<7,8> (16-bit words) default tags at 2048^3 (32-bit justified products, packed 16-bit nodes; QG_NO_PACKED16=1: left-justified 32-bit)", lower(Qu(7, 8), Qu(7, 8), Qu(7, 8), S, S, S), 0),
    ("int<4,3> default tags at 4096^3", lower(Qu(4, 3), Qu(4, 3), Qu(4, 3), 4096, 4096, 4096), 0),
    ("int<8,8>, level type Qu<12,8> (QgemulAddArgs)", lower(E, E, Qu(12, 8), S, S, S, add_args=[Qu(12, 8)]), 0),
    ("same, run-time modes forced", lower(E, E, Qu(12, 8), S, S, S, add_args=[Qu(12, 8)]), capi.OPT_RUNTIME_MODES),
    ("int<8,8> SAT::ZERO, level types Qu<10,8>, Qu<12,6> SAT::ZERO", lower(EZ, EZ, Qu(12, 6, True, TRN.TCPL, SAT.ZERO), S, S, S,
                                                                          add_args=[Qu(10, 8, True, TRN.TCPL, SAT.ZERO), Qu(12, 6, True, TRN.TCPL, SAT.ZERO)]), 0),
    ("product Qu<10,6> RND::POS_INF, levels Qu<12,6>, Qu<16,4>", lower(E, E, Qu(16, 4), S, S, S, mul_args=Qu(10, 6, True, RND.POS_INF, SAT.TCPL),
                                                                      add_args=[Qu(12, 6), Qu(16, 4)]), 0),
    # the README's own Qgemul call (readme.md:28-36, :84-87): type1 = int<6,3> SAT::ZERO, type2 = int<6,-3>, AddArgs<TypeList<type1, type2>>, MulArgs<type1>
    ("README example types at 2048^3 (product and level 0 SAT::ZERO, levels >= 1 int<6,-3> default modes)",
     lower(Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, 3, True, TRN.TCPL, SAT.ZERO), S, S, S,
           mul_args=Qu(6, 3, True, TRN.TCPL, SAT.ZERO), add_args=[Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)], transposed_a=True), 0),
    ("same, run-time modes forced",
     lower(Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, 3, True, TRN.TCPL, SAT.ZERO), S, S, S,
           mul_args=Qu(6, 3, True, TRN.TCPL, SAT.ZERO), add_args=[Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)], transposed_a=True), capi.OPT_RUNTIME_MODES),
    ("int<8,8> RND::CONV, default tags (a rounding that looks at the value: rounding kind of the unbiased form)",
     lower(Qu(8, 8, True, RND.CONV), Qu(8, 8, True, RND.CONV), Qu(8, 8, True, RND.CONV), S, S, S), 0),
    ("same, run-time modes forced", lower(Qu(8, 8, True, RND.CONV), Qu(8, 8, True, RND.CONV), Qu(8, 8, True, RND.CONV), S, S, S), capi.OPT_RUNTIME_MODES),
    ("complex int<6,3>/int<6,-3> DEFAULT modes, TFComplexMul", lower(CD, CD, CD, S, S, S, mul_args=TFComplexMul()), 0),
    ("same, run-time modes forced", lower(CD, CD, CD, S, S, S, mul_args=TFComplexMul()), capi.OPT_RUNTIME_MODES),
    # (the README's own Qcomplex<type1, type2> mixes SAT::ZERO and SAT::TCPL parts, which merge to the default modes: plain compact form)
    ("complex int<6,3>/int<6,-3> SAT::ZERO (an overflow kind of the compact form), TFComplexMul", lower(CZ, CZ, CZ, S, S, S, mul_args=TFComplexMul()), 0),
    ("same, run-time modes forced", lower(CZ, CZ, CZ, S, S, S, mul_args=TFComplexMul()), capi.OPT_RUNTIME_MODES),
    ("complex int<6,3>/int<6,-3> RND::CONV, TFComplexMul", lower(CC, CC, CC, S, S, S, mul_args=TFComplexMul()), 0),
    ("complex int<6,3>/int<6,-3> WRP::TCPL, TFComplexMul", lower(CW, CW, CW, S, S, S, mul_args=TFComplexMul()), 0),
    ("complex int<6,3>/int<6,-3> RND::INF, TFComplexMul", lower(CI, CI, CI, S, S, S, mul_args=TFComplexMul()), 0),
]


def main():
    with capi.Context(0) as ctx:
        for name, d, fl in CASES:
            plan = capi.Plan(ctx, d, fl)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            ms = plan.time_execute(pC, pA, pB, 3, 10)
            print(json.dumps({"case": name, "kernel": capi.KERNEL_NAMES[plan.info.kernel], "steps": plan.info.reason.decode().split("steps: ")[-1], "ms": ms}), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()


if __name__ == "__main__":
    main()
