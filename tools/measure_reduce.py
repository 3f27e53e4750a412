#!/usr/bin/env python3
"""Batched Qreduce on resident packed data: rows x len through the Qgemul path (N = 1).
Prints one JSON line per case with the kernel chosen, time, and the bytes-per-second of reading the batch once."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, SAT, TRN, lower_reduce  # noqa: E402

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E43 = Qu(4, 3)
CASES = [
    ("int<8,8> default levels (tree)", E88, 65536, 4096, None),
    ("int<8,8> levels Qu<20,8> (exact: linear class)", E88, 65536, 4096, [Qu(20, 8)]),
    ("int<4,3> levels Qu<16,3> (linear class, int8)", E43, 65536, 4096, [Qu(16, 3)]),
    ("int<8,8> default levels, short rows", E88, 1 << 20, 64, None),
    # the README's Qreduce<list>(m): list = TypeList<int<6,3> SAT::ZERO, int<6,-3>> (readme.md:28-36, :56-60): per-level formats
    ("README level list <int<6,3> SAT::ZERO, int<6,-3>> on int<6,3> elements (tree, per-level formats)", Qu(6, 3, True, TRN.TCPL, SAT.ZERO), 65536, 4096,
     [Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)]),
    ("same, run-time modes forced", Qu(6, 3, True, TRN.TCPL, SAT.ZERO), 65536, 4096, [Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)], capi.OPT_RUNTIME_MODES),
    ("int<8,8> default modes, levels Qu<10,8>, Qu<12,8> (clamping levels)", Qu(8, 8), 65536, 4096, [Qu(10, 8), Qu(12, 8)]),
    ("same, run-time modes forced", Qu(8, 8), 65536, 4096, [Qu(10, 8), Qu(12, 8)], capi.OPT_RUNTIME_MODES),
    ("README level list, short rows", Qu(6, 3, True, TRN.TCPL, SAT.ZERO), 1 << 20, 64, [Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)]),
    ("same, run-time modes forced", Qu(6, 3, True, TRN.TCPL, SAT.ZERO), 1 << 20, 64, [Qu(6, 3, True, TRN.TCPL, SAT.ZERO), Qu(6, -3)], capi.OPT_RUNTIME_MODES),
]
with capi.Context() as ctx:
    for case in CASES:
        name, e, rows, n, levels = case[:5]
        flags = case[5] if len(case) > 5 else 0
        d = lower_reduce(e, rows, n, levels)
        plan = capi.Plan(ctx, d, flags)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        plan.fill(capi.OPERAND_A, 1, 0, pA)
        plan.fill(capi.OPERAND_B, 2, 0, pB)   # values irrelevant for timing
        ms = min(plan.time_execute(pC, pA, pB, 3, 10) for _ in range(2))
        print(json.dumps({"case": name, "rows": rows, "len": n, "kernel": capi.KERNEL_NAMES[plan.info.kernel], "steps": plan.info.reason.decode().split("steps: ")[-1],
                          "kernel_ms": ms, "packed_A_bytes": int(pb[0]), "GBps_of_packed_A": pb[0] / (ms * 1e-3) / 1e9}), flush=True)
        for p in (pA, pB, pC):
            ctx.free(p)
        plan.close()
