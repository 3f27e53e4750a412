#!/usr/bin/env python3
"""A/B of two builds of the engine on the same box: kernel time of a few workloads with library A and library B in
alternating child processes (one library per process), several rounds.  Used to price a change that is compiled in
unconditionally (no switch to flip): build the previous commit's sources into qublas_amd/build/ab/libqugemm_prev.so and run
    python tools/ab_libs.py qublas_amd/build/ab/libqugemm_prev.so qublas_amd/libqugemm.so
Prints one JSON line per (workload, library) with the per-round medians."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path):
    from qublas_amd import capi
    capi.LIB_PATH = os.path.abspath(path)
    from qublas_amd.desc import Qu, SAT, TRN, Tags, lower
    E43, E88Z = Qu(4, 3), Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    cases = [
        ("c4L 16384^2x4096 int<4,3> -> 1-byte C", lower(E43, E43, E43, 16384, 16384, 4096, mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 20),
        ("c3L 4096^3 int<8,8>", lower(E88Z, E88Z, Qu(23, 8), 4096, 4096, 4096, mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 40),
        ("c2T 1024^3 int<4,3> default tags", lower(E43, E43, E43, 1024, 1024, 1024), 20),
        ("c3T-like 2048^3 int<8,8> SAT::TCPL default tags", lower(Qu(8, 8), Qu(8, 8), Qu(8, 8), 2048, 2048, 2048), 5),
    ]
    out = {}
    with capi.Context(0) as ctx:
        for name, d, iters in cases:
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            plan.time_execute(pC, pA, pB, 3, iters)
            out[name] = plan.time_execute(pC, pA, pB, 1, iters)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    res = {l: {} for l in libs}
    for _ in range(rounds):
        for l in libs:
            r = json.loads(subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", l], text=True).strip().splitlines()[-1])
            for k, v in r.items():
                res[l].setdefault(k, []).append(v)
    for k in next(iter(res.values())):
        for l in libs:
            v = sorted(res[l][k])
            print(json.dumps({"workload": k, "library": l, "ms_rounds": [round(x, 5) for x in res[l][k]], "ms_median": v[len(v) // 2], "ms_min": v[0]}), flush=True)


if __name__ == "__main__":
    main()
