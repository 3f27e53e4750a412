#!/usr/bin/env python3
"""Single-limb linear class, large problems: k_mfma_pp (two wave groups alternating on the matrix cores, 128-byte k-tiles)
against k_mfma16 (lock-step waves, 64-byte k-tiles; QG_OPT_LOCKSTEP_TILES) in ONE process, interleaved rounds, HIP events on
the engine's stream.  One JSON line per (shape, kernel) with mean / median / min over the rounds.  Needs an MI355X."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

E43 = Qu(4, 3)
E88 = Qu(8, 8, True, 5, 1)   # TRN::TCPL, SAT::ZERO
PEAK = 5.0e15
LIMB_SHAPES = [(4096, 4096, 4096), (8192, 8192, 4096), (4096, 4096, 2048), (4096, 4096, 1024), (4096, 4096, 8192), (2048, 2048, 4096)]
SHAPES = [(16384, 16384, 4096), (2048, 16384, 4096), (8192, 8192, 4096), (4096, 4096, 4096), (16384, 16384, 2048), (16384, 16384, 1024), (16384, 16384, 512)]


def main():
    rounds = int(os.environ.get("ROUNDS", "7"))
    iters = int(os.environ.get("ITERS", "20"))
    only = os.environ.get("ONLY")
    with capi.Context(0) as ctx:
        limb = os.environ.get("WL") == "limb"   # the 3 x 3-limb kernels (configuration 3's operands) instead of the single-limb ones
        for M, N, K in (LIMB_SHAPES if limb else SHAPES):
            if only and only != f"{M}x{N}x{K}":
                continue
            if limb:
                d = lower(E88, E88, Qu(23, 8), M, N, K, mul_args=Tags(17, 16), add_args=[Qu(30, 16)])
            else:
                d = lower(E43, E43, E43, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
            arms = []
            arm_list = [("pp", 0, None), ("lockstep", capi.OPT_LOCKSTEP_TILES, None)]
            if os.environ.get("QUBLAS_AMD_DIAG") == "1":   # diagnostic library: the launch-per-tile form of the same kernel, phase structures
                if not limb:
                    arm_list.insert(1, ("pp_launch_per_tile", 0, ("QG_PP_LAUNCH_PER_TILE", "1")))
                if os.environ.get("PHASES") and not limb:
                    arm_list.append(("pp_two_phases", 0, ("QG_PP_PH2", "1")))
                    arm_list.append(("pp_four_phases", 0, ("QG_PP_PH4", "1")))
                    for dm, nm in (("0", "4_4"), ("1", "2_6"), ("2", "0_8")):
                        arm_list.append((f"pp_two_phases_dma_{nm}", 0, ("QG_PP_DMA", dm)))
            for name, flags, envname in arm_list:
                plan = capi.Plan(ctx, d, flags)
                pb = plan.info.packed_bytes
                pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
                plan.fill(capi.OPERAND_A, 1, 0, pA)
                plan.fill(capi.OPERAND_B, 2, 0, pB)
                arms.append((name, plan, pA, pB, pC, [], envname))
            def timed(arm, warm, it):
                if arm[6]:
                    os.environ[arm[6][0]] = arm[6][1]
                try:
                    return arm[1].time_execute(arm[4], arm[2], arm[3], warm, it)
                finally:
                    if arm[6]:
                        del os.environ[arm[6][0]]
            for arm in arms:
                timed(arm, 5, 5)
            for _ in range(rounds):
                for arm in arms:
                    arm[5].append(timed(arm, 1, iters))
            for name, plan, pA, pB, pC, ts, _e in arms:
                s = sorted(ts)
                ops = 2.0 * M * N * K
                med = s[len(s) // 2]
                print(json.dumps({"shape": [M, N, K], "kernel": name, "ms_mean": sum(s) / len(s), "ms_median": med, "ms_min": s[0],
                                  "pct_int8_peak_median": 100.0 * ops / (med * 1e-3) / PEAK, "rounds": rounds, "iters": iters}), flush=True)
                for p in (pA, pB, pC):
                    ctx.free(p)
                plan.close()


if __name__ == "__main__":
    main()
