#!/usr/bin/env python3
"""Tree class on 32-bit words (Q15.16 with default tags: every product and node quantised into Q15.16): kernel time."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, lower  # noqa: E402

def main():
    with capi.Context(0) as ctx:
        for q, S in ((Qu(15, 16), 1024), (Qu(15, 16), 2048), (Qu(11, 12), 2048), (Qu(14, 16), 2048), (Qu(15, 16, True, 5, 3), 2048)):   # (the last two: justified words)
            d = lower(q, q, q, S, S, S)
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            plan.time_execute(pC, pA, pB, 1, 3)
            ms = plan.time_execute(pC, pA, pB, 1, 5)
            print(json.dumps({"case": f"Q{q.intBits}.{q.fracBits} default tags {S}^3", "kernel": capi.KERNEL_NAMES[plan.info.kernel], "reason": plan.info.reason.decode(), "ms": round(ms, 4),
                              "instr_slots_per_mac_at_2.4GHz": round(ms * 1e-3 * 39.3e12 / S ** 3, 1)}), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()

def reduce_part():
    """batched Qreduce of 65 536 vectors of 4096 Q15.16 elements (1 GiB of elements): the 32-bit-word form of the one-column kernel
    against its 64-bit-value form (QG_OPT_RUNTIME_MODES), and a Q15.16 GEMV of the same size"""
    from qublas_amd.desc import lower_reduce
    q = Qu(15, 16)
    M, K = 65536, 4096
    with capi.Context(0) as ctx:
        for name, d in (("Qreduce Q15.16 65536 x 4096", lower_reduce(q, M, K)), ("GEMV Q15.16 65536 x 4096", lower(q, q, q, M, 1, K))):
            for flags in (0, capi.OPT_RUNTIME_MODES):
                plan = capi.Plan(ctx, d, flags)
                pb = plan.info.packed_bytes
                pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
                plan.fill(capi.OPERAND_A, 1, 0, pA)
                plan.fill(capi.OPERAND_B, 2, 0, pB)
                plan.time_execute(pC, pA, pB, 1, 3)
                ms = plan.time_execute(pC, pA, pB, 1, 10)
                print(json.dumps({"case": name, "kernel": capi.KERNEL_NAMES[plan.info.kernel], "reason": plan.info.reason.decode(), "ms": round(ms, 4),
                                  "TBps_of_elements": round(M * K * 4 / (ms * 1e-3) / 1e12, 2)}), flush=True)
                for p in (pA, pB, pC):
                    ctx.free(p)
                plan.close()


if __name__ == "__main__":
    reduce_part()
    main()
