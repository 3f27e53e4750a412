#!/usr/bin/env python3
"""Cost of the element-wise epilogue on the two benchmarked MFMA kernels: plain GEMM, GEMM storing C
+ the chain as one pass (the default), and GEMM with the chain inside the kernel's epilogue (QG_OPT_FUSED_EPILOGUE).
Prints one JSON line per workload.  Needs an MI355X."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Ew, Qu, SAT, TRN, Tags, lower, lower_epilogue  # noqa: E402

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E43 = Qu(4, 3)
WORK = {
    # C of 24 storage bits: the chain (x 7-bit scale, + 17-bit bias) stays within 32-bit arithmetic -> can be fused
    "c3L": (E88, Qu(15, 8), 4096, 4096, 4096, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)])),
    "c2L": (E43, Qu(15, 8), 8192, 8192, 4096, dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)])),
    # C of 32 storage bits: the product with the scale needs 64-bit arithmetic -> never fused
    "c3L_wideC": (E88, Qu(23, 8), 4096, 4096, 4096, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)])),
    "c2L_i8out": (E43, Qu(4, 3), 8192, 8192, 4096, dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)])),
}


def main():
    iters = int(os.environ.get("ITERS", "50"))
    rng = np.random.default_rng(7)
    with capi.Context() as ctx:
        for name, (ea, ec, M, N, K, kw) in WORK.items():
            d = lower(ea, ea, ec, M, N, K, **kw)
            bias = Qu(10, 6) if ec.storage_bits > 8 else Qu(4, 3)
            scale = Qu(3, 3)
            stages = [Ew("mul", scale, Tags(ec.intBits + 1, ec.fracBits), scalar=True, into=Qu(ec.intBits + 1, ec.fracBits)),
                      Ew("add", bias)]
            ep = lower_epilogue(ec, stages, ec)
            res = {"workload": name, "M": M, "N": N, "K": K, "chain": "D = cvt_C(Qadd(Qmul<I+1,F>(C, s), Bias))"}
            base = capi.Plan(ctx, d)
            pA = ctx.alloc(int(base.info.packed_bytes[0]))
            pB = ctx.alloc(int(base.info.packed_bytes[1]))
            pC = ctx.alloc(int(base.info.packed_bytes[2]))
            base.fill(capi.OPERAND_A, 1, 0, pA)
            base.fill(capi.OPERAND_B, 2, 0, pB)
            base.time_execute(pC, pA, pB, 50, 50)   # clock warm-up
            res["plain_ms"] = min(base.time_execute(pC, pA, pB, 50, iters) for _ in range(2))
            Eh = rng.integers(bias.raw_min, bias.raw_max + 1, size=M * N, dtype=np.int64).astype(np.int32)
            dE = ctx.alloc(Eh.nbytes)
            ctx.h2d(dE, Eh)
            for label, flags in (("pass", capi.OPT_UNFUSED_EPILOGUE), ("fused", capi.OPT_FUSED_EPILOGUE), ("default", 0)):
                plan = capi.Plan(ctx, d, flags=flags, epilogue=ep)
                pE = ctx.alloc(plan.packed_e_bytes(1))
                plan.pack_e(1, dE, pE)
                args = plan.ep_args(packed=[0, pE], scalars=[13, 0])
                # 50 untimed launches first: the card's clock sags during the host-side setup above (tools/launch_gap.py)
                res[label + "_ms"] = min(plan.time_execute_ep(pC, pA, pB, args, 50, iters) for _ in range(2))
                res["e_bytes"] = plan.packed_e_bytes(1)
                if label == "fused":
                    res["_fused"] = plan.fuses_epilogue()
                if label == "default":
                    res["default_is_fused"] = plan.fuses_epilogue()
                ctx.sync()
                plan.close()
                ctx.free(pE)
            res["c_bytes"] = int(base.info.packed_bytes[2])
            # algorithmic traffic of the stand-alone pass: read C, read E, write D
            res["pass_bytes"] = 2 * res["c_bytes"] + res["e_bytes"]
            res["pass_GBps"] = res["pass_bytes"] / max(1e-9, (res["pass_ms"] - res["plain_ms"])) / 1e6
            res["fused_eligible"] = bool(res.pop("_fused"))
            print(json.dumps(res), flush=True)
            base.close()
            for ptr in (pA, pB, pC, dE):
                ctx.free(ptr)


if __name__ == "__main__":
    main()
