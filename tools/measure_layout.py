#!/usr/bin/env python3
"""The layout steps either side of the hot path on REAL (full-range, non-zero) host-layout operands, device to device, and the two
ways of getting a reference-layout C: GEMM into packed C + unpack pass, against the GEMM's epilogue storing the reference layout
itself (qgemul_execute_host_c).  Interleaved rounds in one process; one JSON line per shape.  Needs an MI355X."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, SAT, TRN, Tags, lower  # noqa: E402

E43 = Qu(4, 3)
E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
CASES = [("c3L 4096^3 int<8,8> 3x3 limbs", E88, Qu(23, 8), dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 4096, 4096, 4096),
         ("8192^2 x 4096 int<4,3> single limb, 4-byte C", E43, Qu(16, 3), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 8192, 8192, 4096)]


def timed(ctx, fn, n):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    with capi.Context(0) as ctx:
        torch.cuda.set_stream(torch.cuda.ExternalStream(ctx.stream, device=dev))
        for name, e, ec, kw, M, N, K in CASES:
            d = lower(e, e, ec, M, N, K, **kw)
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            hA = torch.randint(e.raw_min, e.raw_max + 1, (M * K,), dtype=torch.int32, device=dev)
            hB = torch.randint(e.raw_min, e.raw_max + 1, (K * N,), dtype=torch.int32, device=dev)
            hC = torch.empty(M * N, dtype=torch.int32, device=dev)
            tA = torch.empty(int(pb[0]), dtype=torch.uint8, device=dev)
            tB = torch.empty(int(pb[1]), dtype=torch.uint8, device=dev)
            tC = torch.empty(int(pb[2]), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            fA = (torch.randn(M * K, dtype=torch.float64, device=dev) * float(2 ** (e.intBits - 2)))   # doubles for quantise-on-load
            torch.cuda.synchronize()
            steps = {"pack_a": lambda: plan.pack(capi.OPERAND_A, hA.data_ptr(), tA.data_ptr()),
                     "pack_a_from_doubles": lambda: plan.pack_f64(capi.OPERAND_A, fA.data_ptr(), tA.data_ptr()),
                     "pack_b": lambda: plan.pack(capi.OPERAND_B, hB.data_ptr(), tB.data_ptr()),
                     "gemm_packed_c": lambda: plan.execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr()),
                     "unpack_c": lambda: plan.unpack_c(tC.data_ptr(), hC.data_ptr()),
                     "gemm_into_host_layout_c": lambda: plan.execute_host_c(hC.data_ptr(), tA.data_ptr(), tB.data_ptr())}
            res = {k: [] for k in steps}
            for _ in range(5):
                for k, fn in steps.items():
                    res[k].append(timed(ctx, fn, 20))
            plan.pack(capi.OPERAND_A, hA.data_ptr(), tA.data_ptr())   # (the last pack_a_from_doubles left quantised doubles in tA)
            rec = {"case": name, "epilogue_stores_host_layout": bool(plan.stores_host_c)}
            for k, v in res.items():
                v.sort()
                rec[k + "_ms"] = v[len(v) // 2]
            rec["host_layout_call_packed_plus_unpack_ms"] = rec["pack_a_ms"] + rec["pack_b_ms"] + rec["gemm_packed_c_ms"] + rec["unpack_c_ms"]
            rec["host_layout_call_direct_ms"] = rec["pack_a_ms"] + rec["pack_b_ms"] + rec["gemm_into_host_layout_c_ms"]
            print(json.dumps(rec), flush=True)
            plan.close()


if __name__ == "__main__":
    main()
