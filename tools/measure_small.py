#!/usr/bin/env python3
"""Single-limb MFMA kernel on small problems: 64x64 tiles (default below 129 tiles of 128x128) vs 128x128 tiles
(QG_NO_SMALL_TILES=1).  Run once per setting; prints one JSON line per shape.  Needs an MI355X."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

E43 = Qu(4, 3)
E88 = Qu(8, 8)
with capi.Context() as ctx:
    for M, N, K, limb in ((1024, 1024, 1024, 0), (512, 512, 4096, 0), (1536, 1024, 1024, 0), (2048, 1024, 512, 0), (256, 256, 4096, 0),
                          (1024, 1024, 1024, 1), (512, 512, 4096, 1), (1536, 1024, 1024, 1)):
        if limb:   # int<8,8>: 3 x 3 int8 limbs
            d = lower(E88, E88, Qu(23, 8), M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
        else:
            d = lower(E43, E43, E43, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
        plan = capi.Plan(ctx, d)
        pb = plan.info.packed_bytes
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        plan.fill(capi.OPERAND_A, 1, 0, pA)
        plan.fill(capi.OPERAND_B, 2, 0, pB)
        ms = min(plan.time_execute(pC, pA, pB, 200, 500) for _ in range(3))
        print(json.dumps({"limbs": 3 if limb else 1, "M": M, "N": N, "K": K, "small_tiles": "QG_NO_SMALL_TILES" not in os.environ, "kernel_us": ms * 1e3,
                          "T_op_per_s": 2.0 * M * N * K / (ms * 1e-3) / 1e12}), flush=True)
        for p in (pA, pB, pC):
            ctx.free(p)
        plan.close()
