#!/usr/bin/env python3
"""In-kernel clock stamps of k_mfma_pp (diagnostic build: QUBLAS_AMD_DIAG=1): where a tile's time goes.  Prints, per shape,
the median over workgroups and waves of each segment in shader cycles: k-loop, levelling barrier, epilogue, start of the next
tile.  Needs an MI355X and libqugemm_diag.so."""
import ctypes as C
import json
import os
import sys

import numpy as np

os.environ["QUBLAS_AMD_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

E43 = Qu(4, 3)


def main():
    L = capi.lib()
    L.qgemul_diag_set_stamps.argtypes = [C.c_void_p]
    L.qgemul_diag_set_stamps.restype = None
    shapes = [(16384, 16384, 4096, E43), (16384, 16384, 4096, Qu(16, 3))]
    with capi.Context(0) as ctx:
        nb = 256 * 8 * 64 * 4
        dbg = ctx.alloc(nb)
        for M, N, K, ec in shapes:
            d = lower(E43, E43, ec, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0, pA)
            plan.fill(capi.OPERAND_B, 2, 0, pB)
            for _ in range(20):
                plan.execute(pC, pA, pB)
            ctx.sync()
            L.qgemul_diag_set_stamps(C.c_void_p(dbg))
            for _ in range(3):
                plan.execute(pC, pA, pB)
            ctx.sync()
            L.qgemul_diag_set_stamps(None)
            raw = np.zeros(nb // 4, np.uint32)
            ctx.d2h(raw, dbg)
            s = raw.reshape(256, 8, 8, 8).astype(np.int64)   # [block][wave][tile][stamp]
            def seg(a, b, tiles=slice(1, 7), waves=slice(0, 8)):
                x = (s[:, waves, tiles, b] - s[:, waves, tiles, a]) & 0xffffffff
                return float(np.median(x))
            def nxt(a, b, waves=slice(0, 8)):   # stamp b of tile t+1 minus stamp a of tile t
                x = (s[:, waves, 2:7, b] - s[:, waves, 1:6, a]) & 0xffffffff
                return float(np.median(x))
            g0, g1 = slice(0, 4), slice(4, 8)
            rec = {"stagger": os.environ.get("QG_PP_STAGGER", "default"), "shape": [M, N, K], "c_bytes": int(pb[2] // (M * N)),
                   "k_loop": seg(0, 2), "first_load_interval": seg(0, 1), "first_ktile_to_phase3": seg(0, 5),
                   "level_barrier_g0": seg(2, 3, waves=g0), "level_barrier_g1": seg(2, 3, waves=g1),
                   "epilogue_g0": seg(3, 4, waves=g0), "epilogue_g1": seg(3, 4, waves=g1),
                   "epilogue_end_to_next_tile_start_g0": nxt(4, 0, g0), "epilogue_end_to_next_tile_start_g1": nxt(4, 0, g1),
                   "tile_period": nxt(0, 0)}
            print(json.dumps(rec), flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()


if __name__ == "__main__":
    main()
