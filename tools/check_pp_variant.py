#!/usr/bin/env python3
"""Diagnostic build only: the packed C of a k_mfma_pp A/B variant (selected by an environment switch of libqugemm_diag.so)
must equal the default kernel's, byte for byte.  usage: QUBLAS_AMD_DIAG=1 python tools/check_pp_variant.py QG_PP_PH2"""
import os
import sys

import numpy as np

os.environ["QUBLAS_AMD_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

E43 = Qu(4, 3)
E88 = Qu(8, 8, True, 5, 1)


def main():
    var = sys.argv[1]
    val = sys.argv[2] if len(sys.argv) > 2 else "1"
    with capi.Context(0) as ctx:
        limb = var.startswith("QG_PPL")
        for M, N, K in (((2048, 2048, 64), (2048, 2048, 128), (2100, 2200, 1000), (4096, 4096, 4096)) if limb else
                        ((4096, 4096, 128), (4096, 4096, 256), (4100, 4300, 1000), (8192, 8192, 4096))):
            if limb:
                d = lower(E88, E88, Qu(23, 8), M, N, K, mul_args=Tags(17, 16), add_args=[Qu(29, 16)])
            else:
                d = lower(E43, E43, E43, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(21, 6)])
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            plan.fill(capi.OPERAND_A, 1, 0 if limb else 1, pA)
            plan.fill(capi.OPERAND_B, 2, 0 if limb else 1, pB)
            outs = []
            for on in (False, True):
                if on:
                    os.environ[var] = val
                plan.execute(pC, pA, pB)
                ctx.sync()
                os.environ.pop(var, None)
                o = np.zeros(pb[2], np.uint8)
                ctx.d2h(o, pC)
                outs.append(o)
                ctx.h2d(pC, np.zeros(pb[2], np.uint8))
            assert np.array_equal(outs[0], outs[1]), (M, N, K)
            assert np.count_nonzero(outs[0]) > 0.3 * M * N
            print("equal", M, N, K, flush=True)
            for p in (pA, pB, pC):
                ctx.free(p)
            plan.close()


if __name__ == "__main__":
    main()
