"""Composite linear plans at full size: kernel time of the limb-group / k-chunk plans on the MFMA kernels against the exact
64-bit tree kernel the same descriptors ran on in round 2 (QG_OPT_FORCE_TREE, ONE launch: it takes seconds), plus the pack
time of the composite operands.  One JSON line per case.  ONLY=<substring> filters; TREE=0 skips the tree arm."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi
from qublas_amd.desc import Qu, SAT, TRN, Tags, lower

E43 = Qu(4, 3)
E88Z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E1212 = Qu(12, 12)
Q1516 = Qu(15, 16)
CASES = [
    ("int<4,3> 4096x4096x262144 (3 k-chunks, single limb)", lower(E43, E43, Qu(27, 6), 4096, 4096, 262144, mul_args=Tags(9, 6), add_args=[Qu(27, 6)])),
    ("int<8,8> 4096x4096x65536 (2 k-chunks, 3x3 limbs)", lower(E88Z, E88Z, Qu(33, 16), 4096, 4096, 65536, mul_args=Tags(17, 16), add_args=[Qu(33, 16)])),
    ("int<12,12> 4096^3 (2x2 groups of 2 limbs)", lower(E1212, E1212, Qu(37, 24), 4096, 4096, 4096, mul_args=Tags(25, 24), add_args=[Qu(37, 24)])),
    ("int<12,12> x int<4,3> 4096^3 (2x1 groups)", lower(E1212, E43, Qu(29, 15), 4096, 4096, 4096, mul_args=Tags(17, 15), add_args=[Qu(29, 15)])),
    ("int<12,12> 2048^3", lower(E1212, E1212, Qu(36, 24), 2048, 2048, 2048, mul_args=Tags(25, 24), add_args=[Qu(36, 24)])),
    # beyond 64 bits: Q15.16 words, exact 64-bit products, exact 76-bit sums (5 balanced limbs per word: groups 3 + 2; 128-bit combine)
    ("Q15.16 2048^3 exact accumulation, C Qu<43,32> (wide)", lower(Q1516, Q1516, Qu(43, 32), 2048, 2048, 2048, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])),
    ("Q15.16 2048^3 exact accumulation, C Q15.16", lower(Q1516, Q1516, Q1516, 2048, 2048, 2048, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])),
    ("Q15.16 4096^3 exact accumulation, C Q15.16", lower(Q1516, Q1516, Q1516, 4096, 4096, 4096, mul_args=Tags(31, 32), add_args=[Qu(43, 32)])),
]
only = os.environ.get("ONLY", "")
with capi.Context(0) as ctx:
    for name, d in CASES:
        if only and only not in name:
            continue
        rec = {"case": name, "M": d.M, "N": d.N, "K": d.K}
        for arm, fl, iters in (("mfma", 0, 10), ("tree", capi.OPT_FORCE_TREE, 1)):
            if arm == "tree" and (os.environ.get("TREE", "1") == "0" or d.K > 65536 or (d.M > 2048 and "Q15.16" in name)):   # (beyond 16 levels only the general kernel applies: tens of seconds)
                continue
            p = capi.Plan(ctx, d, fl)
            pb = p.info.packed_bytes
            a, b, c = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            t0 = time.perf_counter()
            p.fill(capi.OPERAND_A, 1, 0, a)
            p.fill(capi.OPERAND_B, 2, 0, b)
            ctx.sync()
            rec[arm + "_fill_ms"] = (time.perf_counter() - t0) * 1e3
            ms = p.time_execute(c, a, b, 1 if arm == "mfma" else 0, iters)
            rec[arm + "_kernel"] = capi.KERNEL_NAMES[p.info.kernel]
            rec[arm + "_ms"] = ms
            rec[arm + "_Top_s"] = float(p.info.ops) / (ms * 1e-3) / 1e12
            rec[arm + "_reason"] = p.info.reason.decode()
            rec[arm + "_packed_MB"] = [int(x) >> 20 for x in pb]
            p.close()
            for x in (a, b, c):
                ctx.free(x)
        if "tree_ms" in rec:
            rec["speedup"] = rec["tree_ms"] / rec["mfma_ms"]
        print(json.dumps(rec), flush=True)
