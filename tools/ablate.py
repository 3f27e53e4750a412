#!/usr/bin/env python3
"""Price the phases of the MFMA kernels: runs bench-sized launches of the diagnostic variants
(QG_ABLATE, results wrong by construction) in child processes and prints kernel ms for each.
   0 full | 1 no LDS-DMA in loop | 2 no fragment ds_reads in loop | 3 neither | 4 neither, no barrier | 5 no C stores | 6 LDS-DMA always re-reads k-tile 0 (all cache hits) | 32 single-limb kernel on the 32x32x32 MFMA shape instead of 16x16x64"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json
sys.path.insert(0, %r)
import bench
from qublas_amd import capi
wl = bench.workloads()[sys.argv[1]]
S = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
ctx = capi.Context(0)
plan, d = bench.make_plan(ctx, wl, *S)
pb = plan.info.packed_bytes
pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
plan.fill(0, 1, 0, pA); plan.fill(1, 2, 0, pB)
ms = plan.time_execute(pC, pA, pB, 3, 30)
print(json.dumps({"ms": ms}))
''' % ROOT

for wl, shape in (("c3L", (4096, 4096, 4096)), ("c2L", (8192, 8192, 4096))):
    for abl in ((0, 32, 0, 32, 0, 32) if wl == 'c2L' else (0, 1, 2, 3, 4, 5, 6, 0)):
        env = dict(os.environ, QG_ABLATE=str(abl))
        out = subprocess.check_output([sys.executable, "-c", CODE, wl, *map(str, shape)], env=env, text=True)
        ms = json.loads(out.strip().splitlines()[-1])["ms"]
        ops = 2.0 * shape[0] * shape[1] * shape[2]
        print(f"{wl} ablate={abl} kernel_ms={ms:.4f}  equiv {ops / ms / 1e9:.1f} TOP/s", flush=True)
