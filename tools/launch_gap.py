#!/usr/bin/env python3
"""Where does the wall-vs-kernel gap per step come from?  Same plan, same buffers: (a) HIP-event timing of N
back-to-back launches inside the library, (b) N ctypes launches from Python + one sync, wall clock."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qublas_amd import capi  # noqa: E402

wl = bench.workloads()["c3L"]
ctx = capi.Context(0)
plan, d = bench.make_plan(ctx, wl, 4096, 4096, 4096)
pb = plan.info.packed_bytes
pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
plan.fill(0, 1, 0, pA)
plan.fill(1, 2, 0, pB)
ctx.sync()
for rep in range(3):
    ms = plan.time_execute(pC, pA, pB, 5, 100)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        plan.execute(pC, pA, pB)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"events {ms:.4f} ms/launch | python loop: enqueue {1e3 * (t1 - t0) / 100:.4f} ms/launch, wall {1e3 * (t2 - t0) / 100:.4f} ms/launch", flush=True)
