#!/usr/bin/env python3
"""Wall time of the one-shot drop-in call qgemul_run (host buffers in, host buffers out) for small and large problems.
Needs an MI355X."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, SAT, TRN, Tags, lower  # noqa: E402

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
rng = np.random.default_rng(1)
for S, kw, reps in ((4, dict(mul_args=E88, add_args=[E88]), 200), (64, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 200),
                    (512, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 50), (4096, dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 5)):
    ec = E88 if S == 4 else Qu(23, 8)
    d = lower(E88, E88, ec, S, S, S, **kw)
    A = rng.integers(E88.raw_min, E88.raw_max + 1, S * S, dtype=np.int32)
    B = rng.integers(E88.raw_min, E88.raw_max + 1, S * S, dtype=np.int32)
    C = np.zeros(S * S, dtype=np.int32)
    capi.run(d, C, A, B)
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.run(d, C, A, B)
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"S": S, "kernel": capi.KERNEL_NAMES[capi.classify(d).kernel], "ms_per_call": dt * 1e3}), flush=True)
