"""Build libqugemm.so (the C-ABI engine of include/qgemul.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["qg_api.hip", "qg_pack.hip", "qg_eltwise.hip", "qg_tree.hip", "qg_tree_fast.hip", "qg_tree64.hip", "qg_tree_cplx.hip", "qg_gemv.hip", "qg_mfma.hip", "qg_mfma_pp.hip", "qg_mfma_ppl.hip", "qg_comm.hip", "qg_plan.cpp"]
HEADERS = ["qg_ops.h", "qg_plan.h", "qg_kernels.h", "qg_step_all.h", "qg_eltwise.h", "qg_eltwise_args.h", "qg_fix.h", os.path.join("..", "..", "include", "qgemul.h")]
LIB = os.path.join(HERE, "libqugemm.so")
# the same sources with -DQG_DIAG: environment A/B switches and the ablation kernel variants (results may be WRONG by
# construction there).  Loaded only by tools/ (QUBLAS_AMD_DIAG=1, qublas_amd/capi.py); the product library has neither.
LIB_DIAG = os.path.join(HERE, "libqugemm_diag.so")


def _stale(obj: str, deps) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build", "diag") if diag else os.path.join(HERE, "build")
    LIB = LIB_DIAG if diag else globals()["LIB"]
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
                   "-x", "hip", "-c", src, "-o", obj] + (["-DQG_DIAG"] if diag else [])
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]   # (RCCL is bound with dlopen: qg_comm.hip)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
