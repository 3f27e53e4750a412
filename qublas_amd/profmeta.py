"""Which committed rocprofv3 summary may a bench line quote?  (bench.py, tools/summarize_profile.py)

`profiles/kernels.json` holds, per benchmark workload, what the PMC passes of tools/gpu_round.sh measured for the dominant
kernel (vector instructions per MAC, VALU busy share, clock, MFMA pipe share, HBM-side bytes) TOGETHER WITH what it was measured
on: the kernel's symbol, the engine's kernel id and step form (qgemul_info.kernel / .reason), the git HEAD and a hash of every
source file that kernel is compiled from.  bench.py quotes an entry only when the plan it has just timed reports the same
engine kernel and form and the hashes still match the sources in the tree; otherwise the block says why it is null.  Round 2
hard-coded such constants in bench.py and one of them went stale unnoticed (What's weak 1).
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qublas_amd", "csrc")
KERNELS_JSON = os.path.join(ROOT, "profiles", "kernels.json")

COMMON = ["qg_ops.h", "qg_kernels.h"]
FAMILY_SOURCES = {
    "mfma_i8": ["qg_mfma.hip", "qg_mfma_pp.hip", "qg_step_all.h"],
    "mfma_i8_limb": ["qg_mfma.hip", "qg_mfma_ppl.hip", "qg_step_all.h"],
    "mfma_cplx": ["qg_mfma.hip", "qg_mfma_ppl.hip", "qg_step_all.h", "qg_pack.hip"],
    "tree_i32": ["qg_tree_fast.hip", "qg_fix.h"],
    "tree_i64": ["qg_tree64.hip", "qg_tree.hip"],
    "tree_i128": ["qg_tree.hip"],
    "tree_cplx": ["qg_tree.hip"],
    "tree_cplx_i32": ["qg_tree_cplx.hip", "qg_fix.h"],
    "gemv_i32": ["qg_gemv.hip", "qg_fix.h"],
    "gemv_i64": ["qg_gemv.hip", "qg_fix.h"],
}


def source_hashes(engine_kernel: str) -> dict:
    out = {}
    for f in FAMILY_SOURCES.get(engine_kernel, []) + COMMON:
        p = os.path.join(CSRC, f)
        out[f] = hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None
    return out


def git_head() -> str:
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return os.environ.get("QG_GIT_HEAD", "unknown")     # (the GPU box receives a snapshot without .git: gpu_round.sh exports it)


def profile_key(engine_kernel: str, reason: str, macs: float) -> dict:
    """what a profiled bench run records about itself (copied into kernels.json by tools/summarize_profile.py)"""
    return {"engine_kernel": engine_kernel, "engine_reason": reason, "macs": macs, "sources": source_hashes(engine_kernel), "head": git_head()}


def lookup(workload: str, engine_kernel: str, reason: str):
    """(entry, None) when the committed profile of `workload` was taken on the kernel this run launches, else (None, why)"""
    if not os.path.exists(KERNELS_JSON):
        return None, "profiles/kernels.json not present"
    try:
        e = json.load(open(KERNELS_JSON)).get(workload)
    except Exception as ex:
        return None, f"profiles/kernels.json unreadable: {ex}"
    if not e:
        return None, f"no committed profile of workload {workload}"
    if e.get("engine_kernel") != engine_kernel or e.get("engine_reason") != reason:
        return None, (f"profile {e.get('profile')} was taken on engine kernel {e.get('engine_kernel')!r} / {e.get('engine_reason')!r}, "
                      f"this run launches {engine_kernel!r} / {reason!r}")
    now = source_hashes(engine_kernel)
    changed = sorted(f for f, h in now.items() if e.get("sources", {}).get(f) != h)
    if changed:
        return None, f"profile {e.get('profile')} (HEAD {e.get('head')}) predates changes to {', '.join(changed)}"
    return e, None
