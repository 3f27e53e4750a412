#!/usr/bin/env python3
"""bench.py — throughput of the Qgemul hot path on MI355X, per the driver contract.

A "step" is one Qgemul over device-resident, already-packed synthetic fixed-point operands (raw integers uniform over the full
representable range, the distribution Qu::fill() draws, generated on the device by the same counter-based generator the CPU
oracle implements).

  python bench.py --gpus N --steps K --warmup W
      N = 1 runs in this process.  N > 1: this process touches no GPU; it starts N child ranks (one per GPU, RANK / LOCAL_RANK /
      WORLD_SIZE / MASTER_* in their environment), prints rank 0's JSON line and exits non-zero if any rank fails.  Started
      under torch.distributed.run (WORLD_SIZE already in the environment) the process IS a rank and starts nothing.

  primary line (`value`): BASELINE.json's metric configuration — 4096^3 Qgemul, int<8,8> signed operands (configs[2]) in the
      linear class (QgemulMulArgs<intBits<17>,fracBits<16>>, QgemulAddArgs<Qu<intBits<29>,fracBits<16>>>, C = Qu<23,8>) on the
      3 x 3 int8-limb MFMA kernel; N > 1: every rank computes its own 4096 rows of a (4096 N) x 4096 x 4096 product (B
      replicated) and ONE gather of the packed C shards to rank 0 per step is the only collective — the
      LIBRARY's RCCL gather (qgemul_comm_*, include/qgemul.h); scaling = "weak".
  extra.c4 (every N): BASELINE.json configs[3] — 16384 x 16384 x 4096 int<4,3>, linear class, STRONG scaling: rank r owns
      16384 / N rows (whole 256-row packed tiles), B replicated, packed 1-byte C, gathered to rank 0; reported without the
      gather, with one gather per step, and with the gather cut into row chunks that travel while later chunks compute.
  extra.c3T / c3Td / c2L / c2T / c5TF / c5B / c5L / reduce / long_k / w16 / u8 (N = 1): the remaining BASELINE configurations and
      the round's other paths (w16, u8: 16-bit words and unsigned bytes, stored centred), each with its own roofline block against its declared bound.

No PyTorch anywhere: device memory comes from the engine (qgemul_dev_alloc), the collective is the library's, timing uses HIP
events on the engine's stream (qgemul_time_execute) and the host clock around synchronised regions.  (Under
torch.distributed.run the 128-byte RCCL id travels through the launcher agent's TCPStore — qublas_amd/dist.py — nothing else.)

Roofline blocks.  `achieved` always comes from THIS run's HIP-event kernel time.  What a block quotes from rocprofv3 counters
(vector instructions per MAC, VALU busy share, clock, HBM-side traffic) comes from the committed `profiles/kernels.json`, and only
when that profile was taken on the kernel this run launches (engine kernel id + step form + source hashes, qublas_amd/profmeta.py);
otherwise the field is null and `profile_note` says why.  MFMA-class blocks are priced against the dense int8 peak with the
ALGORITHMIC op count (2 M N K, not the limb products the kernel issues); tree-class blocks against the vector-ALU issue rate
(one wave64 instruction per 4 cycles and SIMD at the profile's measured clock): `frac` there is lane-instructions per second over
that rate, i.e. the VALU busy share — it cannot exceed 1 — and `instr_per_mac` is the lever.
`cpu_baseline` times the reference's own primitives (oracle/_ref/ref_bench, built from /root/reference in the build container)
on a bounded block of the same workload, one process per host core.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_DENSE_PEAK_OPS = 5.0e15   # MI355X dense int8 MFMA, /opt/skills/guides/MI355X_MICROARCH.md (I8 = 2x BF16 ~2.5 PF)
HBM_PEAK = 8.0e12
N_SIMD, LANES, CYCLES_PER_VALU_INSTR = 256 * 4, 64, 4.0
# vector instructions per MAC that an exact leaf + node of the form cannot go below, counted on the ISA; None: no account
# (c2T: packed 16-bit halves, per MAC 1/2 multiply-add + 1/2 and + 1/2 add + 1/6 and; c3Td: the same steps unpacked; c5TF: per complex MAC
#  (3 multiply-adds + 3 and + 2 subtracts + 2 adds + 4/3 and) / 2; c3T: SAT::ZERO, split product 5 + range test and select per leaf and node 3)
ISA_FLOOR = {"c3T": 8.0, "c2T": 1.67, "c3Td": 3.33, "c5TF": 5.67, "w32T": 5.0}   # (w32T: v_mad_i64_i32, v_med3, v_lshrrev, v_mad_i32_i24 clamp + one saturating add per node)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3L", choices=["c3L", "c3T", "c3Td", "c2L", "c2T", "c4L", "c5TF", "c5B", "c5L", "reduce", "long_k", "w16", "u8", "w32T", "reduceW"])
    ap.add_argument("--size", type=int, default=4096, help="M=N=K per GPU of the primary workload")
    ap.add_argument("--backend", default="rccl", help="transport of the gather for N > 1: rccl (the library's communicator) or host (rehearsal on a box with fewer GPUs than ranks: packed bands through host memory + TCP)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the collective path even with one rank (the RCCL calls on a 1-GPU box)")
    ap.add_argument("--prewarm", type=int, default=300, help="untimed launches before the warm-up steps (lets the GPU clock settle); 0 for counter passes")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="--gpus N parent: seconds to wait for rank 0 before killing every rank")
    ap.add_argument("--c4-steps", type=int, default=20)
    ap.add_argument("--c4-chunks", type=int, default=0, help="row chunks of a rank's configuration-4 shard gathered while later chunks compute (0 = one per 256 tiles)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv) -> int:
    """Parent of an N-rank run.  Touches no GPU (imports nothing that could): children are fresh processes."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    try:
        out0 = procs[0].communicate(timeout=args.launch_timeout)[0]   # a rank stuck in a collective must not hang the caller for ever
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        print(f"bench.py: rank 0 did not finish within {args.launch_timeout} s; all ranks killed", file=sys.stderr)
        return 1
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(rc != 0 for rc in rcs) or line is None:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.kill()
        return 1
    print(line)
    return 0


# ---------------------------------------------------------------------------------------------------------------- workloads
def workloads():
    from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, RND, SAT, TRN, Tags, TFComplexMul
    e88z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    e43 = Qu(4, 3)
    r63, i63n = Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL)
    c5 = Qcomplex(r63, i63n)
    return {
        "c3L": dict(a=e88z, b=e88z, c=Qu(23, 8), mul=Tags(17, 16), add=[Qu(29, 16)], cfg="configs[2]", ref="c3L",
                    text="4096^3 Qgemul int<8,8> signed, linear class (MulArgs int17/frac16, AddArgs Qu<29,16>, C Qu<23,8>), 3x3 int8-limb MFMA"),
        "c3T": dict(a=e88z, b=e88z, c=e88z, mul=None, add=None, cfg="configs[2] as literally configured", ref="c3T",
                    text="4096^3 Qgemul int<8,8> signed TRN::TCPL/SAT::ZERO, default tags (tree class), exact tree on the vector ALUs"),
        "c3Td": dict(a=Qu(8, 8), b=Qu(8, 8), c=Qu(8, 8), mul=None, add=None, cfg="configs[2] with the reference's default modes", ref=None,
                     text="4096^3 Qgemul int<8,8> signed with the reference's default modes (TRN::TCPL / SAT::TCPL), default tags (tree class)"),
        "c2L": dict(a=e43, b=e43, c=e43, mul=Tags(9, 6), add=[Qu(19, 6)], cfg="configs[1]", ref="c2L",
                    text="1024^3 Qgemul int<4,3> signed, linear class (MulArgs int9/frac6, AddArgs Qu<19,6>), single-limb int8 MFMA"),
        "c2T": dict(a=e43, b=e43, c=e43, mul=None, add=None, cfg="configs[1], default tags", ref=None,
                    text="1024^3 Qgemul int<4,3> signed, default tags (tree class)"),
        "c4L": dict(a=e43, b=e43, c=e43, mul=Tags(9, 6), add=[Qu(21, 6)], cfg="configs[3]", ref="c2L",
                    text="16384x16384x4096 Qgemul int<4,3> signed, linear class (MulArgs int9/frac6, AddArgs Qu<21,6>), single-limb int8 MFMA, 1-byte packed C"),
        "c5TF": dict(a=c5, b=c5, c=c5, mul=TFComplexMul(), add=None, cfg="configs[4]", ref=None,
                     text="2048^3 Qgemul Qcomplex<int<6,3>,int<6,-3>> TFComplexMul (3 mul / 5 add), RND::POS_INF + SAT::TCPL, tree class"),
        "c5B": dict(a=c5, b=c5, c=c5, mul=BasicComplexMul(), add=None, cfg="configs[4] with BasicComplexMul", ref=None,
                    text="2048^3 Qgemul Qcomplex<int<6,3>,int<6,-3>> BasicComplexMul (4 mul / 2 add), RND::POS_INF + SAT::TCPL, tree class"),
        "c5L": dict(a=c5, b=c5, c=c5, mul=BasicComplexMul(acT=Qu(14, 6), bdT=Qu(14, -6), adT=Qu(14, 0), bcT=Qu(14, 0), acbdT=Qu(15, 6), adbcT=Qu(15, 0)),
                    add=[Qcomplex(Qu(30, 6), Qu(30, 0))], cfg="configs[4] operands, BasicComplexMul with exact sub-operation types (linear class)", ref=None,
                    text="2048^3 Qgemul Qcomplex<int<6,3>,int<6,-3>> BasicComplexMul, exact sub-op / level types: stacked 2x2 int8-limb MFMA + combine"),
        "w16": dict(a=Qu(7, 8), b=Qu(7, 8), c=Qu(23, 8), mul=Tags(15, 16), add=[Qu(27, 16)], cfg="16-bit words (Q7.8), the size of configs[2]", ref=None,
                    text="4096^3 Qgemul int<7,8> signed (16-bit words), linear class: operands stored centred (x - 128) in 2 x 2 int8 limbs instead of 3 x 3"),
        "u8": dict(a=Qu(8, 0, False), b=Qu(8, 0, False), c=Qu(28, 0, False), mul=Tags(16, 0, False), add=[Qu(28, 0, False)], cfg="unsigned bytes, the size of configs[2]", ref=None,
                   text="4096^3 Qgemul of unsigned 8-bit integers with exact sums, linear class: operands stored centred (x - 128) in ONE int8 limb, the centres restored in the kernel epilogue"),
        "w32T": dict(a=Qu(15, 16), b=Qu(15, 16), c=Qu(15, 16), mul=None, add=None, cfg="32-bit words (Q15.16), default tags", ref=None,
                     text="2048^3 Qgemul int<15,16> signed (32-bit words) with the reference's default modes and default tags (tree class): exact 64-bit products, one saturating 32-bit add per node"),
        "long_k": dict(a=e88z, b=e88z, c=Qu(33, 16), mul=Tags(17, 16), add=[Qu(33, 16)], cfg="configs[2] operands, K = 65536 (beyond one MFMA launch's exact int32 range)", ref=None,
                       text="4096x4096x65536 Qgemul int<8,8> signed, linear class: 2 k-chunks on the 3x3 int8-limb MFMA kernel + exact combine pass"),
    }


SHAPES = {"c3L": (4096, 4096, 4096), "c3T": (4096, 4096, 4096), "c3Td": (4096, 4096, 4096), "c2L": (1024, 1024, 1024), "c2T": (1024, 1024, 1024),
          "c4L": (16384, 16384, 4096), "c5TF": (2048, 2048, 2048), "c5B": (2048, 2048, 2048), "c5L": (2048, 2048, 2048), "reduce": (65536, 1, 4096),
          "long_k": (4096, 4096, 65536), "w16": (4096, 4096, 4096), "u8": (4096, 4096, 4096), "w32T": (2048, 2048, 2048), "reduceW": (65536, 1, 4096)}
REDUCE_TEXT = "batched Qreduce: 65536 vectors of 4096 int<8,8> elements (TRN::TCPL / SAT::ZERO), every tree node quantised; one wave per row"
REDUCEW_TEXT = "batched Qreduce: 65536 vectors of 4096 int<15,16> elements (32-bit words, default modes), every tree node one saturating 32-bit add; one wave per row"


def make_desc(name, wls, M, N, K):
    from qublas_amd.desc import Qu, SAT, TRN, lower, lower_reduce
    if name == "reduce":
        return lower_reduce(Qu(8, 8, True, TRN.TCPL, SAT.ZERO), M, K)
    if name == "reduceW":
        return lower_reduce(Qu(15, 16), M, K)
    wl = wls[name]
    return lower(wl["a"], wl["b"], wl["c"], M, N, K, mul_args=wl["mul"], add_args=wl["add"])


def make_plan(ctx, name, wls, M, N, K, flags=0):
    from qublas_amd import capi
    d = make_desc(name, wls, M, N, K)
    return capi.Plan(ctx, d, flags), d


class Bufs:
    """device buffers from the engine's allocator, freed together"""

    def __init__(self, ctx):
        self.ctx, self.ptrs = ctx, []

    def alloc(self, nbytes):
        p = self.ctx.alloc(max(int(nbytes), 16))
        self.ptrs.append(p)
        return p

    def free(self):
        for p in self.ptrs:
            self.ctx.free(p)
        self.ptrs = []


def cpu_baseline(variant: str, budget_s: float = 20.0):
    """Reference primitives on the host cores (process-parallel: Reducer keeps static buffers)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs_n = max(1, min(ncpu, 32))
    if os.path.exists(exe):
        # calibrate on a tiny block, then size each process's block to ~budget
        out = subprocess.check_output([exe, variant, "0", "8", "256"], text=True)
        cal = json.loads(out)
        rate = cal["macs"] / max(cal["seconds"], 1e-6)   # MAC/s of one process, measured alone
        cols = 4096 if cal["K"] == 4096 else 1024
        rows = int(max(1, min(4096 // procs_n, rate * budget_s / (cols * cal["K"]))))
        ps = [subprocess.Popen([exe, variant, str(i * rows), str(rows), str(cols)], stdout=subprocess.PIPE, text=True)
              for i in range(procs_n)]
        t0 = time.time()
        res = [json.loads(p.communicate()[0]) for p in ps]
        wall = time.time() - t0
        macs = sum(r["macs"] for r in res)
        return {"value": 2.0 * macs / wall, "unit": "int-op/s (2*M*N*K/s)", "cores": procs_n, "kind": "reference",
                "sample": f"{procs_n} processes x ({rows} rows x {cols} cols x K={res[0]['K']}) of the same workload, "
                          f"reference Qmul+Qreduce+convert via oracle/_ref/ref_bench {variant}, {wall:.1f} s wall"}
    # fall back to the C restatement (kind "port")
    import numpy as np  # noqa: F401
    from oracle import qoracle
    from qublas_amd.desc import lower
    wl = workloads()[variant if variant in ("c3L", "c3T", "c2L") else "c3L"]
    K = 4096 if variant.startswith("c3") else 1024
    rows, cols = 64, 256
    d = lower(wl["a"], wl["b"], wl["c"], rows, cols, K, mul_args=wl["mul"], add_args=wl["add"])
    A = qoracle.fill(wl["a"], rows * K, 1)
    B = qoracle.fill(wl["b"], K * cols, 2)
    t0 = time.time()
    qoracle.gemm(d, A, B, wl["c"], nthreads=procs_n)
    wall = time.time() - t0
    return {"value": 2.0 * rows * cols * K / wall, "unit": "int-op/s (2*M*N*K/s)", "cores": procs_n, "kind": "port",
            "sample": f"{rows}x{cols}x{K} block, oracle/qoracle.c with {procs_n} threads, {wall:.1f} s wall"}


def c_container_bytes(c_elem) -> int:
    """bytes of one packed C container per part: storage bits rounded up to 1 / 2 / 4 / 8 / 16 (qg_api.hip, pow2_bytes)"""
    from qublas_amd.desc import Qcomplex
    parts = [c_elem.real, c_elem.imag] if isinstance(c_elem, Qcomplex) else [c_elem]
    bits = max(p.storage_bits for p in parts)
    cb = 1
    while cb * 8 < bits:
        cb *= 2
    return cb


def roofline_block(name, info, M, N, K, kms, singles, capi, cb):
    """roofline of one kernel against ITS declared bound; kms: mean kernel time of THIS run (HIP events);
    cb: bytes of one packed C container"""
    from qublas_amd import profmeta
    kernel = capi.KERNEL_NAMES[info.kernel]
    reason = info.reason.decode()
    ops = float(info.ops)                       # 2 M N K (real), 6 / 8 M N K (complex TF / Basic real operations)
    macs = ops / 2.0
    achieved = ops / (kms * 1e-3)
    in_bytes = lambda bits: max(1, (bits + 7) // 8)
    parts = 2 if kernel.startswith("tree_cplx") or kernel == "mfma_cplx" else 1
    alg_bytes = int((M * K * in_bytes(info.in_bits[0]) + K * N * in_bytes(info.in_bits[1]) + M * N * cb) * parts)
    if name in ("reduce", "reduceW"):
        alg_bytes = int(4 * M * K + M * cb)       # the one-column kernels read 4-byte packed leaves once
    prof, why = profmeta.lookup(name, kernel, reason)
    r = {"kernel": kernel, "kernel_form": reason, "kernel_ms": kms, "kernel_ms_median_of_single_launches": singles[len(singles) // 2] if singles else None,
         "kernel_ms_min_of_single_launches": singles[0] if singles else None, "algorithmic_bytes": alg_bytes,
         "profile": prof["profile"] if prof else None, "profile_note": why,
         "traffic": prof["hbm"]["bytes_per_launch"] if prof and prof.get("hbm") else None}
    if prof:
        r["kernel_symbol"] = prof.get("kernel_symbol")
        r["traffic_source"] = f"{prof['profile']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of an earlier run of the same command on the same kernel (HEAD {prof.get('head')}), not this run"
    if kernel in ("mfma_i8", "mfma_i8_limb", "mfma_cplx"):
        limbs = max(1, info.limbs[0] * info.limbs[1])
        r.update({"bound": "mfma", "achieved": achieved / 1e12, "peak": INT8_DENSE_PEAK_OPS / 1e12, "unit": "TOP/s (int8-equivalent, algorithmic 2*M*N*K)",
                  "frac": achieved / INT8_DENSE_PEAK_OPS, "limbs": [info.limbs[0], info.limbs[1]], "mfma_issue_frac": achieved * limbs / INT8_DENSE_PEAK_OPS,
                  "mfma_pipe_busy": prof["mfma"]["pipe_busy"] if prof and prof.get("mfma") else None})
    elif kernel in ("gemv_i32", "gemv_i64"):
        r.update({"bound": "hbm", "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_bytes / (kms * 1e-3) / HBM_PEAK})
    else:
        v = prof.get("valu") if prof else None
        clk = prof.get("clock_ghz") if prof else None
        ipm = v["instr_per_mac"] if v else None
        peak = (N_SIMD * LANES * clk * 1e9 / CYCLES_PER_VALU_INSTR) if clk else None
        ach = (macs * ipm / (kms * 1e-3)) if ipm else None
        r.update({"bound": "valu", "unit": "T lane-instr/s (vector ALU: wave64 instructions x 64 lanes)",
                  "achieved": ach / 1e12 if ach else None, "peak": peak / 1e12 if peak else None,
                  "frac": min(1.0, ach / peak) if ach and peak else None,
                  "instr_per_mac": ipm, "instr_per_mac_isa_floor": ISA_FLOOR.get(name), "valu_busy_in_profile": v["valu_busy"] if v else None,
                  "clock_ghz_in_profile": clk,
                  "peak_source": "1024 SIMDs x 64 lanes x (profile's measured clock) / 4 cycles per wave64 instruction; frac = VALU busy share, capped at 1",
                  "ops_per_s": achieved, "pct_of_int8_peak": 100.0 * achieved / INT8_DENSE_PEAK_OPS,
                  "note": "tree class runs on the vector ALUs (no MFMA): the VALUs are the bound and instructions per MAC the lever; the int8-MFMA fraction is quoted only because the metric asks for it"})
    return r


def timed(plan, pc, pa, pb, iters):
    """mean kernel time over `iters` back-to-back launches + sorted single-launch times, HIP events on the engine's stream"""
    warm = 2 if iters <= 10 else 10
    ms = plan.time_execute(pc, pa, pb, warm, iters)
    singles = sorted(plan.time_execute(pc, pa, pb, 0, 1) for _ in range(min(iters, 10)))
    return ms, singles


def measure_config(ctx, name, wls, capi, iters, flags=0):
    M, N, K = SHAPES[name]
    p, _ = make_plan(ctx, name, wls, M, N, K, flags)
    bufs = Bufs(ctx)
    try:
        b = p.info.packed_bytes
        xa, xb, xc = bufs.alloc(b[0]), bufs.alloc(b[1]), bufs.alloc(b[2])
        p.fill(capi.OPERAND_A, 1, 0, xa)
        if name in ("reduce", "reduceW"):
            import numpy as np
            ctx.h2d(xb, np.ones(int(b[1]) // 4, np.int32))       # the Qreduce lowering's vector of ones
        else:
            p.fill(capi.OPERAND_B, 2, 0, xb)
        ms, singles = timed(p, xc, xa, xb, iters)
        rec = {"config": wls[name]["cfg"] if name in wls else "SURVEY.md 8-f #1", "workload": wls[name]["text"] if name in wls else (REDUCEW_TEXT if name == "reduceW" else REDUCE_TEXT), "M": M, "N": N, "K": K,
               "launches_timed": iters, "value": float(p.info.ops) / (ms * 1e-3),
               "unit": "int-op/s (algorithmic: 2*M*N*K real, 6 / 8 M*N*K complex TF / Basic)", "class": "linear" if p.info.cls == 1 else "tree",
               "roofline": roofline_block(name, p.info, M, N, K, ms, singles, capi, c_container_bytes(wls[name]["c"]) if name in wls else 4)}
    finally:
        p.close()
        bufs.free()
    return rec


# ---------------------------------------------------------------------------------------------------------------- the c4 leg
def c4_leg(args, ctx, wls, capi, net, world, rank):
    """BASELINE configs[3], strong scaling: 16384 x 16384 x 4096 int<4,3>, rank r owns a band of whole 256-row tiles."""
    from qublas_amd.dist import row_partition
    M, N, K = SHAPES["c4L"]
    wl = wls["c4L"]
    parts = row_partition(M, world, 256)
    rows = parts[rank][1]
    max_rows = max(p[1] for p in parts)
    out = {"config": wl["cfg"], "workload": wl["text"], "M": M, "N": N, "K": K, "scaling": "strong", "rows_per_rank": [p[1] for p in parts],
           "world_size": world, "rccl_world_size": net.reported_world() if net else 1, "backend": net.name if net else None, "steps": args.c4_steps}
    steps = args.c4_steps
    ops = 2.0 * M * N * K

    def run_variant(nchunks, gather):
        """nchunks row chunks per rank, each its own execute; gather: None | 'end' (one collective per step) | 'chunk'."""
        crow = max_rows // nchunks
        plan, _ = make_plan(ctx, "c4L", wls, crow, N, K)
        bufs = Bufs(ctx)
        try:
            pb = plan.info.packed_bytes
            cbytes = int(pb[2])
            counts = [max(0, min(nchunks, (parts[r][1] + crow - 1) // crow)) for r in range(world)]   # chunks every rank really owns (ragged partitions own fewer)
            myc = counts[rank]
            tB = bufs.alloc(pb[1])
            tAs = [bufs.alloc(pb[0]) for _ in range(nchunks)]
            # two generations of C so that the gather of step i reads while step i+1 writes
            tCs = [[bufs.alloc(cbytes) for _ in range(nchunks)] for _ in range(2)]
            plan.fill(capi.OPERAND_B, 2, 0, tB)
            for c in range(nchunks):
                plan.fill(capi.OPERAND_A, 1 + 1000 * rank + 17 * c, 0, tAs[c])
            land = None
            if gather and rank == 0:
                land = [[[tCs[g][c] if r == 0 else (bufs.alloc(cbytes) if c < counts[r] else None) for r in range(world)] for c in range(nchunks)] for g in range(2)]
            ctx.sync()

            def send(g, c):
                if rank == 0:
                    net.gather(tCs[g][c], cbytes if c < myc else 0, land[g][c], [cbytes if c < counts[r] else 0 for r in range(world)], slot=g)
                else:
                    net.gather(tCs[g][c], cbytes if c < myc else 0, slot=g)

            def step(i):
                g = i & 1
                if gather:
                    net.fence(g)                # (device-side: generation g was handed to gathers two steps ago; they are done before it is rewritten —
                                                #  the gathers of the OTHER generation travel on while this step's GEMMs run)
                for c in range(nchunks):
                    if c < myc:
                        plan.execute(tCs[g][c], tAs[c], tB)
                    if gather == "chunk":
                        send(g, c)
                if gather == "end":
                    for c in range(nchunks):    # (with nchunks == 1 this is THE one gather of the path)
                        send(g, c)

            def fence():
                if net:
                    net.fence()
                    net.barrier()
                ctx.sync()

            # untimed: the clock settles (tools/launch_gap.py: the first few hundred ms of MFMA work after an idle period ramp), as for the primary
            for _ in range(max(0, min(args.prewarm, 60)) // nchunks):
                for c in range(myc):
                    plan.execute(tCs[0][c], tAs[c], tB)
            ctx.sync()
            for i in range(3):
                step(i)
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            fence()
            dt = time.perf_counter() - t0
            if net:
                dt = net.max_f64(dt)
            ev = plan.time_execute(tCs[0][0], tAs[0], tB, 2, max(10, steps)) * myc if myc else 0.0   # this rank's GEMM work of one step by HIP events
            if net:
                ev = net.max_f64(ev)
            info = plan.info
        finally:
            plan.close()
            bufs.free()
        return dt / steps * 1e3, cbytes, ev, info, counts

    ms0, cbytes, ev0, info, counts = run_variant(1, None)
    out["ms_per_step_compute_only"] = ms0
    out["ms_per_step_events_compute_only"] = ev0
    out["value_compute_only"] = ops / (ms0 * 1e-3)
    out["shard_kernel_ms"] = ev0
    out["gather_bytes_per_rank"] = cbytes
    if net:
        ms1, _, ev1, _, _ = run_variant(1, "end")
        out["ms_per_step_one_gather"] = ms1
        out["ms_per_step_events_one_gather"] = ev1
        out["value_one_gather"] = ops / (ms1 * 1e-3)
        # what the gather alone costs: bytes the root receives per step over the time the step grew by
        recv = cbytes * (world - 1)
        out["gather_exposed_ms"] = max(0.0, ms1 - ms0)
        out["gather_GBps_into_root_exposed"] = (recv / (max(ms1 - ms0, 1e-6) * 1e-3) / 1e9) if world > 1 else None
        tiles = (max_rows // 256) * (N // 256)
        nch = args.c4_chunks if args.c4_chunks > 0 else max(1, min(8, tiles // 256))
        while nch > 1 and (max_rows % nch or (max_rows // nch) % 256):
            nch -= 1
        out["chunks"] = nch
        if nch > 1:
            ms2, cb2, ev2, _, _ = run_variant(nch, "chunk")
            out["ms_per_step_chunked_gather"] = ms2
            out["ms_per_step_events_chunked_gather"] = ev2
            out["value_chunked_gather"] = ops / (ms2 * 1e-3)
            out["chunk_bytes"] = cb2
            # share of the one-gather form's exposed time that chunking hid (1 = the gather travels entirely under later chunks' GEMMs)
            out["gather_overlap_fraction"] = (1.0 - max(0.0, ms2 - ms0) / (ms1 - ms0)) if ms1 - ms0 > 1e-6 else None
    if world == 1 and rank == 0:
        out["roofline"] = roofline_block("c4L", info, M, N, K, ev0, [ev0], capi, c_container_bytes(wl["c"]))
    return out


# ---------------------------------------------------------------------------------------------------------------- a rank
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (this pool's driver: dmabuf IPC only — RCCL's peer mappings need it)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args, argv))

    import numpy as np
    from qublas_amd import capi, profmeta
    from qublas_amd.dist import make_transport

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend == "host":
        local = 0                       # rehearsal: the ranks share the first card
    use_dist = world > 1 or args.force_dist
    ctx = capi.Context(local)           # (fails with QG_EINVAL when there is no such device: one rank per GPU)
    net, backend_note = None, None
    if use_dist:
        try:
            net = make_transport(ctx, args.backend)
        except Exception as e:   # e.g. no loadable librccl on this node: say so in the line and move the bands through host memory instead
            if args.backend != "rccl" or world == 1:
                raise
            backend_note = f"rccl unavailable on this node ({type(e).__name__}: {e}); the gather went through host memory + TCP (backend host)"
            print(f"[bench rank {rank}] {backend_note}", file=sys.stderr, flush=True)
            net = make_transport(ctx, "host")

    wls = workloads()
    S = args.size
    name = args.workload
    M, N, K = (S, S, S) if name in ("c3L", "c3T") else SHAPES[name]
    plan, d = make_plan(ctx, name, wls, M, N, K)
    info = plan.info
    pb = info.packed_bytes
    bufs = Bufs(ctx)
    tA, tB = bufs.alloc(pb[0]), bufs.alloc(pb[1])
    tCs = [bufs.alloc(pb[2]) for _ in range(2 if use_dist else 1)]   # the gather of step i travels while the GEMM of step i+1 runs
    tC = tCs[0]
    plan.fill(capi.OPERAND_A, 1 + 1000 * rank, 0, tA)   # rank r's rows of A: a distinct seed stream
    if name in ("reduce", "reduceW"):
        ctx.h2d(tB, np.ones(int(pb[1]) // 4, np.int32))
    else:
        plan.fill(capi.OPERAND_B, 2, 0, tB)
    cbytes = int(pb[2])
    land = None
    if use_dist and rank == 0:
        land = [[tCs[g] if r == 0 else bufs.alloc(cbytes) for r in range(world)] for g in range(2)]
    ctx.sync()
    state = {"i": 0}

    def step():
        g = (state["i"] & 1) if use_dist else 0
        state["i"] += 1
        if use_dist:
            net.fence(g)               # device-side: the GEMM below is ordered behind the gather that still reads buffer g (two steps ago)
        plan.execute(tCs[g], tA, tB)
        if use_dist:                   # the ONE collective of the path
            if rank == 0:
                net.gather(tCs[g], cbytes, land[g], [cbytes] * world, slot=g)
            else:
                net.gather(tCs[g], cbytes, slot=g)

    def barrier():
        if use_dist:
            net.fence()
            net.barrier()
        ctx.sync()

    # setup, untimed: bring the card to its steady clock before the W warm-up steps (the first few hundred ms of MFMA work
    # after an idle period run on a ramping clock, tools/launch_gap.py)
    slow = name in ("c3T", "c3Td", "c5TF", "c5B", "long_k", "w32T")
    PREWARM = max(0, args.prewarm) if not slow else min(max(0, args.prewarm), 20)
    for _ in range(PREWARM):
        plan.execute(tCs[0], tA, tB)
    ctx.sync()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        dt = net.max_f64(dt)

    ops_rank = float(info.ops)
    value = ops_rank * world * args.steps / dt

    out = None
    if rank == 0:
        # dominant kernel, HIP events on the engine's own stream; SURVEY.md §8-d asks for median and min beside the mean
        kms = plan.time_execute(tC, tA, tB, 3, max(10, min(args.steps, 100)))
        singles = sorted(plan.time_execute(tC, tA, tB, 0, 1) for _ in range(30))
        roof = roofline_block(name, info, M, N, K, kms, singles, capi, c_container_bytes(wls[name]["c"]) if name in wls else 4)
        mfma = roof["bound"] == "mfma"
        wtext = wls[name]["text"] if name in wls else (REDUCEW_TEXT if name == "reduceW" else REDUCE_TEXT)
        out = {"metric": "int-MAC/s (2*M*N*K/s) for Qgemul 4096^3 int<8,8>; % of MI355X int8 peak", "value": value,
               "unit": "int-op/s (2*M*N*K/s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "ms_per_step_events": kms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "i8 limbs -> i32/i64" if mfma else "i32", "data": "synthetic",
               "config": {"workload": wtext, "baseline_config": wls[name]["cfg"] if name in wls else "SURVEY.md 8-f #1", "M_per_gpu": M, "N": N, "K": K,
                          "class": "linear" if info.cls == 1 else "tree",
                          "sharding": f"rows of A/C over {world} rank(s), B replicated, one RCCL gather of C to rank 0" if world > 1 else "single GPU"},
               "prewarm_launches": PREWARM,
               "pct_of_int8_peak": 100.0 * value / (INT8_DENSE_PEAK_OPS * world),
               "roofline": roof,
               "profile_key": profmeta.profile_key(roof["kernel"], roof["kernel_form"], ops_rank / 2.0)}
        if use_dist:
            out["value_without_gather"] = ops_rank / (kms * 1e-3) * world   # the curve with and without the gather
            out["gather_bytes_per_step_per_rank"] = cbytes
            out["rccl_world_size"] = net.reported_world()    # rccl: ncclCommCount of the library's communicator
            out["backend"] = net.name
            if backend_note:
                out["backend_note"] = backend_note
            if net.name == "rccl":
                out["rccl_version"] = net.comm.info()[2]
            ms_step = dt / args.steps * 1e3
            out["gather_exposed_ms"] = max(0.0, ms_step - kms)
            out["gather_GBps_into_root_exposed"] = (cbytes * (world - 1) / (max(ms_step - kms, 1e-6) * 1e-3) / 1e9) if world > 1 else None
    extra = {}
    if not args.no_extra:
        # configuration 4 runs on every rank count (strong scaling); collective inside: every rank takes part
        try:
            c4 = c4_leg(args, ctx, wls, capi, net, world, rank)
            if rank == 0:
                extra["c4"] = c4
        except Exception as e:
            if use_dist:
                raise              # a rank that drops out of a collective must not leave the others waiting
            extra["c4"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_extra:
        # the layout steps either side of the hot path, timed separately (never part of `value`)
        try:
            hb = info.host_elem_bytes
            # host-layout operands with full-range values (zeros would select the 2 x 2-limb path through the plane masks and
            # flatter the timings)
            ea, eb_ = wls[name]["a"], wls[name]["b"]
            assert hb[0] == 4 and hb[1] == 4
            rng = np.random.default_rng(1)
            lb = Bufs(ctx)
            hA, hB, hC = lb.alloc(M * K * 4), lb.alloc(K * N * 4), lb.alloc(M * N * hb[2])
            ctx.h2d(hA, rng.integers(ea.raw_min, ea.raw_max + 1, M * K, dtype=np.int32))
            ctx.h2d(hB, rng.integers(eb_.raw_min, eb_.raw_max + 1, K * N, dtype=np.int32))
            lay = {}
            for nm, fn in (("pack_a_ms", lambda: plan.pack(capi.OPERAND_A, hA, tA)),
                           ("pack_b_ms", lambda: plan.pack(capi.OPERAND_B, hB, tB)),
                           ("unpack_c_ms", lambda: plan.unpack_c(tC, hC))):
                fn()
                ctx.sync()
                t1 = time.perf_counter()
                for _ in range(5):
                    fn()
                ctx.sync()
                lay[nm] = (time.perf_counter() - t1) / 5 * 1e3
            # the same C written by the kernel's own epilogue in the reference layout (no packed C, no unpack pass)
            plan.execute_host_c(hC, tA, tB)
            ctx.sync()
            t1 = time.perf_counter()
            for _ in range(10):
                plan.execute_host_c(hC, tA, tB)
            ctx.sync()
            lay["gemm_into_host_layout_c_ms"] = (time.perf_counter() - t1) / 10 * 1e3
            lay["epilogue_stores_host_layout"] = bool(plan.stores_host_c)
            lay["host_layout_call_ms"] = lay["pack_a_ms"] + lay["pack_b_ms"] + lay["gemm_into_host_layout_c_ms"]
            lay["host_layout_bytes"] = [M * K * 4, K * N * 4, M * N * hb[2]]
            out["layout_steps"] = lay
            lb.free()
        except Exception as e:
            out["layout_steps"] = {"error": str(e)}
        for nm, iters in (("c3T", 10), ("c3Td", 10), ("c2L", 200), ("c2T", 20), ("c5TF", 10), ("c5B", 10), ("c5L", 50), ("reduce", 50), ("long_k", 5), ("w16", 50), ("u8", 50), ("w32T", 10), ("reduceW", 50)):
            if nm == name:
                continue
            try:
                extra[nm] = measure_config(ctx, nm, wls, capi, iters)
            except Exception as e:  # an extra line must never take the primary line down
                extra[nm] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and extra:
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline((wls[name]["ref"] if name in wls else None) or "c3L")
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "int-op/s (2*M*N*K/s)", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    plan.close()
    ctx.sync()
    if use_dist:
        net.barrier()
        if net.name == "rccl":
            net.comm.close()
        else:
            net.ch.close()
    bufs.free()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
