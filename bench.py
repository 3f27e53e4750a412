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
      replicated) and ONE gather of the packed C shards to rank 0 per step is the only collective;
      scaling = "weak".  The default-tag (tree class) figure of the same operands rides along in `extra.c3T`.
  extra.c4 (every N): BASELINE.json configs[3] — 16384 x 16384 x 4096 int<4,3>, linear class, STRONG scaling: rank r owns
      16384 / N rows (whole 256-row packed tiles), B replicated, packed 1-byte C, gathered to rank 0; reported without the
      gather, with one gather per step, and with the gather cut into row chunks that travel while later chunks compute.
  extra.c2L / c2T / c5TF / c5B (N = 1): the remaining BASELINE configurations, each with its own roofline block against its
      declared bound (int8 MFMA, or the measured integer-VALU issue roof for the tree class).

`roofline` prices the dominant kernel against the dense int8 MFMA peak with the ALGORITHMIC op count (2 M N K, not the 9 limb
products the kernel issues); kernel time comes from HIP events on the engine's own stream (qgemul_time_execute).
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
# integer vector-ALU issue roof, measured: tools/ubench/valu_rate.hip sustains one wave64 instruction per 4.3 cycles and SIMD
# at 8 waves per SIMD: 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4.3
VALU_LANE_OPS_PEAK = 256 * 4 * 64 * 2.4e9 / 4.3
# vector-ALU instructions per MAC of the tree kernels, from rocprofv3 SQ_INSTS_VALU (profiles/: r03d_c3T)
# (c2T / c3Td: nodes and products that saturate (SAT::TCPL) clamp with one v_med3_i32: 5.3, profiles/r04c_c2T_pmc.json)
VALU_PER_MAC = {"c3T": 6.7, "c2T": 5.3, "c3Td": 5.3, "c5TF": 24.77 / 3.0, "c5B": 27.37 / 4.0}   # per real MAC (complex TF: 24.8 per complex MAC = 3 real MACs, Basic 27.4 = 4 real MACs: profiles/r04k_c5TF_pmc.json, r04k_c5B_pmc.json; TF 51 with run-time modes, 30.9 before the additions moved to the tile staging)
HBM_PEAK = 8.0e12


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3L", choices=["c3L", "c3T", "c2L", "c4L"])
    ap.add_argument("--size", type=int, default=4096, help="M=N=K per GPU of the primary workload")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the script on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the collective path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)")
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
    c5 = Qcomplex(Qu(6, 3, True, RND.POS_INF, SAT.TCPL), Qu(6, -3, True, RND.POS_INF, SAT.TCPL))
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
    }


SHAPES = {"c3L": (4096, 4096, 4096), "c3T": (4096, 4096, 4096), "c3Td": (4096, 4096, 4096), "c2L": (1024, 1024, 1024), "c2T": (1024, 1024, 1024),
          "c4L": (16384, 16384, 4096), "c5TF": (2048, 2048, 2048), "c5B": (2048, 2048, 2048)}


def make_plan(ctx, wl, M, N, K, flags=0):
    from qublas_amd import capi
    from qublas_amd.desc import lower
    d = lower(wl["a"], wl["b"], wl["c"], M, N, K, mul_args=wl["mul"], add_args=wl["add"])
    return capi.Plan(ctx, d, flags), d


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


def load_traffic(workload: str):
    """HBM-side bytes per launch of the dominant kernel from a COMMITTED rocprofv3 PMC summary (not measured in this run)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            rec = json.load(open(p)).get(workload, {})
            if rec.get("hbm_bytes_per_launch") is not None:
                return rec["hbm_bytes_per_launch"], f"profiles/traffic.json ({rec.get('profile', '?')}); rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of an earlier run of the same command, not this run"
        except Exception:
            pass
    return None, None


def c_container_bytes(c_elem) -> int:
    """bytes of one packed C container per part: storage bits rounded up to 1 / 2 / 4 / 8 (qg_api.hip, pow2_bytes)"""
    from qublas_amd.desc import Qcomplex
    parts = [c_elem.real, c_elem.imag] if isinstance(c_elem, Qcomplex) else [c_elem]
    bits = max(p.storage_bits for p in parts)
    cb = 1
    while cb * 8 < bits:
        cb *= 2
    return cb


def roofline_block(name, info, M, N, K, kms, singles, capi, cb):
    """roofline of one kernel against ITS declared bound."""
    kernel = capi.KERNEL_NAMES[info.kernel]
    ops = float(info.ops)                       # 2 M N K (real), 6 / 8 M N K (complex TF / Basic real operations)
    achieved = ops / (kms * 1e-3)
    in_bytes = lambda bits: max(1, (bits + 7) // 8)
    parts = 2 if kernel.startswith("tree_cplx") or kernel == "mfma_cplx" else 1
    alg_bytes = int((M * K * in_bytes(info.in_bits[0]) + K * N * in_bytes(info.in_bits[1]) + M * N * cb) * parts)
    traffic, tsrc = load_traffic(name)
    if kernel in ("mfma_i8", "mfma_i8_limb", "mfma_cplx"):
        limbs = max(1, info.limbs[0] * info.limbs[1])
        r = {"bound": "mfma", "achieved": achieved / 1e12, "peak": INT8_DENSE_PEAK_OPS / 1e12, "unit": "TOP/s (int8-equivalent, algorithmic 2*M*N*K)",
             "frac": achieved / INT8_DENSE_PEAK_OPS, "limbs": [info.limbs[0], info.limbs[1]], "mfma_issue_frac": achieved * limbs / INT8_DENSE_PEAK_OPS}
    elif kernel == "gemv_i32":
        r = {"bound": "hbm", "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_bytes / (kms * 1e-3) / HBM_PEAK}
    else:
        vpm = VALU_PER_MAC.get(name)
        macs = ops / 2.0
        r = {"bound": "valu", "unit": "T lane-instr/s (vector ALU, wave64 instructions x 64)", "peak": VALU_LANE_OPS_PEAK / 1e12,
             "achieved": (macs * vpm / (kms * 1e-3) / 1e12) if vpm else None,
             "frac": (macs * vpm / (kms * 1e-3) / VALU_LANE_OPS_PEAK) if vpm else None, "valu_instr_per_mac": vpm,
             "peak_source": "tools/ubench/valu_rate.hip: one wave64 integer instruction per 4.3 cycles and SIMD at 8 waves per SIMD (DESIGN.md)",
             "ops_per_s": achieved, "pct_of_int8_peak": 100.0 * achieved / INT8_DENSE_PEAK_OPS,
             "note": "tree class runs on the vector ALUs (no MFMA): priced against the integer-VALU issue roof; the int8-MFMA fraction is quoted only because the metric asks for it"}
    r.update({"traffic": traffic, "traffic_source": tsrc, "kernel": kernel, "kernel_ms": kms,
              "kernel_ms_median": singles[len(singles) // 2] if singles else None, "kernel_ms_min": singles[0] if singles else None,
              "algorithmic_bytes": alg_bytes})
    return r


def measure_config(ctx, name, wls, capi, torch, dev, iters, flags=0):
    M, N, K = SHAPES[name]
    p, _ = make_plan(ctx, wls[name], M, N, K, flags)
    b = p.info.packed_bytes
    xa = torch.empty(b[0], dtype=torch.uint8, device=dev)
    xb = torch.empty(b[1], dtype=torch.uint8, device=dev)
    xc = torch.empty(b[2], dtype=torch.uint8, device=dev)
    p.fill(capi.OPERAND_A, 1, 0, xa.data_ptr())
    p.fill(capi.OPERAND_B, 2, 0, xb.data_ptr())
    warm = 2 if iters <= 10 else 10
    ms = p.time_execute(xc.data_ptr(), xa.data_ptr(), xb.data_ptr(), warm, iters)
    singles = sorted(p.time_execute(xc.data_ptr(), xa.data_ptr(), xb.data_ptr(), 0, 1) for _ in range(min(iters, 10)))
    rec = {"config": wls[name]["cfg"], "workload": wls[name]["text"], "M": M, "N": N, "K": K, "launches_timed": iters,
           "value": float(p.info.ops) / (ms * 1e-3), "unit": "int-op/s (algorithmic: 2*M*N*K real, 6 / 8 M*N*K complex TF / Basic)",
           "class": "linear" if p.info.cls == 1 else "tree",
           "roofline": roofline_block(name, p.info, M, N, K, ms, singles, capi, c_container_bytes(wls[name]["c"]))}
    p.close()
    del xa, xb, xc
    return rec


# ---------------------------------------------------------------------------------------------------------------- the c4 leg
def c4_leg(args, ctx, wls, capi, torch, dist, dev, world, rank, use_dist, on_host):
    """BASELINE configs[3], strong scaling: 16384 x 16384 x 4096 int<4,3>, rank r owns a band of whole 256-row tiles."""
    from qublas_amd.dist import row_partition
    M, N, K = SHAPES["c4L"]
    wl = wls["c4L"]
    parts = row_partition(M, world, 256)
    rows = parts[rank][1]
    max_rows = max(p[1] for p in parts)
    out = {"config": wl["cfg"], "workload": wl["text"], "M": M, "N": N, "K": K, "scaling": "strong", "rows_per_rank": [p[1] for p in parts],
           "world_size": world, "rccl_world_size": dist.get_world_size() if use_dist else 1, "backend": (args.backend if use_dist else None),
           "steps": args.c4_steps}
    steps = args.c4_steps
    ops = 2.0 * M * N * K

    def run_variant(nchunks, gather):
        """nchunks row chunks per rank, each its own execute; gather: None | 'end' (one collective per step) | 'chunk'."""
        crow = max_rows // nchunks
        plan, _ = make_plan(ctx, wl, crow, N, K)
        pb = plan.info.packed_bytes
        myc = max(0, min(nchunks, (rows + crow - 1) // crow))      # chunks this rank really owns (ragged partitions own fewer)
        tB = torch.empty(pb[1], dtype=torch.uint8, device=dev)
        tAs = [torch.empty(pb[0], dtype=torch.uint8, device=dev) for _ in range(nchunks)]
        # two generations of C so that the collective of step i reads while step i+1 writes
        tCs = [[torch.zeros(pb[2], dtype=torch.uint8, device=dev) for _ in range(nchunks)] for _ in range(2)]
        plan.fill(capi.OPERAND_B, 2, 0, tB.data_ptr())
        for c in range(nchunks):
            plan.fill(capi.OPERAND_A, 1 + 1000 * rank + 17 * c, 0, tAs[c].data_ptr())
        torch.cuda.synchronize()
        glists = None
        if gather and rank == 0:
            glists = [[[torch.empty(pb[2], dtype=torch.uint8, device="cpu" if on_host else dev) for _ in range(world)]
                       for _ in range(nchunks)] for _ in range(2)]
        pending = [[], []]

        def step(i):
            g = i & 1
            for w in pending[g]:
                w.wait()                       # stream-level: buffers of generation g were handed to collectives two steps ago
            pending[g] = []
            for c in range(nchunks):
                if c < myc:
                    plan.execute(tCs[g][c].data_ptr(), tAs[c].data_ptr(), tB.data_ptr())
                if gather == "chunk":
                    src = tCs[g][c].cpu() if on_host else tCs[g][c]
                    pending[g].append(dist.gather(src, glists[g][c] if rank == 0 else None, dst=0, async_op=True))
            if gather == "end":
                for c in range(nchunks):       # (one collective per chunk buffer; with nchunks == 1 this is THE one gather of the path)
                    src = tCs[g][c].cpu() if on_host else tCs[g][c]
                    pending[g].append(dist.gather(src, glists[g][c] if rank == 0 else None, dst=0, async_op=True))

        def fence():
            for g in (0, 1):
                for w in pending[g]:
                    w.wait()
                pending[g] = []
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(3):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kms = plan.time_execute(tCs[0][0].data_ptr(), tAs[0].data_ptr(), tB.data_ptr(), 2, 10) if nchunks == 1 and rank == 0 else None
        info = plan.info
        plan.close()
        return dt / steps * 1e3, int(pb[2]), kms, info, crow

    ms0, cbytes, kms, info, crow = run_variant(1, None)
    out["ms_per_step_compute_only"] = ms0
    out["value_compute_only"] = ops / (ms0 * 1e-3)
    out["shard_kernel_ms"] = kms
    out["gather_bytes_per_rank"] = cbytes
    if use_dist:
        ms1, _, _, _, _ = run_variant(1, "end")
        out["ms_per_step_one_gather"] = ms1
        out["value_one_gather"] = ops / (ms1 * 1e-3)
        tiles = (max_rows // 256) * (N // 256)
        nch = args.c4_chunks if args.c4_chunks > 0 else max(1, min(8, tiles // 256))
        while nch > 1 and (max_rows % nch or (max_rows // nch) % 256):
            nch -= 1
        out["chunks"] = nch
        if nch > 1:
            ms2, cb2, _, _, _ = run_variant(nch, "chunk")
            out["ms_per_step_chunked_gather"] = ms2
            out["value_chunked_gather"] = ops / (ms2 * 1e-3)
            out["chunk_bytes"] = cb2
    if world == 1 and kms and rank == 0:
        out["roofline"] = roofline_block("c4L", info, M, N, K, kms, [kms], capi, c_container_bytes(wl["c"]))
    return out


# ---------------------------------------------------------------------------------------------------------------- a rank
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args, argv))

    import torch
    import torch.distributed as dist
    from qublas_amd import capi

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)   # one rank per GPU on the driver's node; a rehearsal may stack ranks on one card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    on_host = use_dist and args.backend != "nccl"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    ctx = capi.Context(local)
    # ONE stream for the engine's launches and for torch: the engine's own.  A collective issued with it current waits (on the
    # device) for the GEMM that produced its input, and work.wait() makes the NEXT GEMM wait (on the device) for the collective
    # that still reads its output buffer — no host synchronisation inside a step.
    torch.cuda.set_stream(torch.cuda.ExternalStream(ctx.stream, device=dev))

    wls = workloads()
    wl = wls[args.workload]
    S = args.size
    M, N, K = (S, S, S) if args.workload in ("c3L", "c3T") else SHAPES[args.workload]
    plan, d = make_plan(ctx, wl, M, N, K)
    info = plan.info
    pb = info.packed_bytes
    tA = torch.empty(pb[0], dtype=torch.uint8, device=dev)
    tB = torch.empty(pb[1], dtype=torch.uint8, device=dev)
    tCs = [torch.empty(pb[2], dtype=torch.uint8, device=dev) for _ in range(2 if use_dist else 1)]   # the gather of step i overlaps the GEMM of step i+1
    tC = tCs[0]
    plan.fill(capi.OPERAND_A, 1 + 1000 * rank, 0, tA.data_ptr())   # rank r's rows of A: a distinct seed stream
    plan.fill(capi.OPERAND_B, 2, 0, tB.data_ptr())
    torch.cuda.synchronize()
    gather_lists = [None, None]
    if use_dist and rank == 0:
        gather_lists = [[torch.empty(pb[2], dtype=torch.uint8, device="cpu" if on_host else dev) for _ in range(world)] for _ in range(2)]
    pending = [None, None]
    state = {"i": 0}

    def step():
        b = (state["i"] & 1) if use_dist else 0
        state["i"] += 1
        if pending[b] is not None:
            pending[b].wait()          # device-side: the GEMM below is ordered behind the collective that reads buffer b
            pending[b] = None
        plan.execute(tCs[b].data_ptr(), tA.data_ptr(), tB.data_ptr())
        if use_dist:
            src = tCs[b].cpu() if on_host else tCs[b]
            pending[b] = dist.gather(src, gather_lists[b], dst=0, async_op=True)   # the ONE collective of the path

    def barrier():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, untimed: bring the card to its steady clock before the W warm-up steps (the first few hundred ms of MFMA work
    # after an idle period run on a ramping clock, tools/launch_gap.py)
    PREWARM = max(0, args.prewarm) if args.workload != "c3T" else min(max(0, args.prewarm), 20)
    for _ in range(PREWARM):
        plan.execute(tCs[0].data_ptr(), tA.data_ptr(), tB.data_ptr())
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ops_step = 2.0 * M * N * K * world
    value = ops_step * args.steps / dt

    out = None
    kms = singles = None
    if rank == 0:
        # dominant kernel, HIP events on the engine's own stream
        kms = plan.time_execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), 3, max(10, min(args.steps, 100)))
        # SURVEY.md §8-d asks for median and min beside the mean: 30 single launches, each bracketed by its own HIP events
        singles = sorted(plan.time_execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), 0, 1) for _ in range(30))
        roof = roofline_block(args.workload, info, M, N, K, kms, singles, capi, c_container_bytes(wl["c"]))
        mfma = roof["bound"] == "mfma"
        out = {"metric": "int-MAC/s (2*M*N*K/s) for Qgemul 4096^3 int<8,8>; % of MI355X int8 peak", "value": value,
               "unit": "int-op/s (2*M*N*K/s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "i8 limbs -> i32/i64" if mfma else "i32", "data": "synthetic",
               "config": {"workload": wl["text"], "baseline_config": wl["cfg"], "M_per_gpu": M, "N": N, "K": K, "class": "linear" if info.cls == 1 else "tree",
                          "sharding": f"rows of A/C over {world} rank(s), B replicated, one RCCL gather of C to rank 0" if world > 1 else "single GPU"},
               "prewarm_launches": PREWARM,
               "pct_of_int8_peak": 100.0 * value / (INT8_DENSE_PEAK_OPS * world),
               "roofline": roof}
        if use_dist:
            out["value_without_gather"] = 2.0 * M * N * K / (kms * 1e-3) * world   # the curve with and without the gather
            out["gather_bytes_per_step_per_rank"] = int(pb[2])
            out["rccl_world_size"] = dist.get_world_size()
            out["backend"] = args.backend
    extra = {}
    if not args.no_extra:
        # configuration 4 runs on every rank count (strong scaling); collective inside: every rank takes part
        try:
            c4 = c4_leg(args, ctx, wls, capi, torch, dist, dev, world, rank, use_dist, on_host)
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
            ea, eb_ = wl["a"], wl["b"]
            assert hb[0] == 4 and hb[1] == 4
            hA = torch.randint(ea.raw_min, ea.raw_max + 1, (M * K,), dtype=torch.int32, device=dev).view(torch.uint8)
            hB = torch.randint(eb_.raw_min, eb_.raw_max + 1, (K * N,), dtype=torch.int32, device=dev).view(torch.uint8)
            hC = torch.empty(M * N * hb[2], dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            lay = {}
            for nm, fn in (("pack_a_ms", lambda: plan.pack(capi.OPERAND_A, hA.data_ptr(), tA.data_ptr())),
                           ("pack_b_ms", lambda: plan.pack(capi.OPERAND_B, hB.data_ptr(), tB.data_ptr())),
                           ("unpack_c_ms", lambda: plan.unpack_c(tC.data_ptr(), hC.data_ptr()))):
                fn()
                ctx.sync()
                t1 = time.perf_counter()
                for _ in range(5):
                    fn()
                ctx.sync()
                lay[nm] = (time.perf_counter() - t1) / 5 * 1e3
            # the same C written by the kernel's own epilogue in the reference layout (no packed C, no unpack pass)
            plan.execute_host_c(hC.data_ptr(), tA.data_ptr(), tB.data_ptr())
            ctx.sync()
            t1 = time.perf_counter()
            for _ in range(10):
                plan.execute_host_c(hC.data_ptr(), tA.data_ptr(), tB.data_ptr())
            ctx.sync()
            lay["gemm_into_host_layout_c_ms"] = (time.perf_counter() - t1) / 10 * 1e3
            lay["epilogue_stores_host_layout"] = bool(plan.stores_host_c)
            lay["host_layout_call_ms"] = lay["pack_a_ms"] + lay["pack_b_ms"] + lay["gemm_into_host_layout_c_ms"]
            lay["host_layout_bytes"] = [int(hA.numel()), int(hB.numel()), int(hC.numel())]
            out["layout_steps"] = lay
            del hA, hB, hC
        except Exception as e:
            out["layout_steps"] = {"error": str(e)}
        for name, iters in (("c3T", 10), ("c3Td", 10), ("c2L", 200), ("c2T", 20), ("c5TF", 10), ("c5B", 10)):
            if name == args.workload:
                continue
            try:
                extra[name] = measure_config(ctx, name, wls, capi, torch, dev, iters)
            except Exception as e:  # an extra line must never take the primary line down
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and extra:
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(wl["ref"] or "c3L")
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "int-op/s (2*M*N*K/s)", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    plan.close()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
