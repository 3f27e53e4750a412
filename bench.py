#!/usr/bin/env python3
"""bench.py — throughput of the Qgemul hot path on MI355X, per the driver contract.

A "step" is one Qgemul over device-resident, already-packed synthetic fixed-point operands
(raw integers uniform over the full representable range, the distribution Qu::fill() draws,
generated on the device by the same counter-based generator the CPU oracle implements).

  N = 1 : BASELINE.json's metric configuration — 4096^3 Qgemul, int<8,8> signed operands
          (configs[2]).  Primary line: the linear class (QgemulMulArgs<intBits<17>,fracBits<16>>,
          QgemulAddArgs<Qu<intBits<29>,fracBits<16>>>, C = Qu<intBits<23>,fracBits<8>>) on the
          int8-limb MFMA kernel; the default-tag tree-class figure for the same operands and the
          configuration-2/4 int<4,3> single-limb figure ride along in "extra".
  N > 1 : the same per-GPU problem row-sharded over M (rank r owns rows [r*4096,(r+1)*4096) of a
          (4096*N) x 4096 x 4096 product, B replicated), one RCCL gather of the packed C shards to
          rank 0 per step — the only collective on the path.  scaling = "weak".

One JSON line on rank 0.  `value` = 2*M*N*K*steps / time over the whole job (all ranks, gather
included).  `roofline` prices the dominant kernel against the dense int8 MFMA peak with the
ALGORITHMIC op count (2*M*N*K, not the 9 limb products the kernel issues).  `cpu_baseline` times
the reference's own primitives (oracle/_ref/ref_bench, built from /root/reference in the build
container) on a bounded block of the same workload, one process per host core.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_DENSE_PEAK_OPS = 5.0e15   # MI355X dense int8 MFMA, /opt/skills/guides/MI355X_MICROARCH.md (I8 = 2x BF16 ~2.5 PF)


def workloads():
    from qublas_amd.desc import Qu, SAT, TRN, Tags, lower
    e88z = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
    e43 = Qu(4, 3)
    return {
        # name: (lower kwargs, description)
        "c3L": dict(a=e88z, b=e88z, c=Qu(23, 8), mul=Tags(17, 16), add=[Qu(29, 16)],
                    text="4096^3 Qgemul int<8,8> signed, linear class (MulArgs int17/frac16, AddArgs Qu<29,16>, C Qu<23,8>), 3x3 int8-limb MFMA"),
        "c3T": dict(a=e88z, b=e88z, c=e88z, mul=None, add=None,
                    text="4096^3 Qgemul int<8,8> signed TRN::TCPL/SAT::ZERO, default tags (tree class), exact tree on VALU"),
        "c2L": dict(a=e43, b=e43, c=e43, mul=Tags(9, 6), add=[Qu(21, 6)],
                    text="Qgemul int<4,3> signed, linear class (MulArgs int9/frac6, AddArgs Qu<21,6>), single-limb int8 MFMA"),
    }


def make_plan(ctx, wl, M, N, K):
    from qublas_amd import capi
    from qublas_amd.desc import lower
    d = lower(wl["a"], wl["b"], wl["c"], M, N, K, mul_args=wl["mul"], add_args=wl["add"])
    return capi.Plan(ctx, d), d


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
    wl = workloads()["c3L" if variant == "c3L" else "c3T" if variant == "c3T" else "c2L"]
    from qublas_amd.desc import lower
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
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary, if any."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3L", choices=["c3L", "c3T", "c2L"])
    ap.add_argument("--size", type=int, default=4096, help="M=N=K per GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the script on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the collective path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)")
    ap.add_argument("--prewarm", type=int, default=300, help="untimed launches before the warm-up steps (lets the GPU clock settle); 0 for counter passes")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qublas_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)   # one rank per GPU on the driver's node; a rehearsal may stack ranks on one card
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    wls = workloads()
    wl = wls[args.workload]
    S = args.size
    M = N = K = S
    ctx = capi.Context(local)
    plan, d = make_plan(ctx, wl, M, N, K)
    info = plan.info
    pb = info.packed_bytes
    dev = torch.device("cuda", local)
    tA = torch.empty(pb[0], dtype=torch.uint8, device=dev)
    tB = torch.empty(pb[1], dtype=torch.uint8, device=dev)
    # two C buffers: the gather of step i overlaps the GEMM of step i+1
    tCs = [torch.empty(pb[2], dtype=torch.uint8, device=dev) for _ in range(2 if use_dist else 1)]
    tC = tCs[0]
    # rank r's shard of A: rows [r*M, (r+1)*M) of the (world*M) x K operand -> distinct seed stream
    plan.fill(capi.OPERAND_A, 1 + 1000 * rank, 0, tA.data_ptr())
    plan.fill(capi.OPERAND_B, 2, 0, tB.data_ptr())
    ctx.sync()
    gather_lists = [None, None]
    on_host = use_dist and args.backend != "nccl"
    if use_dist and rank == 0:
        gather_lists = [[torch.empty(pb[2], dtype=torch.uint8, device="cpu" if on_host else dev) for _ in range(world)]
                        for _ in range(2)]
    pending = [None, None]
    state = {"i": 0}

    def step():
        b = (state["i"] & 1) if use_dist else 0
        state["i"] += 1
        if use_dist and pending[b] is not None:
            pending[b].wait()          # buffer b was handed to the collective two steps ago
            if not on_host:
                # with RCCL wait() only orders torch's current stream; the engine launches on its own
                # stream, so block the host until the collective has really finished reading buffer b
                torch.cuda.current_stream().synchronize()
            pending[b] = None
        plan.execute(tCs[b].data_ptr(), tA.data_ptr(), tB.data_ptr())
        if use_dist:
            ctx.sync()                 # the engine launches on its own stream; the collective runs on torch's
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

    # setup, untimed: bring the card to its steady clock before the W warm-up steps (the first few hundred ms of
    # MFMA work after an idle period run on a ramping clock: 0.478 vs 0.450 ms per launch, tools/launch_gap.py)
    PREWARM = max(0, args.prewarm) if args.workload != "c3T" else min(max(0, args.prewarm), 20)
    for _ in range(PREWARM):
        plan.execute(tCs[0].data_ptr(), tA.data_ptr(), tB.data_ptr())
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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ops_step = 2.0 * M * N * K * world
    value = ops_step * args.steps / dt

    out = None
    if rank == 0:
        # dominant kernel, HIP events on the engine's own stream
        kms = plan.time_execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), 3, max(10, min(args.steps, 100)))
        # SURVEY.md §8-d asks for median and min beside the mean: 30 single launches, each bracketed by its own HIP events
        singles = sorted(plan.time_execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), 0, 1) for _ in range(30))
        achieved = 2.0 * M * N * K / (kms * 1e-3)
        bound = "mfma" if info.kernel in (1, 2) else "valu"
        roof = {"bound": "mfma", "achieved": achieved / 1e12, "peak": INT8_DENSE_PEAK_OPS / 1e12, "unit": "TOP/s (int8-equivalent 2*M*N*K)",
                "frac": achieved / INT8_DENSE_PEAK_OPS, "traffic": load_traffic(args.workload),
                "kernel": capi.KERNEL_NAMES[info.kernel], "kernel_ms": kms,
                "kernel_ms_median": singles[len(singles) // 2], "kernel_ms_min": singles[0],
                "limbs": [info.limbs[0], info.limbs[1]],
                "mfma_issue_frac": achieved * max(1, info.limbs[0] * info.limbs[1]) / INT8_DENSE_PEAK_OPS if bound == "mfma" else None,
                "algorithmic_bytes": int((M * K + K * N) * max(1, (info.in_bits[0] + 7) // 8) + M * N * (pb[2] // (M * N) if M * N else 0))}
        if bound != "mfma":
            roof["note"] = "tree class runs on the vector ALUs (no MFMA); fraction is still quoted against the int8 MFMA peak as the metric demands"
        out = {"metric": "int-MAC/s (2*M*N*K/s) for Qgemul 4096^3 int<8,8>; % of MI355X int8 peak", "value": value,
               "unit": "int-op/s (2*M*N*K/s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "i8 limbs -> i32/i64" if bound == "mfma" else "i64", "data": "synthetic",
               "config": {"workload": wl["text"], "M_per_gpu": M, "N": N, "K": K, "class": "linear" if info.cls == 1 else "tree",
                          "sharding": f"rows of A/C over {world} rank(s), B replicated, one RCCL gather of C to rank 0" if world > 1 else "single GPU"},
               "prewarm_launches": PREWARM,
               "pct_of_int8_peak": 100.0 * value / (INT8_DENSE_PEAK_OPS * world),
               "roofline": roof}
        if use_dist:
            # the curve with and without the gather
            out["value_without_gather"] = achieved * world
            out["gather_bytes_per_step_per_rank"] = int(pb[2])
        if world == 1 and not args.no_extra:
            # the layout steps either side of the hot path, timed separately (never part of `value`)
            try:
                hb = info.host_elem_bytes
                hA = torch.zeros(M * K * hb[0], dtype=torch.uint8, device=dev)
                hB = torch.zeros(K * N * hb[1], dtype=torch.uint8, device=dev)
                hC = torch.empty(M * N * hb[2], dtype=torch.uint8, device=dev)
                lay = {}
                for nm, fn in (("pack_a_ms", lambda: plan.pack(capi.OPERAND_A, hA.data_ptr(), tA.data_ptr())),
                               ("pack_b_ms", lambda: plan.pack(capi.OPERAND_B, hB.data_ptr(), tB.data_ptr())),
                               ("unpack_c_ms", lambda: plan.unpack_c(tC.data_ptr(), hC.data_ptr()))):
                    fn(); ctx.sync()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        fn()
                    ctx.sync()
                    lay[nm] = (time.perf_counter() - t1) / 5 * 1e3
                lay["host_layout_bytes"] = [int(hA.numel()), int(hB.numel()), int(hC.numel())]
                # zeros were packed over the synthetic operands: regenerate them for the extras below
                plan.fill(capi.OPERAND_A, 1, 0, tA.data_ptr())
                plan.fill(capi.OPERAND_B, 2, 0, tB.data_ptr())
                ctx.sync()
                out["layout_steps"] = lay
                del hA, hB, hC
            except Exception as e:
                out["layout_steps"] = {"error": str(e)}
        if world == 1 and not args.no_extra:
            extra = {}
            for name, (m2, n2, k2) in (("c3T", (S, S, S)), ("c2L", (8192, 8192, 4096))):
                if name == args.workload:
                    continue
                try:
                    p2, _ = make_plan(ctx, wls[name], m2, n2, k2)
                    b2 = p2.info.packed_bytes
                    xa = torch.empty(b2[0], dtype=torch.uint8, device=dev)
                    xb = torch.empty(b2[1], dtype=torch.uint8, device=dev)
                    xc = torch.empty(b2[2], dtype=torch.uint8, device=dev)
                    p2.fill(capi.OPERAND_A, 1, 0, xa.data_ptr())
                    p2.fill(capi.OPERAND_B, 2, 0, xb.data_ptr())
                    it = 3 if name == "c3T" else 20
                    ms = p2.time_execute(xc.data_ptr(), xa.data_ptr(), xb.data_ptr(), 1, it)
                    ops = 2.0 * m2 * n2 * k2
                    extra[name] = {"workload": wls[name]["text"], "M": m2, "N": n2, "K": k2, "kernel": capi.KERNEL_NAMES[p2.info.kernel],
                                   "kernel_ms": ms, "value": ops / (ms * 1e-3), "pct_of_int8_peak": 100.0 * ops / (ms * 1e-3) / INT8_DENSE_PEAK_OPS}
                    p2.close()
                    del xa, xb, xc
                except Exception as e:  # an extra line must never take the primary line down
                    extra[name] = {"error": str(e)}
            out["extra"] = extra
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "int-op/s (2*M*N*K/s)", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    plan.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
