"""bench.py as the driver runs it (run with -m gpu).  `python bench.py --gpus 2` must work as invoked: the parent touches no GPU
and starts the two ranks itself; on the one-GPU test box both ranks share the card and the packed bands travel over the host
transport (`--backend host`: RCCL refuses two ranks on one device; that run is the driver's).  The RCCL branch — the library's
communicator, the gather on its own stream, device-side ordering against the GEMMs, barrier and max — runs here with a world of
ONE rank.  Both legs of the JSON line are checked: the metric workload and extra.c4 (configuration 4, strong scaling, with and
without the gather).  bench.py imports no PyTorch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_common(j, world):
    assert j["n_gpus"] == world and j["unit"].startswith("int-op/s") and j["higher_is_better"] is True
    assert j["config"]["M_per_gpu"] == 4096 and j["config"]["K"] == 4096 and j["scaling"] == "weak"
    assert j["value"] > 1e12 and j["roofline"]["bound"] == "mfma" and 0.01 < j["roofline"]["frac"] < 0.12
    c4 = j["extra"]["c4"]
    assert "error" not in c4, c4
    assert c4["M"] == 16384 and c4["N"] == 16384 and c4["K"] == 4096 and c4["scaling"] == "strong"
    assert sum(c4["rows_per_rank"]) == 16384 and all(r % 256 == 0 for r in c4["rows_per_rank"])
    assert c4["world_size"] == world and c4["value_compute_only"] > 1e14
    return c4


def test_self_launcher_two_ranks_host_transport():
    j = run_bench("--gpus", "2", "--backend", "host", "--steps", "5", "--warmup", "2", "--prewarm", "20", "--no-cpu", "--c4-steps", "3")
    c4 = check_common(j, 2)
    assert j["rccl_world_size"] == 2 and j["backend"] == "host" and j["gather_bytes_per_step_per_rank"] == 4096 * 4096 * 4
    assert c4["rows_per_rank"] == [8192, 8192] and c4["rccl_world_size"] == 2
    assert c4["gather_bytes_per_rank"] == 8192 * 16384          # packed 1-byte C
    assert c4["value_one_gather"] > 0 and c4["chunks"] >= 2 and c4["value_chunked_gather"] > 0


def test_two_ranks_under_torch_distributed_run_as_the_driver_launches_them():
    """The driver's N > 1 command line: torch.distributed.run starts the ranks, its agent owns MASTER_PORT (the workers are clients
    of its TCPStore), rank 0 prints the line.  The RCCL id / the host channel's port travel through that store (qublas_amd/dist.py)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "host", "--steps", "3",
                        "--warmup", "1", "--prewarm", "10", "--no-cpu", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["backend"] == "host" and j["rccl_world_size"] == 2 and j["value"] > 1e12
    assert j["gather_bytes_per_step_per_rank"] == 4096 * 4096 * 4


def test_self_launcher_three_ranks_ragged_partition():
    """16384 rows over 3 ranks: 64 blocks of 256 rows do not divide — bands of 22 / 21 / 21 blocks; the gather pads to the largest."""
    j = run_bench("--gpus", "3", "--backend", "host", "--steps", "3", "--warmup", "1", "--prewarm", "10", "--no-cpu", "--c4-steps", "2")
    c4 = check_common(j, 3)
    assert c4["rows_per_rank"] == [5632, 5376, 5376] and c4["gather_bytes_per_rank"] == 5632 * 16384
    assert c4["value_one_gather"] > 0


def test_rccl_branch_with_one_rank():
    j = run_bench("--gpus", "1", "--force-dist", "--steps", "5", "--warmup", "2", "--prewarm", "20", "--no-cpu", "--c4-steps", "3")
    c4 = check_common(j, 1)
    assert j["rccl_world_size"] == 1 and j["backend"] == "rccl" and j["rccl_version"] > 20000     # ncclCommCount / ncclGetVersion of the library's communicator
    assert c4["rccl_world_size"] == 1 and c4["backend"] == "rccl" and c4["value_one_gather"] > 1e14
    assert c4["ms_per_step_events_one_gather"] > 0 and c4["ms_per_step_events_compute_only"] > 0


def test_bench_imports_no_torch():
    import subprocess
    code = ("import sys, runpy; sys.argv = ['bench.py', '--help']\n"
            "try:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_a_failing_rank_fails_the_launcher():
    """Exit code: a rank that dies (here: an impossible backend) must make `python bench.py --gpus 2` exit non-zero."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "no_such_backend", "--steps", "1",
                        "--warmup", "0", "--prewarm", "0", "--no-cpu", "--no-extra"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_single_gpu_line_carries_every_baseline_configuration():
    j = run_bench("--steps", "10", "--warmup", "3", "--prewarm", "50", "--c4-steps", "5", timeout=900)
    check_common(j, 1)
    assert j["cpu_baseline"]["value"] and j["cpu_baseline"]["kind"] in ("reference", "port")
    ex = j["extra"]
    for name, bound in (("c2L", "mfma"), ("c2T", "valu"), ("c3T", "valu"), ("c3Td", "valu"), ("w32T", "valu"), ("c5TF", "valu"), ("c5B", "valu"), ("c5L", "mfma"), ("reduce", "hbm"), ("reduceW", "hbm"),
                        ("long_k", "mfma")):
        assert "error" not in ex[name], ex[name]
        r = ex[name]["roofline"]
        assert r["bound"] == bound and r["kernel_ms"] > 0
        assert r["frac"] is None or 0 < r["frac"] <= 1.0, (name, r["frac"])          # no fraction above its roof
        assert (r["profile"] is None) == (r["profile_note"] is not None)             # a block either quotes a matching profile or says why not
        if bound == "valu" and r["profile"]:
            assert r["instr_per_mac"] > 1 and r["valu_busy_in_profile"] <= 1.05
    assert ex["c4"]["roofline"]["bound"] == "mfma" and ex["c4"]["roofline"]["frac"] > 0.3
    # the c4 leg's wall clock per step stays close to its kernel time by HIP events (round 2: 10-13 % apart, no prewarm)
    assert ex["c4"]["ms_per_step_compute_only"] < 1.06 * ex["c4"]["ms_per_step_events_compute_only"]
    assert ex["c3T"]["launches_timed"] >= 10 and ex["reduce"]["roofline"]["frac"] > 0.3
    assert j["ms_per_step_events"] > 0 and j["profile_key"]["engine_kernel"] == "mfma_i8_limb"
    assert j["roofline"]["traffic"] is None or "traffic_source" in j["roofline"]
