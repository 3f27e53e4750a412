"""bench.py's launcher logic on the CPU: `--gpus N` without WORLD_SIZE in the environment makes the process a PARENT that starts
N children (and imports nothing that could touch a GPU before it does); with WORLD_SIZE set it is a rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_imports_no_gpu_module_before_spawning():
    code = (
        "import sys, types\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "started = []\n"
        "class P:\n"
        "    returncode = 0\n"
        "    def __init__(self, cmd, env=None, **kw): started.append((cmd, env)); self.i = len(started) - 1\n"
        "    def communicate(self, timeout=None): return ('{\"ok\": 1}\\n', None)\n"
        "    def wait(self, timeout=None): return 0\n"
        "    def poll(self): return 0\n"
        "bench.subprocess.Popen = P\n"
        "rc = bench.launch_ranks(bench.parse_args(['--gpus', '3']), ['--gpus', '3'])\n"
        "assert rc == 0 and len(started) == 3\n"
        "for r, (cmd, env) in enumerate(started):\n"
        "    assert env['RANK'] == str(r) and env['LOCAL_RANK'] == str(r) and env['WORLD_SIZE'] == '3' and env['MASTER_ADDR'] == '127.0.0.1'\n"
        "    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' or 'HSA_ENABLE_IPC_MODE_LEGACY' in env\n"
        "assert 'torch' not in sys.modules and 'qublas_amd.capi' not in sys.modules\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_nonzero_child_exit_fails_the_parent():
    code = (
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "class P:\n"
        "    n = 0\n"
        "    def __init__(self, cmd, env=None, **kw): self.returncode = 0 if P.n == 0 else 3; P.n += 1\n"
        "    def communicate(self, timeout=None): return ('{\"ok\": 1}\\n', None)\n"
        "    def wait(self, timeout=None): return self.returncode\n"
        "    def poll(self): return self.returncode\n"
        "bench.subprocess.Popen = P\n"
        "sys.exit(bench.launch_ranks(bench.parse_args(['--gpus', '2']), ['--gpus', '2']))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "{" not in r.stdout
