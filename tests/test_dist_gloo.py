"""Row-shard + single-gather orchestration (qublas_amd/dist.py) on CPU: world_size 2 and 3 over the TCP control channel
(qublas_amd.dist.HostChannel; round 2 used torch.distributed's gloo backend here, which the product path no longer imports), the
per-rank engine call replaced by the CPU oracle (tests only).  Checks the row partition, the strided / contiguous shard views
for both A orientations, unequal shards and the reassembly of the column-major C against a single-process oracle run."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qublas_amd.desc import Qcomplex, Qu, RND, SAT, Tags, TFComplexMul, lower  # noqa: E402
from qublas_amd.dist import qgemul_row_sharded, row_partition  # noqa: E402


def test_row_partition():
    assert row_partition(16384, 8, 128) == [(i * 2048, 2048) for i in range(8)]
    p = row_partition(1000, 3, 128)
    assert sum(r for _, r in p) == 1000 and p[0][0] == 0
    assert all(r0 % 128 == 0 for r0, _ in p)
    assert [r for _, r in row_partition(5, 8, 1)] == [1, 1, 1, 1, 1, 0, 0, 0]
    assert row_partition(100, 1, 128) == [(0, 100)]
    for M in (1, 127, 128, 129, 4096, 5000):
        for w in (1, 2, 3, 8):
            q = row_partition(M, w, 128)
            assert q[0][0] == 0 and sum(r for _, r in q) == M
            assert all(q[i][0] + q[i][1] == q[i + 1][0] or q[i + 1][1] == 0 for i in range(w - 1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = {
    "real_nn": dict(ea=Qu(4, 3), eb=Qu(4, 3), ec=Qu(16, 3), M=300, N=37, K=64, kw=dict(mul_args=Tags(9, 6), add_args=[Qu(19, 6)])),
    "real_tn_tree": dict(ea=Qu(8, 8, True, 5, SAT.ZERO), eb=Qu(8, 8, True, 5, SAT.ZERO), ec=Qu(8, 8, True, 5, SAT.ZERO), M=130,
                         N=20, K=37, kw=dict(transposed_a=True)),
    "complex_tf": dict(ea=Qcomplex(Qu(6, 3, True, RND.POS_INF), Qu(6, -3, True, RND.POS_INF)),
                       eb=Qcomplex(Qu(6, 3, True, RND.POS_INF), Qu(6, -3, True, RND.POS_INF)),
                       ec=Qcomplex(Qu(18, 6), Qu(18, 6)), M=257, N=9, K=32, kw=dict(mul_args=TFComplexMul())),
}


def _worker(rank, world, port, name, q):
    from oracle import qoracle
    from qublas_amd.dist import HostChannel
    ch = HostChannel(rank, world, "127.0.0.1", port)
    c = CASES[name]
    M, N, K = c["M"], c["N"], c["K"]
    A = qoracle.fill(c["ea"], M * K, 1, 1)
    B = qoracle.fill(c["eb"], K * N, 2, 1)

    def compute(d, a_view, b, ec, lda, ldb):
        return qoracle.gemm(d, a_view, b, ec, lda=lda, ldb=ldb)

    out = qgemul_row_sharded(A, B, c["ea"], c["eb"], c["ec"], M, N, K, align=128, compute=compute, channel=ch, **c["kw"])
    if rank == 0:
        d = lower(c["ea"], c["eb"], c["ec"], M, N, K, **c["kw"])
        exp = qoracle.gemm(d, A, B, c["ec"])
        q.put(out.tobytes() == exp.tobytes())
    else:
        assert out is None
    ch.barrier()
    assert ch.max_f64(float(rank)) == float(world - 1)
    ch.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_row_sharded_gather_gloo(name, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_the_engine_path_imports_no_torch():
    """qublas_amd.dist and qublas_amd.capi — the product-side Python plumbing of the multi-GPU path — do not import torch"""
    import subprocess
    code = "import sys; import qublas_amd.dist, qublas_amd.capi; assert 'torch' not in sys.modules, 'torch was imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
