"""Row-sharded Qgemul with the REAL engine as every rank's compute (run with -m gpu): two processes share the one GPU of the
test box, rendezvous over gloo on 127.0.0.1, each computes its row block through the C-ABI, rank 0 gathers and compares the
assembled C with the oracle.  (The RCCL transport itself needs one GPU per rank: that run is the driver's 8-GPU bench.)"""
import multiprocessing as mp
import os

import pytest

from test_dist_gloo import CASES, _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import qoracle
    from qublas_amd.desc import lower
    from qublas_amd.dist import qgemul_row_sharded
    c = CASES[name]
    M, N, K = c["M"], c["N"], c["K"]
    A = qoracle.fill(c["ea"], M * K, 1, 1)
    B = qoracle.fill(c["eb"], K * N, 2, 1)
    out = qgemul_row_sharded(A, B, c["ea"], c["eb"], c["ec"], M, N, K, align=128, **c["kw"])   # default compute: the HIP engine
    if rank == 0:
        d = lower(c["ea"], c["eb"], c["ec"], M, N, K, **c["kw"])
        q.put(out.tobytes() == qoracle.gemm(d, A, B, c["ec"]).tobytes())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_row_sharded_with_the_engine(name, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
