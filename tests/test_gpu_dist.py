"""Row-sharded Qgemul with the REAL engine as every rank's compute (run with -m gpu): two or three processes share the one GPU
of the test box; each computes its row block through the C-ABI and the packed bands go to rank 0 over the HostTransport (host
memory + TCP: RCCL refuses two ranks on one device), rank 0 unpacks and compares the assembled C with the oracle.  The RCCL
transport itself runs here with one rank (tests/test_gpu_comm.py, and the last test below) and with one GPU per rank in the
driver's 8-GPU bench."""
import multiprocessing as mp
import os

import pytest

from test_dist_gloo import CASES, _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, q):
    from oracle import qoracle
    from qublas_amd import capi
    from qublas_amd.desc import lower
    from qublas_amd.dist import HostChannel, HostTransport, qgemul_row_sharded
    ch = HostChannel(rank, world, "127.0.0.1", port)
    c = CASES[name]
    M, N, K = c["M"], c["N"], c["K"]
    A = qoracle.fill(c["ea"], M * K, 1, 1)
    B = qoracle.fill(c["eb"], K * N, 2, 1)
    with capi.Context(0) as ctx:
        out = qgemul_row_sharded(A, B, c["ea"], c["eb"], c["ec"], M, N, K, align=128, transport=HostTransport(ctx, ch), **c["kw"])   # the HIP engine
    if rank == 0:
        d = lower(c["ea"], c["eb"], c["ec"], M, N, K, **c["kw"])
        q.put(out.tobytes() == qoracle.gemm(d, A, B, c["ec"]).tobytes())
    else:
        assert out is None
    ch.barrier()
    ch.close()


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


@pytest.mark.parametrize("name", sorted(CASES))
def test_row_sharded_over_the_rccl_transport_one_rank(oracle, name):
    """the same call over the library's RCCL communicator with a world of one rank (communicator, gather call, fence, barrier)"""
    from qublas_amd import capi
    from qublas_amd.desc import lower
    from qublas_amd.dist import RcclTransport, qgemul_row_sharded
    c = CASES[name]
    M, N, K = c["M"], c["N"], c["K"]
    A = oracle.fill(c["ea"], M * K, 1, 1)
    B = oracle.fill(c["eb"], K * N, 2, 1)
    with capi.Context(0) as ctx:
        comm = capi.Comm(ctx, 1, 0, capi.Comm.unique_id())
        out = qgemul_row_sharded(A, B, c["ea"], c["eb"], c["ec"], M, N, K, align=128, transport=RcclTransport(ctx, comm), **c["kw"])
        comm.close()
    d = lower(c["ea"], c["eb"], c["ec"], M, N, K, **c["kw"])
    assert out.tobytes() == oracle.gemm(d, A, B, c["ec"]).tobytes()
