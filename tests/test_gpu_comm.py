"""The library-owned RCCL communicator (include/qgemul.h, qgemul_comm_*; run with -m gpu).  The test box has ONE GPU and RCCL
refuses two ranks on one device, so this runs a world of one rank through the C-ABI: id, ncclCommInitRank, ncclCommCount /
ncclGetVersion through qgemul_comm_info, the gather call (root's own band: a device copy on the communicator's stream, ordered
behind the GEMM by events), fence, barrier and max.  The multi-rank send / receive legs run in the driver's 8-GPU bench;
their partition / reassembly logic is covered with several ranks over the host transport in tests/test_dist_gloo.py and
tests/test_gpu_dist.py."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, Tags, lower

pytestmark = pytest.mark.gpu


def test_world_of_one_rank(oracle):
    e, ec = Qu(4, 3), Qu(16, 3)
    M, N, K = 512, 256, 256
    d = lower(e, e, ec, M, N, K, mul_args=Tags(9, 6), add_args=[Qu(19, 6)])
    with capi.Context(0) as ctx:
        uid = capi.Comm.unique_id()
        assert len(uid) == 128 and any(uid)
        comm = capi.Comm(ctx, 1, 0, uid)
        n, r, ver = comm.info()
        assert (n, r) == (1, 0) and ver > 20000, ver          # ncclGetVersion: 2.x.y -> 2xxyy
        p = capi.Plan(ctx, d)
        b = p.info.packed_bytes
        pa, pb, pc, land = ctx.alloc(b[0]), ctx.alloc(b[1]), ctx.alloc(b[2]), ctx.alloc(b[2])
        hc = ctx.alloc(M * N * 4)
        p.fill(capi.OPERAND_A, 1, 0, pa)
        p.fill(capi.OPERAND_B, 2, 0, pb)
        for i in range(3):                                     # (repeated: the events are re-recorded per call; alternating slots)
            p.execute(pc, pa, pb)
            comm.gather(pc, b[2], [land], [b[2]], 0, slot=i & 1)   # the band lands in the root's buffer
            comm.fence(i & 1)                                  # the context's stream waits for that generation
            p.unpack_c(land, hc)
        got = np.zeros(M * N, np.int32)
        ctx.d2h(got, hc)
        comm.barrier()
        assert comm.max_f64(3.5) == 3.5
        comm.sync()
        comm.close()
        p.close()
        for x in (pa, pb, pc, land, hc):
            ctx.free(x)
    A = oracle.fill(e, M * K, 1)
    B = oracle.fill(e, K * N, 2)
    assert np.array_equal(got, oracle.gemm(d, A, B, ec, nthreads=8))
    # argument errors do not reach RCCL
    assert capi.lib().qgemul_comm_fence(None, 0) == capi.QG_EINVAL
