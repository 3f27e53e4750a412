"""Device-memory stability of the C-ABI: a simulation sweep calls Qgemul thousands of times with changing formats
and shapes (the reference is used that way: one Qgemul per candidate quantisation, QuBLAS.h README "bit-width search"),
so plans, contexts and the per-thread cache of qgemul_run must give back what they take."""
import numpy as np
import pytest
import torch

from qublas_amd import capi
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, TRN, Tags, TFComplexMul, lower, lower_reduce

pytestmark = pytest.mark.gpu

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
E43 = Qu(4, 3)
R63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL)
C5 = Qcomplex(R63, Qu(6, -3, True, RND.POS_INF, SAT.TCPL))


def _descs(i):
    s = 64 + 32 * (i % 5)
    return [
        lower(E43, E43, E43, s, s, 128, mul_args=Tags(9, 6), add_args=[Qu(19, 6)]),          # single-limb MFMA
        lower(E88, E88, Qu(23, 8), s, s, 256, mul_args=Tags(17, 16), add_args=[Qu(29, 16)]),  # limb MFMA (+ plane-mask partner)
        lower(E88, E88, E88, s, s, 100 + i % 7),                                               # 32-bit tree, any K
        lower(E88, E88, Qu(40, 3), s, 8, 64, add_args=[Qu(40, 3)]),                            # 64-bit tree
        lower_reduce(E88, 300 + i, 1024),                                                      # one-column kernel
        lower(C5, C5, C5, 32, 32, 64, mul_args=TFComplexMul()),                                # complex tree
    ]


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_plans_and_contexts_return_their_memory():
    def sweep(n):
        for i in range(n):
            with capi.Context(0) as ctx:
                for d in _descs(i):
                    plan = capi.Plan(ctx, d)
                    pb = plan.info.packed_bytes
                    bufs = [ctx.alloc(max(int(b), 256)) for b in pb]
                    plan.fill(capi.OPERAND_A, 1 + i, 0, bufs[0])
                    plan.fill(capi.OPERAND_B, 2 + i, 0, bufs[1])
                    plan.execute(bufs[2], bufs[0], bufs[1])
                    ctx.sync()
                    for b in bufs:
                        ctx.free(b)
                    plan.close()
    sweep(2)                      # code objects, stream pools
    before = _free_bytes()
    sweep(40)
    after = _free_bytes()
    assert before - after < (8 << 20), (before, after)


def test_run_cache_is_bounded_and_released(oracle):
    rng = np.random.default_rng(3)

    def calls(n):
        for i in range(n):
            for d in _descs(i)[:3]:
                ea = E43 if d.a[0].I == 4 else E88
                A = rng.integers(ea.raw_min, ea.raw_max + 1, d.M * d.K, dtype=np.int32)
                B = rng.integers(ea.raw_min, ea.raw_max + 1, d.K * d.N, dtype=np.int32)
                capi.run(d, np.zeros(d.M * d.N, dtype=np.int32), A, B)
    calls(2)
    held = _free_bytes()
    calls(60)
    assert held - _free_bytes() < (8 << 20)          # grow-only buffers stop growing once the largest shape was seen
    capi.run_release()
    assert _free_bytes() >= held                      # and go back on release
    d = _descs(0)[0]
    A = rng.integers(E43.raw_min, E43.raw_max + 1, d.M * d.K, dtype=np.int32)
    B = rng.integers(E43.raw_min, E43.raw_max + 1, d.K * d.N, dtype=np.int32)
    got = capi.run(d, np.zeros(d.M * d.N, dtype=np.int32), A, B)     # usable again after a release
    assert np.array_equal(got, oracle.gemm(d, A, B, E43, nthreads=4))


def test_fast_pack_paths_write_the_bytes_of_the_generic_kernels():
    """k_pack_limb32 (both axis orders, scalar and 16-byte loads) / k_unpack_c32 against k_pack / k_unpack_c
    (QG_OPT_GENERIC_LAYOUT): same packed bytes, same host C, padded leading dimensions and ragged shapes included."""
    import pack_dump
    fast = pack_dump.dump(0)
    generic = pack_dump.dump(capi.OPT_GENERIC_LAYOUT)
    assert fast == generic
    assert any(rec["kernel"] == "mfma_i8_limb" for rec in fast) and any(rec["kernel"] == "mfma_i8" for rec in fast)
