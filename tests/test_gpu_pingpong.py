"""k_mfma_pp (qg_mfma_pp.hip): the single-limb kernel for large problems — 256x256 tiles, 128-byte k-tiles, two wave groups
alternating on the matrix cores.  Every case runs the whole matrix through it AND through the lock-step kernel it replaced
(QG_OPT_LOCKSTEP_TILES, k_mfma16 on 64-byte k-tiles: different packing, tiles and pipeline) and compares all outputs, then a
block against the oracle.  Shapes cover one and two k-tiles (the prologue's clamped refills), odd numbers of k-tiles, ragged
M / N / K (padding inside the packed tiles), every C container and every QuMode x OfMode of the one conversion."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, WRP, Tags, lower

pytestmark = pytest.mark.gpu

E43 = Qu(4, 3)
KW = dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)])


def run_both(d, seeds=(1, 2), dist=0):
    outs = []
    for flags in (0, capi.OPT_LOCKSTEP_TILES):
        with capi.Context() as ctx:
            plan = capi.Plan(ctx, d, flags)
            info = plan.info
            assert capi.KERNEL_NAMES[info.kernel] == "mfma_i8"
            pb = info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            nbytes = d.M * d.N * info.host_elem_bytes[2]
            dC = ctx.alloc(nbytes)
            plan.fill(capi.OPERAND_A, seeds[0], dist, pA)
            plan.fill(capi.OPERAND_B, seeds[1], dist, pB)
            plan.execute(pC, pA, pB)
            plan.unpack_c(pC, dC)
            out = np.zeros(nbytes, np.uint8)
            ctx.d2h(out, dC)
            for p in (pA, pB, pC, dC):
                ctx.free(p)
            plan.close()
        outs.append(out)
    return outs


def check(oracle, d, ec, got, rows, cols, seeds=(1, 2), dist=0):
    A = oracle.fill(E43, d.M * d.K, seeds[0], dist)
    B = oracle.fill(E43, d.K * d.N, seeds[1], dist)
    cdt = oracle.host_dtype(ec)
    exp = np.zeros(d.M * d.N, dtype=cdt)
    oracle.gemm(d, A, B, ec, rows=rows, cols=cols, nthreads=16, out=exp)
    sl = (slice(cols[0], cols[1]), slice(rows[0], rows[1]))
    assert np.array_equal(got.view(cdt).reshape(d.N, d.M)[sl], exp.reshape(d.N, d.M)[sl])


@pytest.mark.parametrize("M,N,K", [
    (4096, 4096, 128),     # one k-tile: every refill of the loop is a clamped one
    (4096, 4096, 256),     # two k-tiles
    (4096, 4096, 384),     # odd number of k-tiles (buffer parity)
    (4096, 4096, 1000),    # ragged K: zero padding inside the last k-tile
    (3900, 4300, 640),     # ragged M and N: 16 x 17 tiles, padded rows / columns never stored to the host
    (8192, 2048, 2048),    # 32 x 8 tiles
])
def test_shapes_against_lockstep_kernel_and_oracle(oracle, M, N, K):
    ec = Qu(16, 3)
    d = lower(E43, E43, ec, M, N, K, **KW)
    pp, ls = run_both(d)
    assert np.array_equal(pp, ls)
    check(oracle, d, ec, pp, rows=(0, 8), cols=(0, min(N, 512)))
    check(oracle, d, ec, pp, rows=(M - 70, M - 60), cols=(N - 300, N))
    assert np.count_nonzero(pp.view(np.int32)) > 0.9 * M * N


@pytest.mark.parametrize("ec", [
    Qu(4, 3),                                  # 1-byte container (configuration 4's C)
    Qu(4, 3, True, RND.CONV, SAT.SMGN),
    Qu(12, 3),                                 # 2 bytes, truncation + SAT::TCPL (the shift-and-clamp epilogue)
    Qu(9, 6),                                  # 2 bytes, no shift at all
    Qu(9, 3, True, RND.INF, SAT.ZERO),         # 2 bytes, general routine
    Qu(6, 5, False, TRN.SMGN, WRP.TCPL),       # unsigned, wrapping
    Qu(16, 6, True, RND.NEG_INF, SAT.TCPL),    # 4 bytes, no rounding shift
    Qu(20, 2, True, RND.POS_INF, WRP.TCPL),
    Qu(30, 6),                                 # 8-byte container through the 64-bit pass (C beyond 31 value bits)
    Qu(5, 12, True, RND.ZERO, SAT.TCPL),       # left shift by 6 into a narrow C
])
def test_every_container_and_mode(oracle, ec):
    d = lower(E43, E43, ec, 4096, 4096, 512, **KW)
    pp, ls = run_both(d, dist=1)
    assert np.array_equal(pp, ls)
    check(oracle, d, ec, pp, rows=(1000, 1016), cols=(2048, 2304), dist=1)


def test_config4_shard_is_bit_identical_on_both_kernels(oracle):
    """One 2048-row shard of configuration 4 (1 of 8 GPUs): 8 x 64 tiles, K = 4096."""
    ec = E43
    d = lower(E43, E43, ec, 2048, 16384, 4096, **KW)
    pp, ls = run_both(d)
    assert np.array_equal(pp, ls)
    check(oracle, d, ec, pp, rows=(2040, 2048), cols=(16000, 16384))
