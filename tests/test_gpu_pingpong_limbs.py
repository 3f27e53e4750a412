"""k_mfma_ppl (qg_mfma_ppl.hip): the 3 x 3-limb kernel for problems with at least a tile per CU — 128x128 tiles, two wave
groups alternating on the matrix cores, one limb plane of A per phase, persistent workgroups.  Every case runs the whole matrix
through it AND through the lock-step kernel it replaced (QG_OPT_LOCKSTEP_TILES: k_mfma16<3,3> on the same packed layout) and
compares all outputs, then blocks against the oracle.  Full-range operands take the 3 x 3 kernel of the launch pair, half-range
operands (third limb planes empty) its 2 x 2 partner; shapes cover one / two / odd numbers of k-tiles and ragged M, N, K."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, WRP, Tags, lower

pytestmark = pytest.mark.gpu

E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)
KW = dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)])


def run_both(d, dist, seeds=(1, 2), limbs=3):
    outs = []
    for flags in (0, capi.OPT_LOCKSTEP_TILES):
        with capi.Context() as ctx:
            plan = capi.Plan(ctx, d, flags)
            info = plan.info
            assert capi.KERNEL_NAMES[info.kernel] == "mfma_i8_limb" and list(info.limbs)[:2] == [limbs, limbs]
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


def check(oracle, d, ec, got, rows, cols, dist, seeds=(1, 2), e=E88):
    A = oracle.fill(e, d.M * d.K, seeds[0], dist)
    B = oracle.fill(e, d.K * d.N, seeds[1], dist)
    cdt = oracle.host_dtype(ec)
    exp = np.zeros(d.M * d.N, dtype=cdt)
    oracle.gemm(d, A, B, ec, rows=rows, cols=cols, nthreads=16, out=exp)
    sl = (slice(cols[0], cols[1]), slice(rows[0], rows[1]))
    assert np.array_equal(got.view(cdt).reshape(d.N, d.M)[sl], exp.reshape(d.N, d.M)[sl])


@pytest.mark.parametrize("dist", [0, 1])   # 0: full range (3 x 3 limbs); 1: |x| < 2^8 (third planes empty: the 2 x 2 partner)
@pytest.mark.parametrize("M,N,K", [
    (2048, 2048, 64),      # one k-tile: every refill of the loop is a clamped one
    (2048, 2048, 128),     # two k-tiles
    (2048, 2048, 320),     # odd number of k-tiles (buffer parity)
    (2048, 2048, 1000),    # ragged K
    (2100, 2000, 448),     # ragged M and N: 17 x 16 tiles
    (4096, 1024, 2048),    # 32 x 8 tiles
])
def test_shapes_against_lockstep_kernel_and_oracle(oracle, M, N, K, dist):
    ec = Qu(23, 8)
    d = lower(E88, E88, ec, M, N, K, **KW)
    pp, ls = run_both(d, dist)
    assert np.array_equal(pp, ls)
    check(oracle, d, ec, pp, rows=(0, 8), cols=(0, min(N, 256)), dist=dist)
    check(oracle, d, ec, pp, rows=(M - 70, M - 60), cols=(N - 200, N), dist=dist)
    assert np.count_nonzero(pp.view(np.int32)) > 0.9 * M * N


@pytest.mark.parametrize("ec", [
    Qu(23, 8),                                 # 4 bytes, truncation + SAT::TCPL: the shift-and-clamp epilogue
    Qu(14, 16, True, TRN.TCPL, SAT.TCPL),      # no shift at all, saturates
    Qu(12, 8, True, RND.CONV, SAT.SMGN),       # general routine
    Qu(20, 4, False, RND.INF, SAT.ZERO),       # unsigned
    Qu(18, 10, True, TRN.SMGN, WRP.TCPL),
    Qu(29, 16),                                # 8-byte container
    Qu(40, 6, True, RND.NEG_INF, SAT.TCPL),    # 8 bytes, general routine
])
def test_every_container_and_mode(oracle, ec):
    d = lower(E88, E88, ec, 2048, 2048, 512, **KW)
    pp, ls = run_both(d, 0)
    assert np.array_equal(pp, ls)
    check(oracle, d, ec, pp, rows=(1000, 1016), cols=(1024, 1280), dist=0)


E77 = Qu(7, 7)      # 15 storage bits: two limbs, not Karatsuba-eligible (more than 12 value + sign bits)


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 64), (2048, 2048, 192), (2100, 2000, 1000), (4096, 1024, 512)])
def test_two_limb_operands_on_the_two_group_kernel(oracle, M, N, K):
    """2 x 2 limbs on two-plane storage (k_mfma_ppl22): 4-byte and 8-byte C, against the lock-step kernel and the oracle."""
    for ec in (Qu(20, 8), Qu(28, 14, True, RND.CONV, SAT.SMGN)):
        d = lower(E77, E77, ec, M, N, K, mul_args=Tags(15, 14), add_args=[Qu(27, 14)])
        pp, ls = run_both(d, 0, limbs=2)
        assert np.array_equal(pp, ls)
        check(oracle, d, ec, pp, rows=(0, 8), cols=(0, 256), dist=0, e=E77)
        check(oracle, d, ec, pp, rows=(M - 9, M - 1), cols=(N - 200, N), dist=0, e=E77)


def test_karatsuba_eligible_two_limb_operands_keep_their_kernel(oracle):
    """int<6,5> (12 value + sign bits): three products on the lock-step Karatsuba kernel, whatever the tile count."""
    e = Qu(6, 5)
    d = lower(e, e, Qu(20, 8), 2048, 2048, 256, mul_args=Tags(13, 10), add_args=[Qu(25, 10)])
    a, b = run_both(d, 0, limbs=2)
    assert np.array_equal(a, b)
    check(oracle, d, Qu(20, 8), a, rows=(100, 108), cols=(0, 256), dist=0, e=e)
