"""qgemul_execute_host_c: the result in the REFERENCE layout on the device.  On the two-group MFMA kernels with a 4- or 8-byte
C element the epilogue stores its runs of rows straight into that layout (no packed C, no unpack pass); every other plan runs
execute + unpack behind the same call.  Either way the bytes must equal qgemul_execute followed by qgemul_unpack_c — ragged M
and N (rows / columns of the last tiles are masked), padded and odd leading dimensions (vector and scalar stores) — AND a block
of rows of the result must equal the CPU oracle on the same synthetic operands (comparing the engine with itself
proves no parity by itself)."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, Tags, lower

pytestmark = pytest.mark.gpu

E43 = Qu(4, 3)
E88 = Qu(8, 8, True, TRN.TCPL, SAT.ZERO)

CASES = [
    # ea, ec, kw, M, N, K, direct store expected
    (E43, Qu(16, 3), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 4096, 4096, 256, True),
    (E43, Qu(16, 3), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 3999, 4101, 384, True),      # ragged: masked rows and columns
    (E43, Qu(16, 6, True, RND.CONV, SAT.SMGN), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 4096, 4096, 128, True),
    (E43, E43, dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 4096, 4096, 256, False),           # 1-byte container: packed + unpack
    (E43, Qu(16, 3), dict(mul_args=Tags(9, 6), add_args=[Qu(21, 6)]), 512, 512, 256, False),       # small: another kernel
    (E88, Qu(23, 8), dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 2048, 2048, 192, True),   # 3 x 3 limbs
    (E88, Qu(23, 8), dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 2047, 2050, 128, True),
    (E88, Qu(29, 16), dict(mul_args=Tags(17, 16), add_args=[Qu(29, 16)]), 2048, 2048, 128, True),  # 8-byte elements
    (E88, E88, {}, 300, 200, 64, False),                                                           # tree class
]


@pytest.mark.parametrize("pad", [0, 4, 3])    # tight, padded keeping 16-byte alignment, odd (scalar stores)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[3]}x{c[4]}x{c[5]}_{c[1].intBits}_{c[1].fracBits}")
def test_host_layout_c_equals_execute_plus_unpack(oracle, case, pad):
    ea, ec, kw, M, N, K, direct = case
    d = lower(ea, ea, ec, M, N, K, **kw)
    with capi.Context() as ctx:
        plan = capi.Plan(ctx, d)
        assert plan.stores_host_c == direct
        pb = plan.info.packed_bytes
        eb = plan.info.host_elem_bytes[2]
        ldc = M + pad
        pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
        h1, h2 = ctx.alloc(ldc * N * eb), ctx.alloc(ldc * N * eb)
        fill = np.full(ldc * N * eb, 0x5a, np.uint8)
        ctx.h2d(h1, fill)
        ctx.h2d(h2, fill)
        plan.fill(capi.OPERAND_A, 3, 0, pA)
        plan.fill(capi.OPERAND_B, 4, 0, pB)
        plan.execute(pC, pA, pB)
        plan.unpack_c(pC, h1, ldc)
        plan.execute_host_c(h2, pA, pB, ldc)
        ctx.sync()
        a, b = np.zeros(ldc * N * eb, np.uint8), np.zeros(ldc * N * eb, np.uint8)
        ctx.d2h(a, h1)
        ctx.d2h(b, h2)
        for p in (pA, pB, pC, h1, h2):
            ctx.free(p)
        plan.close()
    assert np.array_equal(a, b)
    v = a.view(np.int32 if eb == 4 else np.int64).reshape(N, ldc)
    assert np.count_nonzero(v[:, :M]) > 0.5 * M * N
    if pad:
        assert (a.reshape(N, ldc * eb)[:, M * eb:] == 0x5a).all()     # the padding between columns is never written
    # a block of rows against the oracle (the engine's fill kernel and oracle.fill share the generator: seeds 3 / 4)
    rows = (max(0, M - 24), M)
    A = oracle.fill(ea, M * K, 3)
    B = oracle.fill(ea, K * N, 4)
    exp = oracle.gemm(d, A, B, ec, rows=rows, nthreads=8).reshape(N, M)
    assert np.array_equal(v[:, rows[0]:rows[1]], exp[:, rows[0]:rows[1]])
