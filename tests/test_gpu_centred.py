"""CENTRED operands of the linear class (QPackedGeom::offs, qg_plan.h: qg_limbs_centred): x - c stored in balanced int8 limbs where
that saves a limb — signed formats of exactly 16 / 24 / 32 bits (3 -> 2, 4 -> 3, 5 -> 4 limbs), unsigned formats (uint8: 2 -> 1) —
and the centres taken back out with the operands' row sums, in the MFMA kernels' epilogues (limb kernels) or in the composite
plan's combine pass (single-limb pairs, limb groups, k-chunks).  Every case: the limb counts the planner reports, bit-exact
against the oracle, and byte-identical to the same descriptor on plain balanced limbs (QG_OPT_BALANCED_LIMBS)."""
import numpy as np
import pytest

from qublas_amd import capi
from qublas_amd.desc import Qu, RND, SAT, TRN, WRP, Tags, lower

pytestmark = pytest.mark.gpu

Q78 = Qu(7, 8)                      # 16 bits
U8 = Qu(8, 0, False)
U16 = Qu(10, 6, False)
Q1112 = Qu(11, 12)                  # 24 bits
Q1516 = Qu(15, 16)                  # 32 bits
E43 = Qu(4, 3)
E88 = Qu(8, 8)                      # 17 bits: nothing to gain

CASES = [
    # (A, B, C, lowering keywords, shapes, limbs centred, limbs balanced, kernel)
    (Q78, Q78, Qu(20, 8), dict(mul_args=Tags(15, 16), add_args=[Qu(28, 16)]), [(256, 256, 256), (130, 70, 200), (1, 3, 1), (2048, 2048, 512)], [2, 2], [3, 3], "mfma_i8_limb"),
    (Q78, Q78, Qu(9, 3, True, RND.CONV, SAT.SMGN), dict(mul_args=Tags(15, 16), add_args=[Qu(28, 16)]), [(257, 129, 64)], [2, 2], [3, 3], "mfma_i8_limb"),
    (U8, U8, Qu(26, 0, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), [(256, 256, 256), (130, 70, 200), (2048, 1024, 2048)], [1, 1], [2, 2], "mfma_i8"),
    # large single-limb problems: the two-group kernel's own epilogue restores the sum in 64 bits (C of at most 31 bits), 4-, 2- and 1-byte containers
    (U8, U8, Qu(28, 0, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), [(4096, 4096, 256), (4000, 4100, 320)], [1, 1], [2, 2], "mfma_i8"),
    (U8, U8, Qu(25, -16, False, RND.CONV, SAT.SMGN), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), [(4096, 4096, 512)], [1, 1], [2, 2], "mfma_i8"),
    (U8, U8, Qu(23, -16, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), [(4096, 4096, 512)], [1, 1], [2, 2], "mfma_i8"),
    (U8, E43, Qu(20, 3), dict(mul_args=Tags(12, 3), add_args=[Qu(24, 3)]), [(300, 200, 100), (4096, 4096, 256)], [1, 1], [2, 1], "mfma_i8"),      # one centred, one not
    (U8, U8, Qu(36, 4, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), [(300, 200, 100), (1024, 1024, 1024)], [1, 1], [2, 2], "mfma_i8"),   # a C beyond 31 bits: raw int32 slab + combine pass
    (Q78, E43, Qu(20, 8), dict(mul_args=Tags(12, 11), add_args=[Qu(22, 11)]), [(300, 200, 100), (1024, 1024, 256)], [2, 1], [3, 1], "mfma_i8_limb"),
    (E88, Q78, Qu(20, 8), dict(mul_args=Tags(16, 16), add_args=[Qu(28, 16)]), [(200, 300, 128)], [3, 2], [3, 3], "mfma_i8_limb"),
    (U16, U16, Qu(30, 12, False), dict(mul_args=Tags(20, 12, False), add_args=[Qu(32, 12, False)]), [(256, 384, 512)], [2, 2], [3, 3], "mfma_i8_limb"),
    (Q1112, Q1112, Qu(30, 12), dict(mul_args=Tags(23, 24), add_args=[Qu(35, 24)]), [(256, 256, 256), (130, 70, 200), (1024, 1024, 512)], [3, 3], [4, 4], "mfma_i8_limb"),
    (Q1516, Q1516, Qu(43, 32), dict(mul_args=Tags(31, 32), add_args=[Qu(43, 32)]), [(128, 128, 256), (130, 70, 200)], [4, 4], [5, 5], "mfma_i8_limb"),   # 128-bit combine
    (Q1516, E43, Qu(30, 8), dict(mul_args=Tags(20, 19), add_args=[Qu(30, 19)]), [(200, 100, 300)], [4, 1], [5, 1], "mfma_i8_limb"),
    (U8, U8, Qu(30, 0, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(34, 0, False)]), [(64, 64, 140000)], [1, 1], [2, 2], "mfma_i8"),          # k-chunks + centres
    (Q78, Q78, Qu(30, 8), dict(mul_args=Tags(15, 16), add_args=[Qu(33, 16)]), [(64, 32, 70000)], [2, 2], [3, 3], "mfma_i8_limb"),
]


def _run(d, A, B, ec, oracle, flags=0, **ld):
    M, N = d.M, d.N
    out = np.zeros((ld.get("ldc", 0) or M) * N, dtype=oracle.host_dtype(ec))
    return capi.run(d, out, A, B, flags=flags, **ld)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_centred_operands_vs_oracle_and_balanced_limbs(oracle, case):
    ea, eb, ec, kw, shapes, lc, lb, kernel = CASES[case]
    for M, N, K in shapes:
        for ta in (False, True):
            d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
            info = capi.classify(d)
            assert capi.KERNEL_NAMES[info.kernel] == kernel and list(info.limbs) == lc, (info.reason, list(info.limbs))
            assert list(capi.classify(d, capi.OPT_BALANCED_LIMBS).limbs) == lb
            for dist in (0, 1):
                A = oracle.fill(ea, M * K, 11, dist)
                B = oracle.fill(eb, K * N, 12, dist)
                got = _run(d, A, B, ec, oracle)
                bal = _run(d, A, B, ec, oracle, flags=capi.OPT_BALANCED_LIMBS)
                assert got.tobytes() == bal.tobytes()
                rows = None if M * N * K <= 3e8 else (max(0, M - 16), M)
                exp = oracle.gemm(d, A, B, ec, nthreads=8, rows=rows) if rows else oracle.gemm(d, A, B, ec, nthreads=8)
                g2, e2 = got.reshape(N, M), exp.reshape(N, M)
                if rows:
                    assert np.array_equal(g2[:, rows[0]:rows[1]], e2[:, rows[0]:rows[1]])
                else:
                    assert np.array_equal(g2, e2)
            if M * N * K > 1e8:
                break    # (one orientation of the large shapes)


def test_centred_operands_with_leading_dimensions_and_device_fill(oracle):
    """padded leading dimensions through the generic pack; the engine's own fill kernel (the bench's path) against oracle.fill"""
    ea, ec, kw = Q78, Qu(20, 8), dict(mul_args=Tags(15, 16), add_args=[Qu(28, 16)])
    M, N, K = 150, 90, 260
    d = lower(ea, ea, ec, M, N, K, **kw)
    lda, ldb, ldc = M + 3, K + 5, M + 7
    A = oracle.fill(ea, lda * K, 3)
    B = oracle.fill(ea, ldb * N, 4)
    out = np.zeros(ldc * N, dtype=oracle.host_dtype(ec))
    out.view(np.uint8)[:] = 0x5a
    exp = out.copy()
    capi.run(d, out, A, B, lda=lda, ldb=ldb, ldc=ldc)
    oracle.gemm(d, A, B, ec, lda=lda, ldb=ldb, ldc=ldc, out=exp, nthreads=8)
    assert out.tobytes() == exp.tobytes()
    with capi.Context() as ctx:
        for e, ecc, kww in ((Q78, ec, kw), (U8, Qu(26, 0, False), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]))):
            M, N, K = 512, 384, 640
            d = lower(e, e, ecc, M, N, K, **kww)
            plan = capi.Plan(ctx, d)
            pb = plan.info.packed_bytes
            pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            hC = ctx.alloc(M * N * plan.info.host_elem_bytes[2])
            plan.fill(capi.OPERAND_A, 3, 0, pA)
            plan.fill(capi.OPERAND_B, 4, 0, pB)
            plan.execute(pC, pA, pB)
            plan.unpack_c(pC, hC, M)
            ctx.sync()
            got = np.zeros(M * N, dtype=oracle.host_dtype(ecc))
            ctx.d2h(got.view(np.uint8), hC)
            exp = oracle.gemm(d, oracle.fill(e, M * K, 3), oracle.fill(e, K * N, 4), ecc, nthreads=8)
            assert np.array_equal(got, exp)
            for p in (pA, pB, pC, hC):
                ctx.free(p)
            plan.close()


def test_centred_operands_with_element_wise_chains(oracle):
    """the reference's lazy tensor operators after a GEMM on centred operands: as the pass after the kernel, fused into the limb
    kernels' epilogue (the centres go back in BEFORE the one round + overflow into C, then the chain), after a composite plan's
    combine pass — against oracle GEMM + oracle chain and against the plain balanced limbs"""
    from qublas_amd.desc import Ew, lower_epilogue
    B106, S34 = Qu(10, 6), Qu(3, 4)
    chains = [([Ew("mul", S34, Tags(24, 8), scalar=True, into=Qu(24, 8)), Ew("add", B106)], Qu(16, 6, True, RND.ZERO, SAT.TCPL)),
              ([], Qu(9, 3, True, RND.NEG_INF, SAT.SMGN))]
    cfgs = [(Q78, Qu(15, 8), dict(mul_args=Tags(15, 16), add_args=[Qu(27, 16)]), 200, 136, 192),
            (Q1112, Qu(15, 8), dict(mul_args=Tags(23, 24), add_args=[Qu(35, 24)]), 200, 136, 192),
            (U8, Qu(15, 8), dict(mul_args=Tags(16, 0, False), add_args=[Qu(28, 0, False)]), 130, 260, 128)]
    for ea, ec, kw, M, N, K in cfgs:
        d = lower(ea, ea, ec, M, N, K, **kw)
        A, B = oracle.fill(ea, M * K, 1), oracle.fill(ea, K * N, 2)
        sh = ea.intBits + ea.fracBits - 10 if ea.isSigned else 4   # small magnitudes: C must not saturate everywhere, or a wrong sum would hide behind the clamp
        A, B = (A >> sh).astype(A.dtype), (B >> sh).astype(B.dtype)
        Cx = oracle.gemm(d, A, B, ec, nthreads=8).astype(np.int64)
        for stages, dq in chains:
            ep = lower_epilogue(ec, stages, dq)
            st, info = capi.classify_ep_status(d, ep)
            assert st == capi.QG_OK, info.reason
            Eh = [oracle.fill(s.e, 1 if s.scalar else M * N, 70 + k, 0) for k, s in enumerate(stages)]
            exp = oracle.eltwise(ep, ec, Cx, [e.astype(np.int64) for e in Eh])
            assert np.mean(np.abs(Cx) >= ec.raw_max) < 0.5
            for fl in (0, capi.OPT_FUSED_EPILOGUE, capi.OPT_UNFUSED_EPILOGUE, capi.OPT_BALANCED_LIMBS):
                out = np.zeros(M * N, dtype=np.int32 if dq.storage_bits <= 32 else np.int64)
                got = capi.run_ep(d, ep, out, A, B, Eh, flags=fl)
                assert np.array_equal(got.astype(np.int64), exp), (str(ea), fl, len(stages))


def test_wide_plans_keep_plain_limbs_where_int64_row_sums_could_overflow(oracle):
    """Two finds of the wide fuzzer: in a plan with the 128-bit combine the row sums (int64) of a 56- or 61-bit operand over K terms
    do not fit, so such descriptors keep the plain balanced limbs (a 64-bit plan may let the row sums wrap: all of its correction
    is arithmetic modulo 2^64)."""
    cases = [(Qu(13, 48, False), Qu(9, 14), Qu(5, 0, True, TRN.TCPL, SAT.TCPL), 257, 129, 7, True, dict(mul_args=Qu(23, 62), add_args=[Qu(26, 62)]), [8, 4]),
             (Qu(44, 11), Qu(10, 21), Qu(48, 24, True, TRN.TCPL, SAT.SMGN), 16, 16, 43519, False, dict(mul_args=Qu(55, 32), add_args=[Qu(71, 32)]), [8, 5]),
             (Q1516, Q1516, Qu(43, 32), 64, 48, 4096, False, dict(mul_args=Tags(31, 32), add_args=[Qu(43, 32)]), [4, 4])]     # (fits: centred)
    for ea, eb, ec, M, N, K, ta, kw, limbs in cases:
        d = lower(ea, eb, ec, M, N, K, transposed_a=ta, **kw)
        info = capi.classify(d)
        assert list(info.limbs) == limbs and b"128-bit" in info.reason, (list(info.limbs), info.reason)
        A, B = oracle.fill(ea, M * K, 21), oracle.fill(eb, K * N, 22)
        got = _run(d, A, B, ec, oracle)
        exp = oracle.gemm(d, A, B, ec, nthreads=8)
        assert got.tobytes() == exp.tobytes()
