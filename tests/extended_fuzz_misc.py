#!/usr/bin/env python3
"""Opt-in long fuzz run (MI355X, not collected by pytest) of the paths the other sweeps reach rarely:
  * quantise-on-load: random formats and modes, awkward doubles (ties, subnormals, huge, non-finite) packed on the device,
    the packed operand read back through a GEMM with the identity matrix and compared with the oracle's from_double;
  * BitStream export: random formats, shapes and chunk sizes against the oracle's string;
  * complex linear class: BasicComplexMul with exact sub-op types on the stacked MFMA path against the oracle.
usage: python tests/extended_fuzz_misc.py [rounds] [seed]"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import qoracle as oracle  # noqa: E402
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import BasicComplexMul, Qcomplex, Qu, lower  # noqa: E402
from test_gpu_fuzz import rand_qu  # noqa: E402
from test_gpu_quantize_on_load import awkward_doubles, quantize  # noqa: E402

ONE = Qu(1, 0, False)


def fail(what, **kw):
    print(json.dumps({"mismatch": what, **{k: str(v) for k, v in kw.items()}}), flush=True)
    sys.exit(1)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    oracle.lib()
    counts = {"f64": 0, "bits": 0, "cplx_linear": 0}
    with capi.Context() as ctx:
        for it in range(rounds):
            # ---- quantise-on-load: C = quantised(A) * I reads the packed operand back exactly
            e = rand_qu(rng, rng.choice([7, 12, 16, 22, 30]))
            M, K = rng.randint(1, 90), rng.randint(1, 70)
            x = awkward_doubles(nrng, max(M * K, 64), 2.0 ** rng.randint(-4, 12))[:M * K].copy()
            d = lower(e, ONE, e, M, K, K, mul_args=e, add_args=[e])
            if capi.classify_status(d)[0] == capi.QG_OK:
                plan = capi.Plan(ctx, d, capi.OPT_ARITHMETIC_CONV)   # (RND::CONV elements: the arithmetic definition, include/qgemul.h)
                pb = plan.info.packed_bytes
                pA, pB, pC = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
                dX = ctx.alloc(x.nbytes)
                ctx.h2d(dX, x)
                plan.pack_f64(capi.OPERAND_A, dX, pA)
                I = np.eye(K, dtype=np.int32).reshape(-1)
                dI = ctx.alloc(I.nbytes)
                ctx.h2d(dI, I)
                plan.pack(capi.OPERAND_B, dI, pB)
                plan.execute(pC, pA, pB)
                out = np.zeros(M * K, dtype=oracle.host_dtype(e))
                dO = ctx.alloc(out.nbytes)
                plan.unpack_c(pC, dO)
                ctx.d2h(out, dO)
                exp = quantize(oracle, e, x)
                if not np.array_equal(out.astype(np.int64), exp):
                    fail("pack_f64", e=e, M=M, K=K)
                # ---- BitStream of that result tensor with random chunking
                w = e.intBits + e.fracBits + int(e.isSigned)
                n = M * K
                tcs = [0] + [c for c in (1, 2, 3, 4, 5, 7, 8, 16) if n % c == 0] + [n]
                ecs = [0] + [c for c in range(1, w + 1) if w % c == 0]
                tc, ec_ = rng.choice(tcs), rng.choice(ecs)
                for fmt in ((capi.BITS_ASCII, capi.BITS_PACKED) if w > 0 else ()):
                    nb = plan.bitstream_bytes(fmt)
                    dBits = ctx.alloc(nb)
                    plan.export_bitstream(pC, dBits, tc, ec_, fmt)
                    raw = np.zeros(nb, dtype=np.uint8)
                    ctx.d2h(raw, dBits)
                    ctx.free(dBits)
                    ref = oracle.bitstream(e, exp, tc, ec_)
                    got = raw.tobytes() if fmt == capi.BITS_ASCII else bytes(np.unpackbits(raw)[:len(ref)] + ord("0"))
                    if got != ref:
                        fail("bitstream", e=e, M=M, K=K, tc=tc, ec=ec_, fmt=fmt)
                    counts["bits"] += 1
                counts["f64"] += 1
                plan.close()
                for p in (pA, pB, pC, dX, dI, dO):
                    ctx.free(p)
            # ---- complex linear class on the stacked MFMA path
            ra, ia = rand_qu(rng, 10), rand_qu(rng, 10)
            rb, ib = (ra, ia) if rng.random() < 0.5 else (rand_qu(rng, 10), rand_qu(rng, 10))
            ex = lambda p, q: Qu(p.intBits + q.intBits + 1, p.fracBits + q.fracBits, p.isSigned or q.isSigned)  # noqa: E731
            ac, bd, ad, bc = ex(ra, rb), ex(ia, ib), ex(ra, ib), ex(ia, rb)
            sub = Qu(max(ac.intBits, bd.intBits) + 2, max(ac.fracBits, bd.fracBits), True)
            add = Qu(max(ad.intBits, bc.intBits) + 2, max(ad.fracBits, bc.fracBits), ad.isSigned or bc.isSigned)
            m = BasicComplexMul(acT=ac, bdT=bd, adT=ad, bcT=bc, acbdT=sub, adbcT=add)
            lv = Qcomplex(Qu(sub.intBits + 12, sub.fracBits, True), Qu(add.intBits + 12, add.fracBits, add.isSigned))
            ec = Qcomplex(rand_qu(rng, rng.choice([12, 20, 34])), rand_qu(rng, rng.choice([12, 20, 34])))
            ca, cb = Qcomplex(ra, ia), Qcomplex(rb, ib)
            M, N, K = rng.randint(1, 150), rng.randint(1, 150), rng.choice([1, 9, 64, 100, 700])
            try:
                d = lower(ca, cb, ec, M, N, K, mul_args=m, add_args=[lv], transposed_a=rng.random() < 0.5)
            except ValueError:
                continue
            st, info = capi.classify_status(d)
            if st != capi.QG_OK:
                continue
            A = oracle.fill(ca, M * K, rng.randint(1, 1 << 30), rng.randint(0, 1))
            B = oracle.fill(cb, K * N, rng.randint(1, 1 << 30), rng.randint(0, 1))
            got = capi.run(d, np.zeros(M * N, dtype=oracle.host_dtype(ec)), A, B)
            exp = oracle.gemm(d, A, B, ec, nthreads=8)
            if not (np.array_equal(got["re"], exp["re"]) and np.array_equal(got["im"], exp["im"])):
                fail("complex", kernel=capi.KERNEL_NAMES[info.kernel], a=ca, b=cb, c=ec, M=M, N=N, K=K)
            counts["cplx_" + ("linear" if capi.KERNEL_NAMES[info.kernel] == "mfma_cplx" else capi.KERNEL_NAMES[info.kernel])] = \
                counts.get("cplx_" + ("linear" if capi.KERNEL_NAMES[info.kernel] == "mfma_cplx" else capi.KERNEL_NAMES[info.kernel]), 0) + 1
    print(json.dumps({"rounds": rounds, **counts, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
