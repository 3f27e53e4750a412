#!/usr/bin/env python3
"""Opt-in long fuzz run of element-wise chains after COMPLEX GEMMs on an MI355X (not collected by pytest): random chains
of 0..4 operators (complex add / sub with realT / imagT / loose tags, real add / sub / mul in both orders, scalar and tensor
operands, random intermediate tensor types) behind random complex GEMMs (Basic / TF, default and tagged sub-operations,
exact "linear class" variants), against oracle GEMM + the oracle's part-wise chains; every third case also exports D as a
BitStream with random chunking against the oracle's string.
usage: python tests/extended_fuzz_cplx_eltwise.py [cases] [seed]"""
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
from qublas_amd.desc import BasicComplexMul, EwC, Qcomplex, Qu, TFComplexMul, lower, lower_epilogue_cplx  # noqa: E402
from test_gpu_fuzz import rand_qu, rand_tags  # noqa: E402


def rand_c(rng, bits):
    return Qcomplex(rand_qu(rng, bits), rand_qu(rng, bits))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
    oracle.lib()
    ran = skipped = streams = 0
    kernels, ops = {}, {}
    for it in range(cases):
        ea = rand_c(rng, rng.choice([5, 7, 9]))
        kind = rng.random()
        if kind < 0.35:      # every sub-operation and level exact: the stacked MFMA path
            a, b = ea.real, ea.imag
            w = lambda x, y: Qu(x.intBits + y.intBits + 1, x.fracBits + y.fracBits)
            ac, bd, ad, bc = w(a, a), w(b, b), w(a, b), w(b, a)
            re = Qu(max(ac.intBits, bd.intBits) + 1, max(ac.fracBits, bd.fracBits))
            im = Qu(max(ad.intBits, bc.intBits) + 1, max(ad.fracBits, bc.fracBits))
            kw = dict(mul_args=BasicComplexMul(acT=ac, bdT=bd, adT=ad, bcT=bc, acbdT=re, adbcT=im),
                      add_args=[Qcomplex(Qu(re.intBits + 11, re.fracBits), Qu(im.intBits + 11, im.fracBits))])
        elif kind < 0.7:
            kw = dict(mul_args=TFComplexMul() if rng.random() < 0.5 else BasicComplexMul(),
                      add_args=[rand_c(rng, 14) for _ in range(rng.randint(0, 2))] or None)
        else:
            kw = dict(mul_args=BasicComplexMul(acT=rand_tags(rng, ea.real), adbcT=rand_tags(rng, ea.imag), loose=rand_tags(rng, ea.real))
                      if rng.random() < 0.5 else TFComplexMul(abT=rand_tags(rng, ea.real), ABT=rand_tags(rng, ea.imag)))
        ec = rand_c(rng, rng.choice([7, 16, 24, 30]))
        M, N, K = rng.randint(1, 150), rng.randint(1, 150), rng.choice([1, 7, 64, 100, 256])
        stages = []
        for _ in range(rng.randint(0, 4)):
            op = rng.choice(["add", "sub", "mul"])
            cplx = op != "mul" and rng.random() < 0.55
            e = rand_c(rng, rng.choice([6, 10, 16])) if cplx else rand_qu(rng, rng.choice([6, 10, 16]))
            base = e.real if cplx else e
            pick = rng.random()
            kwt = {}
            if pick < 0.3:
                kwt = dict(real_tags=rand_tags(rng, base), imag_tags=rand_tags(rng, base))
            elif pick < 0.5:
                kwt = dict(real_tags=rand_tags(rng, base)) if rng.random() < 0.5 else dict(imag_tags=rand_tags(rng, base))
            elif pick < 0.8:
                kwt = dict(tags=rand_tags(rng, base))
            stages.append(EwC(op, e, x_first=rng.random() < 0.6, scalar=rng.random() < 0.4,
                              into=rand_c(rng, rng.choice([12, 20, 30])) if rng.random() < 0.5 else None, **kwt))
            ops[(op, cplx)] = ops.get((op, cplx), 0) + 1
        dq = rand_c(rng, rng.choice([7, 15, 24, 40]))
        try:
            d = lower(ea, ea, ec, M, N, K, transposed_a=rng.random() < 0.5, **kw)
            epc = lower_epilogue_cplx(ec, stages, dq)
        except ValueError:
            skipped += 1
            continue
        flags = rng.choice([0, 0, capi.OPT_FORCE_TREE, capi.OPT_GENERIC_TREE, capi.OPT_RUNTIME_MODES])
        st, info = capi.classify_ep_status(d, epc, flags)
        if st != capi.QG_OK:
            skipped += 1
            continue
        dist = rng.randint(0, 1)
        A = oracle.fill(ea, M * K, rng.randint(1, 1 << 30), dist)
        B = oracle.fill(ea, K * N, rng.randint(1, 1 << 30), dist)
        Eh, Ere, Eim = [], [], []
        for s_ in stages:
            h = oracle.fill(s_.e, 1 if s_.scalar else M * N, rng.randint(1, 1 << 30), rng.randint(0, 1))
            Eh.append(h)
            if isinstance(s_.e, Qcomplex):
                Ere.append(h["re"].astype(np.int64)); Eim.append(h["im"].astype(np.int64))
            else:
                Ere.append(h.astype(np.int64))
                Eim.append(h.astype(np.int64) if s_.op == "mul" else np.zeros(1, dtype=np.int64))
        out = np.zeros(M * N, dtype=oracle.host_dtype(dq))
        capi.run_ep(d, epc, out, A, B, Eh, flags=flags)
        Cx = oracle.gemm(d, A, B, ec, nthreads=8)
        exp_re, exp_im = oracle.eltwise_cplx(epc, ec, Cx["re"].astype(np.int64), Cx["im"].astype(np.int64), Ere, Eim)
        k = capi.KERNEL_NAMES[info.kernel]

        def fail(what):
            print(json.dumps({"mismatch": it, "what": what, "kernel": k, "M": M, "N": N, "K": K, "flags": flags, "a": str(ea), "c": str(ec),
                              "d": str(dq), "kw": str(kw), "stages": str(stages)}), flush=True)
            sys.exit(1)
        if not (np.array_equal(out["re"].astype(np.int64), exp_re) and np.array_equal(out["im"].astype(np.int64), exp_im)):
            fail("chain")
        if it % 3 == 0:
            # BitStream of D through the resident entry points, random chunking
            w = sum(f.intBits + f.fracBits + int(f.isSigned) for f in (dq.real, dq.imag)) + 4
            divs = [c for c in range(1, w + 1) if w % c == 0]
            tdivs = [c for c in (1, 2, 3, 4, 5, 8, M, N, M * N) if c > 0 and (M * N) % c == 0]
            ec_, tc = rng.choice([0] + divs), rng.choice([0] + tdivs)
            fmt = rng.choice([capi.BITS_ASCII, capi.BITS_PACKED])
            with capi.Context() as ctx:
                plan = capi.Plan(ctx, d, flags=flags, epilogue=epc)
                dA, dB = ctx.alloc(A.nbytes), ctx.alloc(B.nbytes)
                ctx.h2d(dA, A); ctx.h2d(dB, B)
                pA, pB, pD = (ctx.alloc(max(16, int(plan.info.packed_bytes[i]))) for i in range(3))
                plan.pack(capi.OPERAND_A, dA, pA); plan.pack(capi.OPERAND_B, dB, pB)
                packed, sre, sim = [], [], []
                for kk, s_ in enumerate(stages):
                    if plan.packed_e_bytes(kk) == 0:
                        packed.append(0); sre.append(int(Ere[kk][0])); sim.append(int(Eim[kk][0]))
                        continue
                    dE, pE = ctx.alloc(Eh[kk].nbytes), ctx.alloc(plan.packed_e_bytes(kk))
                    ctx.h2d(dE, Eh[kk]); plan.pack_e(kk, dE, pE)
                    packed.append(pE); sre.append(0); sim.append(0)
                plan.execute_ep(pD, pA, pB, plan.ep_args(packed=packed, scalars=sre, scalars_im=sim))
                nb = plan.bitstream_bytes(fmt)
                dBits = ctx.alloc(max(16, nb))
                plan.export_bitstream(pD, dBits, tc, ec_, fmt)
                got = np.zeros(nb, dtype=np.uint8)
                ctx.d2h(got, dBits)
                plan.close()
            ref = oracle.bitstream_cplx(dq, exp_re, exp_im, tc, ec_)
            if fmt == capi.BITS_ASCII:
                ok = got.tobytes() == ref
            else:
                want = np.frombuffer(bytes(ch for ch in ref if ch in b"01"), dtype=np.uint8) - ord("0")
                bits = np.unpackbits(got)
                ok = np.array_equal(bits[:want.size], want) and not bits[want.size:].any()
            if not ok:
                fail(f"bitstream tc={tc} ec={ec_} fmt={fmt}")
            streams += 1
        kernels[k] = kernels.get(k, 0) + 1
        ran += 1
    print(json.dumps({"complex_chains_run": ran, "skipped_unsupported": skipped, "bitstreams": streams, "kernels": kernels,
                      "stages_by_op_and_complex_operand": {f"{o}/{'c' if c else 'r'}": n for (o, c), n in sorted(ops.items())},
                      "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
