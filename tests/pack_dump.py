"""Helper of tests/test_gpu_resources.py: pack seeded operands and unpack a seeded packed C through the C-ABI, print a
SHA-256 per buffer.  dump(0): fast pack paths; dump(capi.OPT_GENERIC_LAYOUT): the any-format kernels."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qublas_amd import capi  # noqa: E402
from qublas_amd.desc import Qu, Tags, lower  # noqa: E402

CASES = [  # (elem A, elem B, C, M, N, K, transposed_a, lda pad)
    (Qu(8, 8), Qu(8, 8), Qu(23, 8), 300, 200, 130, False, 0),
    (Qu(8, 8), Qu(4, 3), Qu(23, 8), 257, 129, 64, True, 3),
    (Qu(7, 7), Qu(7, 7), Qu(20, 8), 128, 128, 256, False, 4),
    (Qu(4, 3), Qu(4, 3), Qu(12, 3), 515, 70, 1000, False, 1),
    (Qu(8, 8), Qu(8, 8), Qu(23, 8), 1024, 1024, 512, False, 0),
    (Qu(8, 8), Qu(8, 8), Qu(23, 8), 260, 200, 1000, True, 4),     # k contiguous, 16-byte aligned rows, ragged K: vector + scalar loads
    (Qu(4, 3), Qu(4, 3), Qu(16, 3), 4096, 4096, 256, False, 0),   # the 256-row / 128-byte k-tile layout of the two-group kernel
    # centred operands (x - centre, row sums behind the planes): 16-bit words, 24-bit words x unsigned bytes, both axis orders
    (Qu(7, 8), Qu(7, 8), Qu(23, 8), 300, 200, 130, False, 0),
    (Qu(7, 8), Qu(7, 8), Qu(23, 8), 260, 200, 1000, True, 4),
    (Qu(11, 12), Qu(8, 0, False), Qu(30, 12), 257, 129, 192, False, 3),
    (Qu(8, 0, False), Qu(7, 8), Qu(23, 8), 1024, 512, 256, True, 0),
]


def dump(flags):
    out = []
    with capi.Context(0) as ctx:
        for ea, eb, ec, M, N, K, ta, pad in CASES:
            I, F = ea.intBits + eb.intBits + 1, ea.fracBits + eb.fracBits
            d = lower(ea, eb, ec, M, N, K, mul_args=Tags(I, F), add_args=[Qu(I + 12, F)], transposed_a=ta)
            plan = capi.Plan(ctx, d, flags)
            pb = [int(x) for x in plan.info.packed_bytes]
            rng = np.random.default_rng(M + N + K)
            rec = {"case": [M, N, K, ta, pad], "kernel": capi.KERNEL_NAMES[plan.info.kernel]}
            for op, e, rows, cols in ((capi.OPERAND_A, ea, K if ta else M, M if ta else K), (capi.OPERAND_B, eb, K, N)):
                ld = rows + pad
                host = np.zeros(ld * cols, dtype=np.int32)
                vals = rng.integers(e.raw_min, e.raw_max + 1, (cols, rows), dtype=np.int32)
                host.reshape(cols, ld)[:, :rows] = vals
                hd = ctx.alloc(host.nbytes)
                ctx.h2d(hd, host)
                pk = ctx.alloc(pb[0 if op == capi.OPERAND_A else 1])
                plan.pack(op, hd, pk, ld)
                ctx.sync()
                buf = np.zeros(pb[0 if op == capi.OPERAND_A else 1], dtype=np.uint8)
                ctx.d2h(buf, pk)
                t_off, rs_off, rows_p, centre = plan.packed_layout(op)
                if t_off:
                    # the plane mask is the OR of the trailer's 64 words; which word a wave ORs into is the kernel's business
                    tr = buf[t_off:t_off + 256].view(np.uint32)
                    m = np.bitwise_or.reduce(tr)
                    tr[:] = 0
                    tr[0] = m
                rec["A" if op == capi.OPERAND_A else "B"] = hashlib.sha256(buf.tobytes()).hexdigest()
                ctx.free(hd)
                ctx.free(pk)
            # unpack: a seeded packed C image -> host layout with a padded leading dimension
            pc = rng.integers(-1000, 1000, pb[2] // 4, dtype=np.int32)
            pcd = ctx.alloc(pb[2])
            ctx.h2d(pcd, pc)
            ldc = M + pad
            hc = ctx.alloc(ldc * N * 4)
            ctx.h2d(hc, np.full(ldc * N, 77, dtype=np.int32))
            plan.unpack_c(pcd, hc, ldc)
            ctx.sync()
            res = np.zeros(ldc * N, dtype=np.int32)
            ctx.d2h(res, hc)
            rec["C"] = hashlib.sha256(res.tobytes()).hexdigest()
            out.append(rec)
            for q in (pcd, hc):
                ctx.free(q)
            plan.close()
    return out


if __name__ == "__main__":
    print(json.dumps(dump(capi.OPT_GENERIC_LAYOUT if "--generic" in sys.argv else 0)))
