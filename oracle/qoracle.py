"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
engine package (qublas_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from qublas_amd.desc import (Qcomplex, Qu, elem_parts, host_layout, qfmt,  # noqa: E402
                             qgemul_desc, qgemul_epilogue)

_lib = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "qoracle.c")
    hdr = os.path.join(_ROOT, "include", "qgemul.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-o", so, src, "-lpthread"])
    return so


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.qoracle_gemm.restype = C.c_int
        L.qoracle_gemm.argtypes = [C.POINTER(qgemul_desc), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_int]
        L.qoracle_convert.restype = C.c_int64
        L.qoracle_convert.argtypes = [C.c_int64, qfmt, qfmt]
        L.qoracle_round.restype = C.c_int64
        L.qoracle_round.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.qoracle_overflow.restype = C.c_int64
        L.qoracle_overflow.argtypes = [C.c_int64, qfmt]
        L.qoracle_mul.restype = C.c_int64
        L.qoracle_mul.argtypes = [C.c_int64, qfmt, C.c_int64, qfmt, qfmt]
        L.qoracle_add.restype = C.c_int64
        L.qoracle_add.argtypes = [C.c_int64, qfmt, C.c_int64, qfmt, qfmt, C.c_int]
        L.qoracle_reduce.restype = C.c_int64
        L.qoracle_reduce.argtypes = [C.POINTER(C.c_int64), C.c_int64, qfmt, C.POINTER(qfmt), C.c_int]
        L.qoracle_convert128.restype = C.c_int64
        L.qoracle_convert128.argtypes = [C.c_int64, C.c_uint64, qfmt, qfmt]
        L.qoracle_from_double.restype = C.c_int64
        L.qoracle_from_double.argtypes = [C.c_double, qfmt]
        L.qoracle_synth.restype = C.c_int64
        L.qoracle_synth.argtypes = [qfmt, C.c_uint64, C.c_int, C.c_uint64, C.c_int]
        L.qoracle_fill.restype = None
        L.qoracle_fill.argtypes = [C.POINTER(qfmt), C.c_int, C.c_uint64, C.c_int, C.c_int64, C.c_void_p]
        L.qoracle_eltwise.restype = C.c_int
        L.qoracle_eltwise.argtypes = [C.POINTER(qgemul_epilogue), qfmt, C.c_int64, C.POINTER(C.c_int64),
                                      C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]
        L.qoracle_bitstream_cplx.restype = C.c_int
        L.qoracle_bitstream_cplx.argtypes = [qfmt, qfmt, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_char_p]
        L.qoracle_bitstream.restype = C.c_int
        L.qoracle_bitstream.argtypes = [qfmt, C.c_int64, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_char_p]
        pw = C.POINTER(C.c_uint64)
        L.qoracle_convert_w.restype = None
        L.qoracle_convert_w.argtypes = [pw, qfmt, qfmt, pw]
        L.qoracle_mul_w.restype = None
        L.qoracle_mul_w.argtypes = [pw, qfmt, pw, qfmt, qfmt, pw]
        L.qoracle_add_w.restype = None
        L.qoracle_add_w.argtypes = [pw, qfmt, pw, qfmt, qfmt, C.c_int, pw]
        L.qoracle_elem_bytes.restype = C.c_int
        L.qoracle_elem_bytes.argtypes = [C.POINTER(qfmt), C.c_int]
        L.qoracle_imag_offset.restype = C.c_int
        L.qoracle_imag_offset.argtypes = [C.POINTER(qfmt), C.c_int]
        _lib = L
    return _lib


WIDE = np.dtype([("lo", "<u8"), ("hi", "<i8")])   # ArbiInt<N > 64>: two little-endian words, the top one carries the sign (QuBLAS.h:572-573)


def _part_dtype(nbytes: int):
    return "<i4" if nbytes == 4 else "<i8" if nbytes == 8 else WIDE


def host_dtype(e) -> np.dtype:
    """numpy dtype of one host-layout element (structured for complex, and for parts of more than 64 storage bits)."""
    size, off, (sr, si) = host_layout(e)
    if not isinstance(e, Qcomplex):
        return np.dtype(_part_dtype(sr))
    return np.dtype({"names": ["re", "im"], "formats": [_part_dtype(sr), _part_dtype(si)], "offsets": [0, off], "itemsize": size})


def to_host(values, e: Qu) -> np.ndarray:
    """Python integers (any size) -> a host-layout array of a REAL element type"""
    dt = host_dtype(e)
    if dt != WIDE:
        return np.asarray([int(v) for v in values], dtype=dt)
    out = np.zeros(len(values), dtype=WIDE)
    for i, v in enumerate(values):
        v = int(v)
        out[i] = (v & (2**64 - 1), v >> 64)
    return out


def from_host(arr: np.ndarray) -> list:
    """a real host-layout array -> Python integers"""
    if arr.dtype == WIDE:
        return [(int(h) << 64) | int(l) for l, h in zip(arr["lo"], arr["hi"])]
    return [int(v) for v in arr]


def _w(v: int):
    v = int(v)
    return (C.c_uint64 * 2)(v & (2**64 - 1), (v >> 64) & (2**64 - 1))


def _unw(w) -> int:
    v = (int(w[1]) << 64) | int(w[0])
    return v - (1 << 128) if v >> 127 else v


def convert_w(x: int, f: Qu, to: Qu) -> int:
    y = (C.c_uint64 * 2)()
    lib().qoracle_convert_w(_w(x), f.c(), to.c(), y)
    return _unw(y)


def mul_w(a: int, fa: Qu, b: int, fb: Qu, r: Qu) -> int:
    y = (C.c_uint64 * 2)()
    lib().qoracle_mul_w(_w(a), fa.c(), _w(b), fb.c(), r.c(), y)
    return _unw(y)


def add_w(a: int, fa: Qu, b: int, fb: Qu, r: Qu, sub: bool = False) -> int:
    y = (C.c_uint64 * 2)()
    lib().qoracle_add_w(_w(a), fa.c(), _w(b), fb.c(), r.c(), int(sub), y)
    return _unw(y)


def _f2(e):
    r, i = elem_parts(e)
    arr = (qfmt * 2)(r.c(), i.c())
    return arr


def fill(e, n: int, seed: int, dist: int = 0) -> np.ndarray:
    """Synthetic tight host tensor of n elements (same generator as the engine's fill kernel)."""
    out = np.zeros(n, dtype=host_dtype(e))
    lib().qoracle_fill(_f2(e), int(isinstance(e, Qcomplex)), seed, dist, n, out.ctypes.data_as(C.c_void_p))
    return out


def gemm(desc: qgemul_desc, A: np.ndarray, B: np.ndarray, c_elem, *, lda=0, ldb=0, ldc=0,
         rows=(0, 0), cols=(0, 0), nthreads: int = 1, out: np.ndarray | None = None) -> np.ndarray:
    """Run the CPU restatement.  A, B: contiguous host-layout arrays (column-major flattening)."""
    M, N = desc.M, desc.N
    ld = ldc or M
    if out is None:
        out = np.zeros(ld * N, dtype=host_dtype(c_elem))
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    st = lib().qoracle_gemm(C.byref(desc), out.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                            B.ctypes.data_as(C.c_void_p), lda, ldb, ldc, rows[0], rows[1], cols[0], cols[1], nthreads)
    if st != 0:
        raise RuntimeError(f"qoracle_gemm failed: {st}")
    return out


def eltwise(ep: qgemul_epilogue, c: Qu, x: np.ndarray, E) -> np.ndarray:
    """CPU restatement of the element-wise chain on raw values: x = C's values, E[k] = operand k (array, or 1 element
    for a scalar stage).  Returns D's raw values as int64."""
    x = np.ascontiguousarray(x, dtype=np.int64)
    Es = [np.ascontiguousarray(np.asarray(e).reshape(-1), dtype=np.int64) for e in E]
    ptrs = (C.POINTER(C.c_int64) * max(1, len(Es)))(*[e.ctypes.data_as(C.POINTER(C.c_int64)) for e in Es])
    out = np.zeros(x.size, dtype=np.int64)
    st = lib().qoracle_eltwise(C.byref(ep), c.c(), x.size, x.ctypes.data_as(C.POINTER(C.c_int64)), ptrs,
                               out.ctypes.data_as(C.POINTER(C.c_int64)))
    if st != 0:
        raise RuntimeError(f"qoracle_eltwise failed: {st}")
    return out


def eltwise_cplx(epc, c, x_re: np.ndarray, x_im: np.ndarray, E_re, E_im):
    """A complex chain (qgemul_epilogue_cplx) = the chain of the real parts and the chain of the imaginary parts
    (QuBLAS.h:3549-3589, :3604-3707: every supported operator is part-wise).  E_re[k] / E_im[k]: the values stage k of that
    part reads (array or 1 element; anything for a PASS stage).  Returns (D_re, D_im) raw values."""
    return (eltwise(epc.part[0], c.real, x_re, E_re), eltwise(epc.part[1], c.imag, x_im, E_im))


def bitstream(f: Qu, x: np.ndarray, tensor_chunk: int = 0, elem_chunk: int = 0) -> bytes:
    """CPU restatement of BitStream<tensorProcessT, elemProcessT>(tensor): x = raw values in storage order; chunk 0 = l2r,
    k > 0 = r2l<k>.  Returns the '0'/'1' characters."""
    x = np.ascontiguousarray(x, dtype=np.int64)
    w = f.intBits + f.fracBits + (1 if f.isSigned else 0)
    buf = C.create_string_buffer(int(x.size) * w)
    st = lib().qoracle_bitstream(f.c(), x.size, x.ctypes.data_as(C.POINTER(C.c_int64)), tensor_chunk, elem_chunk, buf)
    if st != 0:
        raise ValueError("invalid chunk for this tensor / element width")
    return buf.raw


def bitstream_cplx(e: Qcomplex, re: np.ndarray, im: np.ndarray, tensor_chunk: int = 0, elem_chunk: int = 0) -> bytes:
    """The same for a complex tensor: every element is "(re-bits, im-bits)" (QuBLAS.h:2553-2556) and the element-level chunk
    reversal acts on that whole string."""
    re = np.ascontiguousarray(re, dtype=np.int64)
    im = np.ascontiguousarray(im, dtype=np.int64)
    w = sum(f.intBits + f.fracBits + (1 if f.isSigned else 0) for f in (e.real, e.imag)) + 4
    buf = C.create_string_buffer(int(re.size) * w)
    p = C.POINTER(C.c_int64)
    st = lib().qoracle_bitstream_cplx(e.real.c(), e.imag.c(), re.size, re.ctypes.data_as(p), im.ctypes.data_as(p), tensor_chunk, elem_chunk, buf)
    if st != 0:
        raise ValueError("invalid chunk for this tensor / element width")
    return buf.raw

