"""ctypes binding of libqugemm.so — the C-ABI of include/qgemul.h.

This is plumbing for tests and the benchmark harness; the product boundary is the C-ABI itself
and the C++23 header include/QuBLAS_amd.h that lowers the reference's tag API onto it.
The library has no CPU arithmetic path: loading fails loudly if it has not been built, and every
compute call fails with QG_ENOGPU when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .desc import (Qcomplex, Qu, host_layout, qgemul_desc, qgemul_ep_args, qgemul_epilogue, qgemul_epilogue_cplx, qgemul_info,
                   qgemul_opts)

_HERE = os.path.dirname(os.path.abspath(__file__))
# QUBLAS_AMD_DIAG=1 (tools/ only) loads the diagnostic build: environment A/B switches and ablation variants exist there and
# nowhere else (qublas_amd/build.py).  Tests, smoke and bench load the product library.
LIB_PATH = os.path.join(_HERE, "libqugemm_diag.so" if os.environ.get("QUBLAS_AMD_DIAG") == "1" else "libqugemm.so")

QG_OK, QG_EINVAL, QG_EUNSUPPORTED, QG_EHIP, QG_ERCCL, QG_ERANGE, QG_ENOGPU = 0, -1, -2, -3, -4, -5, -6
OPT_FORCE_TREE, OPT_CHECK_RANGE, OPT_GENERIC_TREE, OPT_RUNTIME_MODES, OPT_FUSED_EPILOGUE, OPT_UNFUSED_EPILOGUE = 1, 2, 4, 8, 16, 32
OPT_GENERIC_LAYOUT, OPT_LOCKSTEP_TILES, OPT_ARITHMETIC_CONV, OPT_ALL_DEVICES = 64, 128, 256, 512
OPT_BALANCED_LIMBS = 1024   # never centre an operand (x - c in balanced limbs): the plain balanced limbs, result-identical
OPERAND_A, OPERAND_B, OPERAND_C = 0, 1, 2
BITS_ASCII, BITS_PACKED = 0, 1
KERNEL_NAMES = {0: "none", 1: "mfma_i8", 2: "mfma_i8_limb", 3: "tree_i32", 4: "tree_i64", 5: "tree_cplx", 6: "tree_cplx_i32", 7: "mfma_cplx", 8: "gemv_i32", 9: "gemv_i64", 10: "tree_i128"}

EXPORTS = [
    "qgemul_classify", "qgemul_strerror", "qgemul_abi_version", "qgemul_last_hip_error", "qgemul_run",
    "qgemul_ctx_create", "qgemul_ctx_destroy", "qgemul_ctx_sync", "qgemul_ctx_stream",
    "qgemul_plan_create", "qgemul_plan_destroy", "qgemul_plan_info",
    "qgemul_dev_alloc", "qgemul_dev_free", "qgemul_memcpy_h2d", "qgemul_memcpy_d2h",
    "qgemul_pack", "qgemul_pack_f64", "qgemul_unpack_c", "qgemul_execute", "qgemul_fill_packed", "qgemul_time_execute",
    "qgemul_classify_ep", "qgemul_plan_create_ep", "qgemul_packed_e_bytes", "qgemul_pack_e", "qgemul_execute_ep",
    "qgemul_time_execute_ep", "qgemul_run_ep", "qgemul_plan_fuses_epilogue", "qgemul_plan_packed_layout", "qgemul_classify_epc", "qgemul_plan_create_epc", "qgemul_run_epc",
    "qgemul_bitstream_bytes", "qgemul_export_bitstream", "qgemul_run_release", "qgemul_run_sharded", "qgemul_execute_host_c", "qgemul_plan_stores_host_c",
    "qgemul_comm_unique_id", "qgemul_comm_create", "qgemul_comm_destroy", "qgemul_comm_info", "qgemul_gather_packed_c", "qgemul_comm_fence",
    "qgemul_comm_sync", "qgemul_comm_barrier", "qgemul_comm_max_f64", "qgemul_last_rccl_error", "qgemul_ctx_device",
]

_lib = None


class QgemulError(RuntimeError):
    def __init__(self, status: int, what: str = ""):
        self.status = status
        msg = lib().qgemul_strerror(status).decode()
        if status == QG_EHIP:
            msg += f" (hipError {lib().qgemul_last_hip_error()})"
        super().__init__(f"{what}: {msg}" if what else msg)


def lib() -> C.CDLL:
    """Load the engine.  Raises if the library is missing: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -m qublas_amd.build` (hipcc, gfx950). "
                              "The engine has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, i64, u32, u64 = C.c_void_p, C.c_int64, C.c_uint32, C.c_uint64
        pd = C.POINTER(qgemul_desc)
        L.qgemul_classify.argtypes = [pd, u32, C.POINTER(qgemul_info)]
        L.qgemul_strerror.restype = C.c_char_p
        L.qgemul_strerror.argtypes = [C.c_int]
        L.qgemul_abi_version.restype = u32
        L.qgemul_run.argtypes = [pd, vp, vp, vp, C.POINTER(qgemul_opts)]
        L.qgemul_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.qgemul_ctx_destroy.argtypes = [vp]
        L.qgemul_ctx_destroy.restype = None
        L.qgemul_ctx_sync.argtypes = [vp]
        L.qgemul_ctx_stream.argtypes = [vp]
        L.qgemul_ctx_stream.restype = vp
        L.qgemul_plan_create.argtypes = [vp, pd, u32, C.POINTER(vp)]
        L.qgemul_plan_destroy.argtypes = [vp]
        L.qgemul_plan_destroy.restype = None
        L.qgemul_plan_info.argtypes = [vp, C.POINTER(qgemul_info)]
        L.qgemul_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.qgemul_dev_free.argtypes = [vp, vp]
        L.qgemul_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.qgemul_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.qgemul_pack.argtypes = [vp, C.c_int, vp, i64, vp]
        L.qgemul_pack_f64.argtypes = [vp, C.c_int, vp, i64, vp]
        L.qgemul_unpack_c.argtypes = [vp, vp, vp, i64]
        L.qgemul_execute.argtypes = [vp, vp, vp, vp]
        L.qgemul_fill_packed.argtypes = [vp, C.c_int, u64, C.c_int, vp]
        L.qgemul_time_execute.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
        pe = C.POINTER(qgemul_epilogue)
        pa = C.POINTER(qgemul_ep_args)
        L.qgemul_classify_ep.argtypes = [pd, pe, u32, C.POINTER(qgemul_info)]
        L.qgemul_plan_create_ep.argtypes = [vp, pd, pe, u32, C.POINTER(vp)]
        pec = C.POINTER(qgemul_epilogue_cplx)
        L.qgemul_classify_epc.argtypes = [pd, pec, u32, C.POINTER(qgemul_info)]
        L.qgemul_plan_create_epc.argtypes = [vp, pd, pec, u32, C.POINTER(vp)]
        L.qgemul_run_epc.argtypes = [pd, pec, vp, vp, vp, C.POINTER(vp), C.POINTER(qgemul_opts)]
        L.qgemul_plan_fuses_epilogue.argtypes = [vp]
        L.qgemul_plan_packed_layout.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
        L.qgemul_run_release.argtypes = []
        L.qgemul_run_release.restype = None
        L.qgemul_bitstream_bytes.argtypes = [vp, C.c_int]
        L.qgemul_bitstream_bytes.restype = i64
        L.qgemul_export_bitstream.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.qgemul_packed_e_bytes.argtypes = [vp, C.c_int]
        L.qgemul_packed_e_bytes.restype = i64
        L.qgemul_pack_e.argtypes = [vp, C.c_int, vp, i64, vp]
        L.qgemul_execute_ep.argtypes = [vp, vp, vp, vp, pa]
        L.qgemul_time_execute_ep.argtypes = [vp, vp, vp, vp, pa, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.qgemul_run_ep.argtypes = [pd, pe, vp, vp, vp, C.POINTER(vp), C.POINTER(qgemul_opts)]
        L.qgemul_comm_unique_id.argtypes = [vp]
        L.qgemul_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.qgemul_comm_destroy.argtypes = [vp]
        L.qgemul_comm_destroy.restype = None
        pi = C.POINTER(C.c_int)
        L.qgemul_comm_info.argtypes = [vp, pi, pi, pi]
        L.qgemul_gather_packed_c.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.c_int]
        L.qgemul_comm_fence.argtypes = [vp, C.c_int]
        L.qgemul_comm_sync.argtypes = [vp]
        L.qgemul_comm_barrier.argtypes = [vp]
        L.qgemul_comm_max_f64.argtypes = [vp, C.POINTER(C.c_double)]
        L.qgemul_ctx_device.argtypes = [vp]
        _lib = L
    return _lib


def _chk(st: int, what: str = ""):
    if st != QG_OK:
        raise QgemulError(st, what)


def classify(desc: qgemul_desc, flags: int = 0) -> qgemul_info:
    info = qgemul_info()
    _chk(lib().qgemul_classify(C.byref(desc), flags, C.byref(info)), "qgemul_classify")
    return info


def run_release():
    """free the calling thread's qgemul_run cache (context, plan, device buffers)"""
    lib().qgemul_run_release()


def classify_ep_status(desc: qgemul_desc, ep, flags: int = 0):
    info = qgemul_info()
    fn = lib().qgemul_classify_epc if isinstance(ep, qgemul_epilogue_cplx) else lib().qgemul_classify_ep
    st = fn(C.byref(desc), C.byref(ep), flags, C.byref(info))
    return st, info


def run_ep(desc: qgemul_desc, ep, D_out: np.ndarray, A: np.ndarray, B: np.ndarray, E, *, lda: int = 0,
           ldb: int = 0, ldc: int = 0, device: int = -1, flags: int = 0) -> np.ndarray:
    """qgemul_run_ep / qgemul_run_epc: E[k] = host-layout tensor (tight, column-major flattening) or a 1-element array for a
    scalar stage ({re, im} structured elements for a complex operand)."""
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    E = [np.ascontiguousarray(e) for e in E]
    ptrs = (C.c_void_p * max(1, len(E)))(*[e.ctypes.data for e in E])
    o = qgemul_opts(lda, ldb, ldc, device, flags)
    fn = lib().qgemul_run_epc if isinstance(ep, qgemul_epilogue_cplx) else lib().qgemul_run_ep
    _chk(fn(C.byref(desc), C.byref(ep), D_out.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
            B.ctypes.data_as(C.c_void_p), ptrs, C.byref(o)), "qgemul_run_ep")
    return D_out


def classify_status(desc: qgemul_desc, flags: int = 0):
    info = qgemul_info()
    st = lib().qgemul_classify(C.byref(desc), flags, C.byref(info))
    return st, info


def run(desc: qgemul_desc, C_out: np.ndarray, A: np.ndarray, B: np.ndarray, *, lda: int = 0, ldb: int = 0,
        ldc: int = 0, device: int = -1, flags: int = 0) -> np.ndarray:
    """qgemul_run on host-layout numpy buffers (see oracle.qoracle.host_dtype for the element dtype)."""
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    assert C_out.flags["C_CONTIGUOUS"]
    o = qgemul_opts(lda, ldb, ldc, device, flags)
    _chk(lib().qgemul_run(C.byref(desc), C_out.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                          B.ctypes.data_as(C.c_void_p), C.byref(o)), "qgemul_run")
    return C_out


def run_sharded(desc: qgemul_desc, C_out: np.ndarray, A: np.ndarray, B: np.ndarray, devices, *, lda: int = 0, ldb: int = 0,
                ldc: int = 0, flags: int = 0) -> np.ndarray:
    """qgemul_run_sharded: one process, the rows of C in bands over `devices` (an ordinal may repeat)."""
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    assert C_out.flags["C_CONTIGUOUS"]
    o = qgemul_opts(lda, ldb, ldc, -1, flags)
    devs = (C.c_int * len(devices))(*devices)
    L = lib()
    L.qgemul_run_sharded.argtypes = [C.POINTER(qgemul_desc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(qgemul_opts), C.POINTER(C.c_int), C.c_int]
    _chk(L.qgemul_run_sharded(C.byref(desc), C_out.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                              B.ctypes.data_as(C.c_void_p), C.byref(o), devs, len(devices)), "qgemul_run_sharded")
    return C_out


class Context:
    def __init__(self, device: int = -1):
        self.h = C.c_void_p()
        _chk(lib().qgemul_ctx_create(device, C.byref(self.h)), "qgemul_ctx_create")

    def close(self):
        if self.h:
            lib().qgemul_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def sync(self):
        _chk(lib().qgemul_ctx_sync(self.h), "qgemul_ctx_sync")

    @property
    def stream(self) -> int:
        return lib().qgemul_ctx_stream(self.h)

    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _chk(lib().qgemul_dev_alloc(self.h, nbytes, C.byref(p)), "qgemul_dev_alloc")
        return p.value

    def free(self, ptr: int):
        _chk(lib().qgemul_dev_free(self.h, C.c_void_p(ptr)), "qgemul_dev_free")

    def h2d(self, dst: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _chk(lib().qgemul_memcpy_h2d(self.h, C.c_void_p(dst), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")

    def d2h(self, arr: np.ndarray, src: int):
        _chk(lib().qgemul_memcpy_d2h(self.h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(src), arr.nbytes), "d2h")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Comm:
    """The library-owned RCCL communicator of one rank (include/qgemul.h, qgemul_comm_*): the gather of packed C bands."""
    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _chk_rccl(lib().qgemul_comm_unique_id(buf), "qgemul_comm_unique_id")
        return buf.raw

    def __init__(self, ctx: "Context", nranks: int, rank: int, unique_id: bytes):
        assert len(unique_id) == Comm.ID_BYTES
        self.h = C.c_void_p()
        self.nranks, self.rank = nranks, rank
        self._id = C.create_string_buffer(unique_id, Comm.ID_BYTES)
        _chk_rccl(lib().qgemul_comm_create(ctx.h, nranks, rank, self._id, C.byref(self.h)), "qgemul_comm_create")

    def info(self):
        """(ranks per ncclCommCount, this rank per ncclCommUserRank, ncclGetVersion code)"""
        n, r, v = C.c_int(), C.c_int(), C.c_int()
        _chk_rccl(lib().qgemul_comm_info(self.h, C.byref(n), C.byref(r), C.byref(v)), "qgemul_comm_info")
        return n.value, r.value, v.value

    def gather(self, send_ptr: int, send_bytes: int, recv_ptrs=None, recv_bytes=None, root: int = 0, slot: int = 0):
        rp = rb = None
        if recv_ptrs is not None:
            rp = (C.c_void_p * self.nranks)(*[p or None for p in recv_ptrs])
            rb = (C.c_size_t * self.nranks)(*recv_bytes)
        _chk_rccl(lib().qgemul_gather_packed_c(self.h, C.c_void_p(send_ptr), send_bytes, rp, rb, root, slot), "qgemul_gather_packed_c")

    def fence(self, slot: int = -1):
        _chk_rccl(lib().qgemul_comm_fence(self.h, slot), "qgemul_comm_fence")

    def sync(self):
        _chk_rccl(lib().qgemul_comm_sync(self.h), "qgemul_comm_sync")

    def barrier(self):
        _chk_rccl(lib().qgemul_comm_barrier(self.h), "qgemul_comm_barrier")

    def max_f64(self, v: float) -> float:
        x = C.c_double(v)
        _chk_rccl(lib().qgemul_comm_max_f64(self.h, C.byref(x)), "qgemul_comm_max_f64")
        return x.value

    def close(self):
        if self.h:
            lib().qgemul_comm_destroy(self.h)
            self.h = C.c_void_p()


def _chk_rccl(st: int, what: str):
    if st == QG_ERCCL:
        raise QgemulError(st, f"{what} (ncclResult {lib().qgemul_last_rccl_error()})")
    _chk(st, what)


class Plan:
    def __init__(self, ctx: Context, desc: qgemul_desc, flags: int = 0, epilogue=None):
        self.ctx = ctx
        self.desc = desc
        self.epilogue = epilogue
        self.h = C.c_void_p()
        if epilogue is None:
            _chk(lib().qgemul_plan_create(ctx.h, C.byref(desc), flags, C.byref(self.h)), "qgemul_plan_create")
        elif isinstance(epilogue, qgemul_epilogue_cplx):
            _chk(lib().qgemul_plan_create_epc(ctx.h, C.byref(desc), C.byref(epilogue), flags, C.byref(self.h)), "qgemul_plan_create_epc")
        else:
            _chk(lib().qgemul_plan_create_ep(ctx.h, C.byref(desc), C.byref(epilogue), flags, C.byref(self.h)), "qgemul_plan_create_ep")
        self.info = qgemul_info()
        _chk(lib().qgemul_plan_info(self.h, C.byref(self.info)), "qgemul_plan_info")

    def close(self):
        if self.h:
            lib().qgemul_plan_destroy(self.h)
            self.h = C.c_void_p()

    def pack(self, operand: int, src_dev: int, packed_dev: int, ld: int = 0):
        _chk(lib().qgemul_pack(self.h, operand, C.c_void_p(src_dev), ld, C.c_void_p(packed_dev)), "qgemul_pack")

    def pack_f64(self, operand: int, src_dev: int, packed_dev: int, ld: int = 0):
        _chk(lib().qgemul_pack_f64(self.h, operand, C.c_void_p(src_dev), ld, C.c_void_p(packed_dev)), "qgemul_pack_f64")

    def fill(self, operand: int, seed: int, dist: int, packed_dev: int):
        _chk(lib().qgemul_fill_packed(self.h, operand, seed, dist, C.c_void_p(packed_dev)), "qgemul_fill_packed")

    def execute(self, pC: int, pA: int, pB: int):
        _chk(lib().qgemul_execute(self.h, C.c_void_p(pC), C.c_void_p(pA), C.c_void_p(pB)), "qgemul_execute")

    def execute_host_c(self, C_dev: int, pA: int, pB: int, ldc: int = 0):
        """packed A, packed B -> C in the reference layout on the device (no packed C where the kernel can store it directly)"""
        L = lib()
        L.qgemul_execute_host_c.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _chk(L.qgemul_execute_host_c(self.h, C.c_void_p(C_dev), ldc, C.c_void_p(pA), C.c_void_p(pB)), "qgemul_execute_host_c")

    @property
    def stores_host_c(self) -> bool:
        L = lib()
        L.qgemul_plan_stores_host_c.argtypes = [C.c_void_p]
        return bool(L.qgemul_plan_stores_host_c(self.h))

    def bitstream_bytes(self, fmt: int = 0) -> int:
        return int(lib().qgemul_bitstream_bytes(self.h, fmt))

    def export_bitstream(self, pC: int, out_dev: int, tensor_chunk: int = 0, elem_chunk: int = 0, fmt: int = 0):
        _chk(lib().qgemul_export_bitstream(self.h, C.c_void_p(pC), tensor_chunk, elem_chunk, fmt, C.c_void_p(out_dev)),
             "qgemul_export_bitstream")

    def packed_layout(self, operand):
        """(trailer offset, row-sum offset, padded rows, centre) of a packed operand of the linear class (include/qgemul.h)"""
        out = (C.c_int64 * 4)()
        _chk(lib().qgemul_plan_packed_layout(self.h, operand, out), "qgemul_plan_packed_layout")
        return tuple(int(x) for x in out)

    def fuses_epilogue(self) -> bool:
        return bool(lib().qgemul_plan_fuses_epilogue(self.h))

    def packed_e_bytes(self, stage: int) -> int:
        return int(lib().qgemul_packed_e_bytes(self.h, stage))

    def pack_e(self, stage: int, src_dev: int, packed_dev: int, ld: int = 0):
        _chk(lib().qgemul_pack_e(self.h, stage, C.c_void_p(src_dev), ld, C.c_void_p(packed_dev)), "qgemul_pack_e")

    @staticmethod
    def ep_args(packed=(), scalars=(), scalars_im=()) -> qgemul_ep_args:
        a = qgemul_ep_args()
        for k, ptr in enumerate(packed):
            a.e_packed[k] = ptr or None
        for k, v in enumerate(scalars):
            a.e_scalar[k] = int(v or 0)
        for k, v in enumerate(scalars_im):
            a.e_scalar_im[k] = int(v or 0)
        return a

    def execute_ep(self, pD: int, pA: int, pB: int, args: qgemul_ep_args):
        _chk(lib().qgemul_execute_ep(self.h, C.c_void_p(pD), C.c_void_p(pA), C.c_void_p(pB), C.byref(args)), "qgemul_execute_ep")

    def time_execute_ep(self, pD: int, pA: int, pB: int, args: qgemul_ep_args, warmup: int, iters: int) -> float:
        ms = C.c_float()
        _chk(lib().qgemul_time_execute_ep(self.h, C.c_void_p(pD), C.c_void_p(pA), C.c_void_p(pB), C.byref(args), warmup, iters,
                                          C.byref(ms)), "qgemul_time_execute_ep")
        return ms.value

    def unpack_c(self, pC: int, dst_dev: int, ld: int = 0):
        _chk(lib().qgemul_unpack_c(self.h, C.c_void_p(pC), C.c_void_p(dst_dev), ld), "qgemul_unpack_c")

    def time_execute(self, pC: int, pA: int, pB: int, warmup: int, iters: int) -> float:
        ms = C.c_float()
        _chk(lib().qgemul_time_execute(self.h, C.c_void_p(pC), C.c_void_p(pA), C.c_void_p(pB), warmup, iters,
                                       C.byref(ms)), "qgemul_time_execute")
        return ms.value
