"""Row-sharded Qgemul across the GPUs of one node, one process per GPU.

The M*N outputs are independent, so rank r owns rows [row0_r, row0_r + rows_r) of C and needs the
matching rows of A' plus all of B.  The only exchange step is ONE gather of the C row blocks to
rank 0 (torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU for the
tests).  No other collective touches the data path.

Default engine path (`compute` not given): everything between the host operands and the host result
stays on the device and in the engine's PACKED layout — pack the band of A and B, GEMM, gather the
packed C bands (info.packed_bytes[2]: 1-byte containers for configuration 4, a quarter of the
host-layout bytes), and rank 0 unpacks every band straight into the one host-layout C with the band's
row offset and the full leading dimension.  The collective runs on the engine's own stream order
(the engine's stream is torch's current stream): no host synchronisation between GEMM and gather.

`compute` replaces the per-rank engine call by a host function (tests inject the CPU oracle so that
the partition / gather / reassembly logic is covered with world_size-2 gloo runs on machines without
a GPU); that path gathers host-layout blocks.  (One PROCESS driving several GPUs: qgemul_run_sharded,
include/qgemul.h.)
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .desc import Elem, MulArgs, Qcomplex, host_layout, lower


def row_partition(M: int, world: int, align: int = 1) -> List[Tuple[int, int]]:
    """Contiguous row blocks (row0, rows) for each rank; block starts are multiples of `align`
    (the MFMA path's 128/256-row packed tiles), sizes differ by at most one alignment unit."""
    if world < 1 or M < 0 or align < 1:
        raise ValueError("bad partition request")
    units = (M + align - 1) // align
    out, u0 = [], 0
    for r in range(world):
        u = units // world + (1 if r < units % world else 0)
        row0 = min(u0 * align, M)
        row1 = min((u0 + u) * align, M)
        out.append((row0, row1 - row0))
        u0 += u
    return out


def _host_dtype(e: Elem) -> np.dtype:
    size, off, (sr, si) = host_layout(e)
    if not isinstance(e, Qcomplex):
        return np.dtype("<i4" if sr == 4 else "<i8")
    return np.dtype({"names": ["re", "im"], "formats": ["<i4" if sr == 4 else "<i8", "<i4" if si == 4 else "<i8"],
                     "offsets": [0, off], "itemsize": size})


def _engine_row_sharded(A, B, ea, eb, ec, M, N, K, parts, rank, world, group, kw):
    """The default path: packed, device-resident, one gather of packed C bands, unpack on rank 0."""
    import torch
    import torch.distributed as dist
    from . import capi

    backend = dist.get_backend(group) if world > 1 else None
    on_host = world > 1 and backend != "nccl"
    local = torch.cuda.current_device()
    dev = torch.device("cuda", local)
    adt, bdt, cdt = _host_dtype(ea), _host_dtype(eb), _host_dtype(ec)
    transposed_a = kw.get("transposed_a", False)
    row0, rows = parts[rank]
    max_rows = max(p[1] for p in parts)
    with capi.Context(local) as ctx:
        prev = torch.cuda.current_stream(dev)
        torch.cuda.set_stream(torch.cuda.ExternalStream(ctx.stream, device=dev))
        try:
            plans = {}                        # band height -> plan (at most two heights in a partition)

            def plan_for(r):
                if r not in plans:
                    plans[r] = capi.Plan(ctx, lower(ea, eb, ec, r, N, K, **kw))
                return plans[r]

            pmax = plan_for(max_rows)
            cbytes = int(pmax.info.packed_bytes[2])          # every rank sends this many bytes (short bands are padded)
            tC = torch.zeros(cbytes, dtype=torch.uint8, device=dev)
            if rows > 0:
                p = plan_for(rows)
                pb = p.info.packed_bytes
                A = np.ascontiguousarray(A).view(adt).reshape(-1)
                B = np.ascontiguousarray(B).view(bdt).reshape(-1)
                if transposed_a:
                    a_band, lda = A[row0 * K:(row0 + rows) * K], K              # A is K x M: the band is a run of columns
                else:
                    a_band = np.ascontiguousarray(A.reshape(K, M)[:, row0:row0 + rows]).reshape(-1)   # strided rows -> tight band
                    lda = rows
                hA = torch.from_numpy(a_band.view(np.uint8)).to(dev)
                hB = torch.from_numpy(B.view(np.uint8)).to(dev)
                tA = torch.empty(int(pb[0]), dtype=torch.uint8, device=dev)
                tB = torch.empty(int(pb[1]), dtype=torch.uint8, device=dev)
                p.pack(capi.OPERAND_A, hA.data_ptr(), tA.data_ptr(), lda)
                p.pack(capi.OPERAND_B, hB.data_ptr(), tB.data_ptr(), K)
                p.execute(tC.data_ptr(), tA.data_ptr(), tB.data_ptr())
            if world > 1:
                src = tC.cpu() if on_host else tC
                gathered = [torch.empty(cbytes, dtype=torch.uint8, device="cpu" if on_host else dev) for _ in range(world)] if rank == 0 else None
                dist.gather(src, gathered, dst=0, group=group)            # the single collective on the path
            else:
                gathered = [tC]
            out = None
            if rank == 0:
                hC = torch.zeros(M * N * cdt.itemsize, dtype=torch.uint8, device=dev)
                for (r0, rr), t in zip(parts, gathered):
                    if rr == 0:
                        continue
                    t = t.to(dev)
                    plan_for(rr).unpack_c(t.data_ptr(), hC.data_ptr() + r0 * cdt.itemsize, M)   # row offset r0, leading dimension M
                out = hC.cpu().numpy().view(cdt).copy()
            torch.cuda.current_stream(dev).synchronize()
            for p in plans.values():
                p.close()
        finally:
            torch.cuda.set_stream(prev)
    return out


def qgemul_row_sharded(A: np.ndarray, B: np.ndarray, ea: Elem, eb: Elem, ec: Elem, M: int, N: int, K: int, *,
                       add_args: Optional[Sequence[Elem]] = None, mul_args: MulArgs = None, transposed_a: bool = False,
                       align: int = 256, group=None,
                       compute: Optional[Callable] = None) -> Optional[np.ndarray]:
    """Every rank passes the full host-layout A (column-major M x K, or K x M when transposed) and B.
    Returns the full column-major C on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    parts = row_partition(M, world, align)
    if compute is None:
        return _engine_row_sharded(A, B, ea, eb, ec, M, N, K, parts, rank, world, group,
                                   dict(add_args=add_args, mul_args=mul_args, transposed_a=transposed_a))
    row0, rows = parts[rank]
    cdt = _host_dtype(ec)
    max_rows = max(p[1] for p in parts)
    local = np.zeros(max_rows * N, dtype=cdt)  # padded to a common size for the gather
    if rows > 0:
        d = lower(ea, eb, ec, rows, N, K, add_args=add_args, mul_args=mul_args, transposed_a=transposed_a)
        adt = _host_dtype(ea)
        A = np.ascontiguousarray(A).view(adt).reshape(-1)
        if transposed_a:
            a_view, lda = A[row0 * K:], K           # A is K x M: rows of A' are columns, contiguous
        else:
            a_view, lda = A[row0:], M               # A is M x K column-major: the shard is strided
        c_shard = compute(d, a_view, B, ec, lda, K)
        local[:rows * N] = np.asarray(c_shard).view(cdt).reshape(-1)
    if world == 1:
        return local[:M * N].copy()
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t_local = torch.from_numpy(local.view(np.uint8)).to(dev)
    gathered = [torch.empty_like(t_local) for _ in range(world)] if rank == 0 else None
    dist.gather(t_local, gathered, dst=0, group=group)     # the single collective on the path
    if rank != 0:
        return None
    C = np.zeros(M * N, dtype=cdt)
    Cm = C.reshape(N, M)                                    # column-major: C[i + j*M] -> Cm[j, i]
    for (r0, rr), t in zip(parts, gathered):
        if rr == 0:
            continue
        blk = t.cpu().numpy().view(cdt)[:rr * N].reshape(N, rr)
        Cm[:, r0:r0 + rr] = blk
    return C
