"""Row-sharded Qgemul across the GPUs of one node.

The M*N outputs are independent, so rank r owns rows [row0_r, row0_r + rows_r) of C and needs the
matching rows of A' plus all of B.  The only exchange step is ONE gather of the C row blocks to
rank 0 (torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU for the
tests).  No other collective touches the data path.

`compute` is the per-rank engine call.  The default is the HIP engine through the C-ABI
(qublas_amd.capi.run); tests inject the CPU oracle so the partition / gather / reassembly logic is
covered with world_size-2 gloo runs on machines without a GPU.
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


def _default_compute(desc, A, B, c_elem, lda, ldb):
    from . import capi
    out = np.zeros(desc.M * desc.N, dtype=_host_dtype(c_elem))
    return capi.run(desc, out, A, B, lda=lda, ldb=ldb)


def qgemul_row_sharded(A: np.ndarray, B: np.ndarray, ea: Elem, eb: Elem, ec: Elem, M: int, N: int, K: int, *,
                       add_args: Optional[Sequence[Elem]] = None, mul_args: MulArgs = None, transposed_a: bool = False,
                       align: int = 128, group=None,
                       compute: Callable = _default_compute) -> Optional[np.ndarray]:
    """Every rank passes the full host-layout A (column-major M x K, or K x M when transposed) and B.
    Returns the full column-major C on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    parts = row_partition(M, world, align)
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
