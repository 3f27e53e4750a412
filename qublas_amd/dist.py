"""Row-sharded Qgemul across the GPUs of one node, one process per GPU.

The M*N outputs are independent, so rank r owns rows [row0_r, row0_r + rows_r) of C and needs the matching rows of A' plus all
of B.  The only exchange step is ONE gather of the packed C bands to rank 0 — the LIBRARY's RCCL gather
(include/qgemul.h: qgemul_comm_*, grouped ncclSend / ncclRecv over xGMI on the communicator's own stream).  No other collective
touches the data path, and no PyTorch: device memory comes from the engine (qgemul_dev_alloc), the communicator's 128-byte id
travels over a TCP socket on MASTER_ADDR:MASTER_PORT (or, under torch.distributed.run — whose agent owns that port — through
the agent's TCPStore, the one place torch is imported, and only for those 128 bytes).

Transports (what `qgemul_row_sharded` and bench.py use between GEMM and unpack):
  RcclTransport   the product path: capi.Comm
  HostTransport   rehearsal on a machine with fewer GPUs than ranks (RCCL refuses two ranks on one device) and on CPU: the
                  packed bands go through host memory and the TCP channel below.  Same partition, same packed bytes, same
                  reassembly.

`compute` replaces the per-rank engine call by a host function (tests inject the CPU oracle so that the partition / gather /
reassembly logic is covered with several ranks on machines without a GPU); that path gathers host-layout blocks.  (One PROCESS
driving several GPUs: qgemul_run_sharded, include/qgemul.h.)
"""
from __future__ import annotations

import os
import socket
import struct
import time
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
    wide = np.dtype([("lo", "<u8"), ("hi", "<i8")])
    part = lambda n: "<i4" if n == 4 else "<i8" if n == 8 else wide
    if not isinstance(e, Qcomplex):
        return np.dtype(part(sr))
    return np.dtype({"names": ["re", "im"], "formats": [part(sr), part(si)], "offsets": [0, off], "itemsize": size})


# --------------------------------------------------------------------------------------------------------- the control channel
class HostChannel:
    """World-size TCP star on (host, port): rank 0 listens, every other rank keeps one connection.  Carries the communicator id,
    and — for the HostTransport rehearsal — barriers, a max and the packed bands themselves.  Nothing here touches a GPU."""

    def __init__(self, rank: int, world: int, host: str, port: int, timeout: float = 120.0):
        self.rank, self.world = rank, world
        self.peers: List[Optional[socket.socket]] = [None] * world
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((host, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                self.peers[r] = c
            srv.close()
        else:
            t0 = time.time()
            while True:
                try:
                    c = socket.create_connection((host, port), timeout=5)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            c.settimeout(None)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.sendall(struct.pack("<i", rank))
            self.peers[0] = c

    @staticmethod
    def _recvn(c: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(min(n - len(buf), 1 << 22))
            if not chunk:
                raise ConnectionError("peer closed the control channel")
            buf += chunk
        return bytes(buf)

    def _send(self, c, payload: bytes):
        c.sendall(struct.pack("<q", len(payload)) + payload)

    def _recv(self, c) -> bytes:
        n = struct.unpack("<q", self._recvn(c, 8))[0]
        return self._recvn(c, n)

    def bcast(self, payload: Optional[bytes]) -> bytes:
        """rank 0's bytes to everybody"""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for c in self.peers[1:]:
                self._send(c, payload)
            return payload
        return self._recv(self.peers[0])

    def gather(self, payload: bytes) -> Optional[List[bytes]]:
        """everybody's bytes to rank 0 (a list on rank 0, None elsewhere)"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [self._recv(c) for c in self.peers[1:]]
        self._send(self.peers[0], payload)
        return None

    def barrier(self):
        self.gather(b"")
        self.bcast(b"x" if self.rank == 0 else None)

    def max_f64(self, v: float) -> float:
        vals = self.gather(struct.pack("<d", v))
        m = max(struct.unpack("<d", x)[0] for x in vals) if vals is not None else 0.0
        return struct.unpack("<d", self.bcast(struct.pack("<d", m) if self.rank == 0 else None))[0]

    def close(self):
        for c in self.peers:
            if c is not None:
                c.close()
        self.peers = [None] * self.world


def env_rendezvous():
    """(rank, world, local_rank, host, port) from the launcher's environment (torch.distributed.run's variables)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")),
            os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533")))


def share_comm_id(rank: int, world: int, make_id: Callable[[], bytes], channel: Optional[HostChannel] = None) -> bytes:
    """Rank 0's RCCL unique id on every rank.  Under torch.distributed.run the agent process owns MASTER_PORT (its TCPStore, which the
    workers join as clients: TORCHELASTIC_USE_AGENT_STORE): the id goes through that store — the only use of torch on this path, and
    only for these 128 bytes.  Otherwise over the TCP channel."""
    if world == 1:
        return make_id()
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True" and channel is None:
        store = _agent_store()
        key = "qugemm_rccl_id_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        if rank == 0:
            store.set(key, make_id())
        return bytes(store.get(key))
    assert channel is not None
    return channel.bcast(make_id() if rank == 0 else None)


# --------------------------------------------------------------------------------------------------------- transports
class RcclTransport:
    """The product path: the library's communicator (capi.Comm).  Gathers are asynchronous on the communicator's own stream."""
    name = "rccl"

    def __init__(self, ctx, comm):
        self.ctx, self.comm = ctx, comm     # (the communicator orders its gathers behind THIS context's stream)
        self.world, self.rank = comm.nranks, comm.rank

    def gather(self, send_ptr, send_bytes, recv_ptrs=None, recv_bytes=None, slot=0):
        self.comm.gather(send_ptr, send_bytes, recv_ptrs, recv_bytes, 0, slot)

    def fence(self, slot=-1):
        self.comm.fence(slot)

    def barrier(self):
        self.comm.barrier()

    def max_f64(self, v):
        return self.comm.max_f64(v)

    def reported_world(self):
        return self.comm.info()[0]          # ncclCommCount


class HostTransport:
    """Rehearsal: the same packed bands through host memory and the TCP channel (synchronous)."""
    name = "host"

    def __init__(self, ctx, channel: HostChannel):
        self.ctx, self.ch = ctx, channel
        self.world, self.rank = channel.world, channel.rank

    def gather(self, send_ptr, send_bytes, recv_ptrs=None, recv_bytes=None, slot=0):
        buf = np.empty(send_bytes, np.uint8)
        if send_bytes:
            self.ctx.d2h(buf, send_ptr)
        got = self.ch.gather(buf.tobytes())
        if got is not None:
            for r, b in enumerate(got):
                if recv_ptrs[r] and recv_bytes[r] and not (r == 0 and recv_ptrs[0] == send_ptr):
                    assert len(b) == recv_bytes[r], (r, len(b), recv_bytes[r])
                    self.ctx.h2d(recv_ptrs[r], np.frombuffer(b, np.uint8))

    def fence(self, slot=-1):
        pass

    def barrier(self):
        self.ctx.sync()
        self.ch.barrier()

    def max_f64(self, v):
        return self.ch.max_f64(v)

    def reported_world(self):
        return self.world


def _agent_store():
    """torch.distributed.run's agent store (it owns MASTER_PORT; the workers are its clients), or None without that launcher"""
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") != "True":
        return None
    from datetime import timedelta

    from torch.distributed import TCPStore
    _, world, _, host, port = env_rendezvous()
    return TCPStore(host, port, world, False, timedelta(seconds=300))


def channel_from_env(rank: int, world: int, host: str, port: int) -> HostChannel:
    """The TCP star of this job.  Under torch.distributed.run MASTER_PORT belongs to the agent: rank 0 then listens on a free port
    of its own and tells the others through the agent's store."""
    store = _agent_store() if world > 1 else None
    if store is not None:
        key = "qugemm_host_channel_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        if rank == 0:
            probe = socket.socket()
            probe.bind((host, 0))
            free = probe.getsockname()[1]
            probe.close()
            store.set(key, str(free))
        port = int(bytes(store.get(key)).decode())
    return HostChannel(rank, world, host, port)


def make_transport(ctx, backend: str = "rccl", channel: Optional[HostChannel] = None):
    """The transport of this rank from the launcher's environment.  backend "rccl" (product) or "host" (rehearsal)."""
    from . import capi
    rank, world, _, host, port = env_rendezvous()
    if backend == "host":
        return HostTransport(ctx, channel or channel_from_env(rank, world, host, port))
    if backend != "rccl":
        raise ValueError(f"unknown backend {backend!r}: rccl | host")
    ch = channel
    if world > 1 and ch is None and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") != "True":
        ch = HostChannel(rank, world, host, port)
    uid = share_comm_id(rank, world, capi.Comm.unique_id, ch)
    if ch is not None and channel is None:
        ch.close()
    return RcclTransport(ctx, capi.Comm(ctx, world, rank, uid))


# --------------------------------------------------------------------------------------------------------- the sharded call
def _engine_row_sharded(A, B, ea, eb, ec, M, N, K, parts, transport, device, kw):
    """The default path: packed, device-resident, one gather of packed C bands, unpack on rank 0.  Engine calls only."""
    from . import capi

    world = transport.world if transport is not None else 1
    rank = transport.rank if transport is not None else 0
    adt, bdt, cdt = _host_dtype(ea), _host_dtype(eb), _host_dtype(ec)
    transposed_a = kw.get("transposed_a", False)
    row0, rows = parts[rank]
    ctx = transport.ctx if transport is not None else None     # (the transport's context: its gathers follow that stream)
    own_ctx = ctx is None
    if own_ctx:
        ctx = capi.Context(device)
    bufs = []

    def alloc(n):
        p = ctx.alloc(max(int(n), 16))
        bufs.append(p)
        return p

    plans = {}                        # band height -> plan (at most two heights in a partition)

    def plan_for(r):
        if r not in plans:
            plans[r] = capi.Plan(ctx, lower(ea, eb, ec, r, N, K, **kw))
        return plans[r]

    try:
        band_bytes = [int(plan_for(rr).info.packed_bytes[2]) if rr else 0 for _, rr in parts]
        pC = alloc(band_bytes[rank])
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
            hA, hB = alloc(a_band.nbytes), alloc(B.nbytes)
            ctx.h2d(hA, a_band.view(np.uint8))
            ctx.h2d(hB, B.view(np.uint8))
            tA, tB = alloc(pb[0]), alloc(pb[1])
            p.pack(capi.OPERAND_A, hA, tA, lda)
            p.pack(capi.OPERAND_B, hB, tB, K)
            p.execute(pC, tA, tB)
        land = [pC] + [None] * (world - 1)
        if world > 1:
            if rank == 0:
                land = [pC] + [alloc(band_bytes[r]) if band_bytes[r] else None for r in range(1, world)]
                transport.gather(pC, band_bytes[0], land, band_bytes)     # the single collective on the path
            else:
                transport.gather(pC, band_bytes[rank])
            transport.fence()
        out = None
        if rank == 0:
            hC = alloc(M * N * cdt.itemsize)
            zero = np.zeros(M * N * cdt.itemsize, np.uint8)
            ctx.h2d(hC, zero)
            for (r0, rr), t in zip(parts, land):
                if rr == 0:
                    continue
                plan_for(rr).unpack_c(t, hC + r0 * cdt.itemsize, M)       # row offset r0, leading dimension M
            out = np.zeros(M * N, dtype=cdt)
            ctx.d2h(out, hC)
        ctx.sync()
        if world > 1:
            transport.barrier()       # (nobody frees a buffer a peer may still be sending from / into)
        return out
    finally:
        for p in plans.values():
            p.close()
        for b in bufs:
            ctx.free(b)
        if own_ctx:
            ctx.close()


def qgemul_row_sharded(A: np.ndarray, B: np.ndarray, ea: Elem, eb: Elem, ec: Elem, M: int, N: int, K: int, *,
                       add_args: Optional[Sequence[Elem]] = None, mul_args: MulArgs = None, transposed_a: bool = False,
                       align: int = 256, transport=None, device: int = 0, channel: Optional[HostChannel] = None,
                       compute: Optional[Callable] = None) -> Optional[np.ndarray]:
    """Every rank passes the full host-layout A (column-major M x K, or K x M when transposed) and B.
    Returns the full column-major C on rank 0, None elsewhere.
      engine path (default): `transport` = an RcclTransport / HostTransport (None: one rank);
      compute=...: host blocks from the given function, gathered over `channel` (tests on machines without a GPU)."""
    if compute is None:
        world = transport.world if transport is not None else 1
        parts = row_partition(M, world, align)
        return _engine_row_sharded(A, B, ea, eb, ec, M, N, K, parts, transport, device,
                                   dict(add_args=add_args, mul_args=mul_args, transposed_a=transposed_a))
    world = channel.world if channel is not None else 1
    rank = channel.rank if channel is not None else 0
    parts = row_partition(M, world, align)
    row0, rows = parts[rank]
    cdt = _host_dtype(ec)
    local = np.zeros(rows * N, dtype=cdt)
    if rows > 0:
        d = lower(ea, eb, ec, rows, N, K, add_args=add_args, mul_args=mul_args, transposed_a=transposed_a)
        adt = _host_dtype(ea)
        A = np.ascontiguousarray(A).view(adt).reshape(-1)
        if transposed_a:
            a_view, lda = A[row0 * K:], K           # A is K x M: rows of A' are columns, contiguous
        else:
            a_view, lda = A[row0:], M               # A is M x K column-major: the shard is strided
        c_shard = compute(d, a_view, B, ec, lda, K)
        local[:] = np.asarray(c_shard).view(cdt).reshape(-1)
    if world == 1:
        return local.copy()
    gathered = channel.gather(local.tobytes())      # the single exchange step of the path
    if rank != 0:
        return None
    C = np.zeros(M * N, dtype=cdt)
    Cm = C.reshape(N, M)                            # column-major: C[i + j*M] -> Cm[j, i]
    for (r0, rr), b in zip(parts, gathered):
        if rr == 0:
            continue
        Cm[:, r0:r0 + rr] = np.frombuffer(b, dtype=cdt).reshape(N, rr)
    return C
