"""Halo transport between tiles: the coarray PUT + `sync images` pattern of
src/objects/exchangeable_obj.f90:138-356 re-expressed as batched neighbour send/recv.

One message per neighbour per step carries ALL exchanged scalars (the reference issues one PUT per
variable per direction; at 512x512x40 on 2x2 those are 41 kB each and latency-bound, SURVEY.md
section 5).  Transport is torch.distributed P2P: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.  RCCL runs the transfers on its own stream, so the interior microphysics issued
between send() and retrieve() overlaps them exactly as time_step.f90:512-526 orders it.

The tile object only has to provide
    halo_count(dir, halo) -> elements per field
    halo_pack(dir, halo, field_ids, buffer) / halo_unpack(dir, halo, field_ids, buffer)
    new_buffer(n) -> 1-D float32 torch tensor on the tile's device
domain_t implements them with the HIP pack/unpack kernels; tests use a host-array double.
"""
import torch
import torch.distributed as dist

DIR_NORTH, DIR_SOUTH, DIR_EAST, DIR_WEST = 0, 1, 2, 3
_OPPOSITE = {DIR_NORTH: DIR_SOUTH, DIR_SOUTH: DIR_NORTH, DIR_EAST: DIR_WEST, DIR_WEST: DIR_EAST}
_NAMES = {DIR_NORTH: "north", DIR_SOUTH: "south", DIR_EAST: "east", DIR_WEST: "west"}


class HaloComm:
    def __init__(self, grid, image, group=None, halo=None):
        self.grid = grid
        self.image = image                      # 1-based, = rank + 1
        self.group = group
        self.halo = grid.halo_size if halo is None else halo
        nb = grid.neighbors(image)
        # direction -> neighbour rank (0-based) ; boundaries have no entry
        self.peers = {d: nb[_NAMES[d]] - 1 for d in _NAMES if nb[_NAMES[d]] is not None}
        self._send = {}
        self._recv = {}
        self._hsend = {}
        self._hrecv = {}
        self._reqs = []
        self._nf = None
        self._stage = False

    def _buffers(self, tile, nfields):
        if self._nf != nfields:
            self._send = {d: tile.new_buffer(tile.halo_count(d, self.halo) * nfields) for d in self.peers}
            self._recv = {d: tile.new_buffer(tile.halo_count(d, self.halo) * nfields) for d in self.peers}
            self._nf = nfields
            # gloo moves host memory only: device buffers are staged through pinned host tensors (what a coarray /
            # MPI host without GPU-aware transport does).  RCCL takes the device buffers.
            any_buf = next(iter(self._send.values()), None)
            self._stage = (any_buf is not None and any_buf.is_cuda and dist.get_backend(self.group) == "gloo")
            if self._stage:
                self._hsend = {d: torch.empty(b.shape, dtype=b.dtype).pin_memory() for d, b in self._send.items()}
                self._hrecv = {d: torch.empty(b.shape, dtype=b.dtype).pin_memory() for d, b in self._recv.items()}

    def send(self, tile, field_ids):
        """exchangeable%send for every variable: pack my edge planes, post send+recv per neighbour."""
        if not self.peers or not field_ids:
            return
        self._buffers(tile, len(field_ids))
        ops = []
        for d, peer in self.peers.items():
            tile.halo_pack(d, self.halo, field_ids, self._send[d])
        sbuf, rbuf = self._send, self._recv
        self._host_sync = bool(getattr(tile, "needs_host_sync", lambda: False)()) and not self._stage
        if self._host_sync:
            tile.synchronize()                          # the pack kernels must be complete before RCCL reads the buffers
        if self._stage:
            if hasattr(tile, "synchronize"):
                tile.synchronize()                      # pack kernels run on the context's stream
            for d in self.peers:
                self._hsend[d].copy_(self._send[d])
            sbuf, rbuf = self._hsend, self._hrecv
        # deterministic global order of the P2P list avoids cross-rank deadlock in batch mode
        for d in sorted(self.peers):
            peer = self.peers[d]
            ops.append(dist.P2POp(dist.isend, sbuf[d], peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, rbuf[d], peer, group=self.group))
        self._reqs = dist.batch_isend_irecv(ops)

    def retrieve(self, tile, field_ids):
        """exchangeable%retrieve: `sync images(neighbors)` == wait for the posted transfers,
        then copy each inbox into the halo planes facing that neighbour."""
        if not self.peers or not field_ids:
            return
        for r in self._reqs:
            r.wait()
        self._reqs = []
        if getattr(self, "_host_sync", False):
            torch.cuda.current_stream().synchronize()   # r.wait() only blocks torch's stream; unpack runs on the context's
        for d in self.peers:
            if self._stage:
                self._recv[d].copy_(self._hrecv[d])
                torch.cuda.synchronize()                # the copy ran on torch's stream, unpack runs on the context's
            tile.halo_unpack(d, self.halo, field_ids, self._recv[d])


def co_min(value, group=None, device=None):
    """time_step.f90:413 `call co_min(seconds)`: all-reduce(min) of one REAL(8)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())
