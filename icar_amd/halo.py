"""Halo topology of a tile and its transport: the coarray PUT + `sync images` pattern of
src/objects/exchangeable_obj.f90:138-356 re-expressed as one batched message per neighbour.

For a device tile (domain_t) the transport lives BEHIND THE C ABI (icar_amd/csrc/comm.hip): HaloComm.attach()
hands the library the neighbour ranks and initialises its communicator -- RCCL over xGMI when torch.distributed runs on
"nccl" (the unique id travels through one broadcast), the host-staged shared-memory path when it runs on "gloo" (several
images sharing one GPU: a functional check) -- and domain_t.halo_send / halo_retrieve / update_dt call
icar_hip_halo_send / icar_hip_halo_retrieve / icar_hip_update_dt.  torch.distributed is only the launcher's rendezvous.

The send() / retrieve() methods below drive a HOST-array double of the tile (tests/host_tile.py) over torch.distributed
P2P: the CPU-side statement of the same exchange, which the world-size-2/4 gloo tests and the tiled CPU oracle runs use.
The tile object only has to provide
    halo_count(dir, halo) -> elements per field
    halo_pack(dir, halo, field_ids, buffer) / halo_unpack(dir, halo, field_ids, buffer)
    new_buffer(n) -> 1-D float32 torch tensor
exchange_uv() (iterative_winds' exchange_u / exchange_v, per forcing step, not on the sub-step path) serves both kinds of tile.
"""
import torch
import torch.distributed as dist

DIR_NORTH, DIR_SOUTH, DIR_EAST, DIR_WEST = 0, 1, 2, 3
_OPPOSITE = {DIR_NORTH: DIR_SOUTH, DIR_SOUTH: DIR_NORTH, DIR_EAST: DIR_WEST, DIR_WEST: DIR_EAST}
_NAMES = {DIR_NORTH: "north", DIR_SOUTH: "south", DIR_EAST: "east", DIR_WEST: "west"}


def _pack_all(tile, dirs, halo, field_ids, bufs):
    """every direction in one launch where the tile can (device tiles), one call per direction otherwise (host double)"""
    if hasattr(tile, "halo_pack_many") and len(dirs) > 1:
        tile.halo_pack_many(dirs, halo, field_ids, bufs)
    else:
        for d, b in zip(dirs, bufs):
            tile.halo_pack(d, halo, field_ids, b)


def _unpack_all(tile, dirs, halo, field_ids, bufs):
    """N/S rows and E/W columns both cover the corner cells; the reference retrieves N, S, E, W in that order
    (exchangeable_obj.f90:138-151), so E/W win there.  The one-launch kernel reproduces that (its rows skip the corners
    an E/W message of the same call fills); separate launches run in the same order."""
    if hasattr(tile, "halo_unpack_many") and len(dirs) > 1:
        tile.halo_unpack_many(dirs, halo, field_ids, bufs)
    else:
        for d, b in sorted(zip(dirs, bufs), key=lambda t: t[0]):
            tile.halo_unpack(d, halo, field_ids, b)


class HaloComm:
    def __init__(self, grid, image, group=None, halo=None, loopback=False):
        """loopback=True: edges WITHOUT a neighbouring image wrap around to the tile's own opposite edge (a periodic
        domain; what src/tests/test_mpdata.f90 does by hand): same pack / unpack kernels, no transport."""
        self.grid = grid
        self.image = image                      # 1-based, = rank + 1
        self.group = group
        self.halo = grid.halo_size if halo is None else halo
        nb = grid.neighbors(image)
        # direction -> neighbour rank (0-based) ; boundaries have no entry
        self.peers = {d: nb[_NAMES[d]] - 1 for d in _NAMES if nb[_NAMES[d]] is not None}
        self.loop = [d for d in _NAMES if loopback and d not in self.peers and _OPPOSITE[d] not in self.peers]
        self._loopbuf = {}
        self._send = {}
        self._recv = {}
        self._hsend = {}
        self._hrecv = {}
        self._reqs = []
        self._nf = None
        self._stage = False
        self.transport_note = None

    def attach(self, domain):
        """Initialise the library's communicator of a device tile (icar_hip_comm_init / _init_host): collective over the
        images.  Edges that wrap (loopback) become ICAR_NEIGHBOR_SELF, boundaries ICAR_NEIGHBOR_NONE."""
        import ctypes
        from .capi import lib, check, NEIGHBOR_NONE, NEIGHBOR_SELF
        nb = [self.peers[d] if d in self.peers else (NEIGHBOR_SELF if d in self.loop else NEIGHBOR_NONE) for d in (0, 1, 2, 3)]
        arr = (ctypes.c_int * 4)(*nb)
        # a process group of ONE rank over "nccl" still gets the RCCL communicator (the same ncclCommInitRank / ncclAllReduce an
        # N-rank run makes: tests/test_gpu_bench_ranks.py runs bench.py that way on the one-GPU box)
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or dist.get_backend(self.group) == "nccl")
        if not multi:
            if self.peers:
                raise RuntimeError("HaloComm: neighbouring images but no torch.distributed process group")
            check(lib().icar_hip_comm_init(domain.ctx, 1, 0, None, arr), "icar_hip_comm_init")
            return
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        self.transport_note = None
        if dist.get_backend(self.group) == "nccl":
            # RCCL.  Every image must end up with the SAME transport, so the outcome of the initialisation is agreed on over the
            # launcher's own process group; if any image could not open its communicator (no librccl, a refused device ...) all of
            # them raise -- or, with ICAR_ALLOW_HOST_STAGED=1, all of them fall back to the host-staged transport with a warning
            # (bench.py refuses to time such a run either way).
            uid = ctypes.create_string_buffer(128)
            err = None
            try:
                if rank == 0:
                    check(lib().icar_hip_comm_unique_id(uid), "icar_hip_comm_unique_id")
            except RuntimeError as e:
                err = str(e)
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).to(f"cuda:{domain.device}")
            dist.broadcast(t, 0, group=self.group)                      # co_broadcast(uid, 1) in a coarray host
            if err is None and bool(t.any().item()):
                try:
                    check(lib().icar_hip_comm_init(domain.ctx, world, rank, bytes(t.cpu().numpy().tobytes()), arr), "icar_hip_comm_init")
                except RuntimeError as e:
                    err = str(e)
            elif err is None:
                err = "image 1 could not make an RCCL unique id"
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=f"cuda:{domain.device}")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 1:
                return
            lib().icar_hip_comm_destroy(domain.ctx)
            import os, sys
            why = err or "another image failed"
            # every image takes this branch together (the all_reduce above): they all raise, or they all degrade
            if os.environ.get("ICAR_ALLOW_HOST_STAGED") != "1":
                raise RuntimeError("HaloComm: the RCCL communicator is not available on every image (%s).  The host-staged "
                                   "transport (POSIX shared memory, ~100x slower, one node only) is a functional path, not a "
                                   "substitute: set ICAR_ALLOW_HOST_STAGED=1 to run on it anyway" % why)
            self.transport_note = "RCCL communicator not available on every image (%s): host-staged transport (ICAR_ALLOW_HOST_STAGED=1)" % why
            sys.stderr.write("icar_amd WARNING [image %d]: %s\n" % (rank + 1, self.transport_note))
        import os
        name = [f"icar_hip_{os.getpid()}_{id(self) & 0xffffff:x}" if rank == 0 else None]
        dist.broadcast_object_list(name, 0, group=self.group)
        dev = f"cuda:{domain.device}" if dist.get_backend(self.group) == "nccl" else "cpu"
        need = torch.tensor([max(domain.halo_count(d, self.halo) for d in (0, 1, 2, 3)) * 4 * 11], dtype=torch.int64, device=dev)
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=self.group)
        check(lib().icar_hip_comm_init_host(domain.ctx, world, rank, name[0].encode(), int(need.item()), arr), "icar_hip_comm_init_host")

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
        if self.loop and field_ids:
            if self._loopbuf.get("nf") != len(field_ids):
                self._loopbuf = {d: tile.new_buffer(tile.halo_count(d, self.halo) * len(field_ids)) for d in self.loop}
                self._loopbuf["nf"] = len(field_ids)
            _pack_all(tile, self.loop, self.halo, field_ids, [self._loopbuf[d] for d in self.loop])
        if not self.peers or not field_ids:
            return
        self._buffers(tile, len(field_ids))
        ops = []
        _pack_all(tile, list(self.peers), self.halo, field_ids, [self._send[d] for d in self.peers])
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
        if not field_ids:
            return
        # ONE unpack for everything that arrived -- wrapped edges (my north edge is what arrives from the south, etc.) and the
        # peers' messages -- so that the corner cells follow the reference's N, S, E, W retrieve order whatever mix of
        # wrapping and neighbouring edges a tile has (exchangeable_obj.f90:138-151)
        dirs, bufs = [], []
        if self.loop:
            dirs += [_OPPOSITE[d] for d in self.loop]; bufs += [self._loopbuf[d] for d in self.loop]
        if self.peers:
            for r in self._reqs:
                r.wait()
            self._reqs = []
            if getattr(self, "_host_sync", False):
                torch.cuda.current_stream().synchronize()   # r.wait() only blocks torch's stream; unpack runs on the context's
            if self._stage:
                for d in self.peers:
                    self._recv[d].copy_(self._hrecv[d])
                torch.cuda.synchronize()                    # the copies ran on torch's stream, unpack runs on the context's
            dirs += list(self.peers); bufs += [self._recv[d] for d in self.peers]
        if dirs:
            order = sorted(range(len(dirs)), key=lambda n: dirs[n])       # N, S, E, W
            _unpack_all(tile, [dirs[n] for n in order], self.halo, field_ids, [bufs[n] for n in order])

    # ---- exchange_u / exchange_v (exchangeable_obj.f90:158-229) -----------------------------------------------
    def exchange_uv(self, tile, u_field, v_field, which=0):
        """`call domain%u%exchange_u(); call domain%v%exchange_v()` (wind.f90:404-405, :482-483) as ONE message per
        neighbour.  The two fields are independent, so batching them is equivalent; all boxes are packed before any
        is unpacked (PUTs before `sync images`), and N/S are unpacked before E/W like the reference's retrieve order.
        The tile provides box_pack / box_unpack (field, which, i0, ni, j0, nj, buffer) and dims via tile.nx/nz/ny."""
        if not self.peers:
            return
        plan = staggered_boxes(tile.nx, tile.ny, self.halo)
        nz = tile.nz
        flds = {"u": u_field, "v": v_field}
        key = ("uv", tile.nx, tile.ny, nz)
        if getattr(self, "_uvkey", None) != key:
            size = lambda boxes: sum(b[2] * b[4] * nz for b in boxes)
            self._uvsend = {d: tile.new_buffer(size(plan[d][0])) for d in self.peers}
            self._uvrecv = {d: tile.new_buffer(size(plan[d][1])) for d in self.peers}
            self._uvkey = key
        def walk(d, boxes, buf, fn):
            off = 0
            for kind, i0, ni, j0, nj in boxes:
                n = ni * nj * nz
                fn(flds[kind], which, i0, ni, j0, nj, buf[off:off + n])
                off += n
        for d in self.peers:
            walk(d, plan[d][0], self._uvsend[d], tile.box_pack)
        self._transfer(tile, self._uvsend, self._uvrecv)
        for d in sorted(self.peers):                       # 0,1 = north, south first; then east, west
            walk(d, plan[d][1], self._uvrecv[d], tile.box_unpack)

    def _transfer(self, tile, send, recv):
        """Blocking neighbour exchange of already packed buffers (same transport rules as send()/retrieve())."""
        any_buf = next(iter(send.values()))
        stage = any_buf.is_cuda and dist.get_backend(self.group) == "gloo"
        host_sync = any_buf.is_cuda and (stage or bool(getattr(tile, "needs_host_sync", lambda: False)()))
        if host_sync:
            tile.synchronize()
        sbuf, rbuf = send, recv
        if stage:
            sbuf = {d: b.cpu() for d, b in send.items()}
            rbuf = {d: torch.empty(b.shape, dtype=b.dtype) for d, b in recv.items()}
        ops = []
        for d in sorted(self.peers):
            ops.append(dist.P2POp(dist.isend, sbuf[d], self.peers[d], group=self.group))
            ops.append(dist.P2POp(dist.irecv, rbuf[d], self.peers[d], group=self.group))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if stage:
            for d in self.peers:
                recv[d].copy_(rbuf[d])
        if host_sync:
            torch.cuda.synchronize()


def staggered_boxes(nx, ny, h):
    """Index table of exchange_u + exchange_v for a tile of nx x ny mass cells (memory extents, halos included).
    Returns {dir: (send_boxes, recv_boxes)}; a box is (kind, i0, ni, j0, nj), 0-based, kind "u" (nx+1 columns) or "v"
    (ny+1 rows).  recv_boxes[dir] is where the message FROM the neighbour in direction dir lands.
      u: N/S like exchange (put_north/put_south over the full staggered width, :160-161);
         east PUT is halo+1 columns n-2h..n-h (:166), landing in the neighbour's columns start..start+h (:188);
         west PUT is columns start+h+1..start+2h (:172), landing in the neighbour's last h columns (retrieve_east_halo)
      v: E/W like exchange; north PUT rows n-2h..n-h (:202) -> rows start..start+h (:221); south PUT rows
         start+h+1..start+2h (:207) -> the neighbour's last h rows (retrieve_north_halo)."""
    X, Y = nx + 1, ny + 1
    return {
        DIR_NORTH: ([("u", 0, X, ny - 2 * h, h), ("v", 0, nx, Y - 1 - 2 * h, h + 1)],
                    [("u", 0, X, ny - h, h), ("v", 0, nx, Y - h, h)]),
        DIR_SOUTH: ([("u", 0, X, h, h), ("v", 0, nx, h + 1, h)],
                    [("u", 0, X, 0, h), ("v", 0, nx, 0, h + 1)]),
        DIR_EAST: ([("u", X - 1 - 2 * h, h + 1, 0, ny), ("v", nx - 2 * h, h, 0, Y)],
                   [("u", X - h, h, 0, ny), ("v", nx - h, h, 0, Y)]),
        DIR_WEST: ([("u", h + 1, h, 0, ny), ("v", h, h, 0, Y)],
                   [("u", 0, h + 1, 0, ny), ("v", 0, h, 0, Y)]),
    }


def co_min(value, group=None, device=None):
    """time_step.f90:413 `call co_min(seconds)`: all-reduce(min) of one REAL(8)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    if device is None:
        if dist.get_backend(group) == "nccl":           # RCCL reduces device memory only: the caller names its tile's device (domain.device)
            raise ValueError("co_min over RCCL needs device=domain.device (the product path is domain_t.co_min -> icar_hip_co_min)")
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())
