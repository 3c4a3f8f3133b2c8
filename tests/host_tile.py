"""Host-array double of the tile interface icar_amd.halo.HaloComm drives (TEST ONLY).
Same face definitions / buffer layouts as the HIP pack kernels (icar_amd/csrc/capi.hip), written
independently from exchangeable_obj.f90:248-356, so the GPU test can compare the two."""
import numpy as np
import torch


class HostTile:
    def __init__(self, grid, fields, dqdt=None):
        self.grid = grid
        self.f = fields                 # {field_id: float32 array (ny, nz, nx)} with halos
        self.dq = dqdt or {}            # {field_id: dqdt_3d mirror}
        self.nx, self.nz, self.ny = grid.ime - grid.ims + 1, grid.kme - grid.kms + 1, grid.jme - grid.jms + 1

    def halo_count(self, d, h):
        return self.nx * self.nz * h if d in (0, 1) else h * self.nz * self.ny

    def new_buffer(self, n):
        return torch.empty(int(n), dtype=torch.float32)

    def _sel(self, d, h, unpack):
        ny, nx = self.ny, self.nx
        if d == 0: return (slice(ny - h, ny) if unpack else slice(ny - 2 * h, ny - h)), slice(None)
        if d == 1: return (slice(0, h) if unpack else slice(h, 2 * h)), slice(None)
        if d == 2: return slice(None), (slice(nx - h, nx) if unpack else slice(nx - 2 * h, nx - h))
        return slice(None), (slice(0, h) if unpack else slice(h, 2 * h))

    def halo_pack(self, d, h, field_ids, buf):
        sj, si = self._sel(d, h, False)
        out = [np.ascontiguousarray(self.f[fid][sj, :, si]).ravel() for fid in field_ids]
        buf.copy_(torch.from_numpy(np.concatenate(out)))

    def halo_unpack(self, d, h, field_ids, buf):
        sj, si = self._sel(d, h, True)
        per = self.halo_count(d, h)
        b = buf.numpy()
        for m, fid in enumerate(field_ids):
            tgt = self.f[fid][sj, :, si]
            tgt[...] = b[m * per:(m + 1) * per].reshape(tgt.shape)

    # staggered boxes (exchange_u / exchange_v): same [nj][nz][ni] layout as icar_hip_box_pack
    def box_pack(self, field, which, i0, ni, j0, nj, buf):
        a = (self.dq if which else self.f)[field]
        buf.copy_(torch.from_numpy(np.ascontiguousarray(a[j0:j0 + nj, :, i0:i0 + ni]).ravel()))

    def box_unpack(self, field, which, i0, ni, j0, nj, buf):
        a = (self.dq if which else self.f)[field]
        a[j0:j0 + nj, :, i0:i0 + ni] = buf.numpy().reshape(nj, self.nz, ni)
