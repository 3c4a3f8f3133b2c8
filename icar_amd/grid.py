"""grid_t: x-y tile index algebra (row G1). Integer-exact restatement of
src/objects/grid_obj.f90:39-255 (domain_decomposition, my_n, my_start, update_with_halos)."""
from dataclasses import dataclass
import numpy as np
from .constants import kDEFAULT_HALO_SIZE


def domain_decomposition(nx, ny, nimages, ratio=1.0):
    """grid_obj.f90:39-114.  REAL(4) arithmetic like the reference so ties break identically."""
    f = np.float32
    mult = f(ratio)

    def score(xsplit, ysplit):
        x = f(nx) / f(xsplit); y = f(ny) / f(ysplit)
        if y > mult * x:
            return abs(f(1) - (y / (mult * x)))
        return abs(f(1) - ((mult * x) / y))

    xs, ys = 1, nimages
    best = score(xs, ys)
    for i in range(nimages, 0, -1):
        if nimages % i == 0:
            ysplit, xsplit = i, nimages // i
            cur = score(xsplit, ysplit)
            if cur < best:
                best, xs, ys = cur, xsplit, ysplit
    return xs, ys


def my_n(n_global, me, nimg):           # grid_obj.f90:116-121
    return n_global // nimg + (1 if me <= n_global % nimg else 0)


def my_start(n_global, me, nimg):       # grid_obj.f90:128-138
    base_n = n_global // nimg
    return (me - 1) * base_n + min(me - 1, n_global % nimg) + 1


@dataclass
class grid_t:
    """Same member names as src/objects/grid_h.f90:10-31 (1-based, inclusive)."""
    nx_global: int = 0; ny_global: int = 0; nz: int = 0
    ximages: int = 1; yimages: int = 1; ximg: int = 1; yimg: int = 1
    ims: int = 1; ime: int = 1; jms: int = 1; jme: int = 1; kms: int = 1; kme: int = 1
    its: int = 1; ite: int = 1; jts: int = 1; jte: int = 1; kts: int = 1; kte: int = 1
    ids: int = 1; ide: int = 1; jds: int = 1; jde: int = 1; kds: int = 1; kde: int = 1
    nx: int = 0; ny: int = 0; halo_size: int = kDEFAULT_HALO_SIZE
    ns_halo_nx: int = 0; ew_halo_ny: int = 0; halo_nz: int = 0

    def set_grid_dimensions(self, nx, ny, nz, nimages, image, nx_extra=0, ny_extra=0, halo_width=None):
        """grid_obj.f90:140-226; `image` is 1-based (this_image())."""
        halo = kDEFAULT_HALO_SIZE if halo_width is None else halo_width
        self.ximages, self.yimages = domain_decomposition(nx, ny, nimages)
        self.ximg = (image - 1) % self.ximages + 1
        self.yimg = (image - 1) // self.ximages + 1
        self.ny_global = ny + ny_extra; self.nx_global = nx + nx_extra; self.nz = nz
        self.nx = my_n(self.nx_global - nx_extra, self.ximg, self.ximages)
        self.ny = my_n(self.ny_global - ny_extra, self.yimg, self.yimages)
        self.ims = my_start(self.nx_global - nx_extra, self.ximg, self.ximages)
        self.ime = self.ims + self.nx + nx_extra - 1
        self.jms = my_start(self.ny_global - ny_extra, self.yimg, self.yimages)
        self.jme = self.jms + self.ny + ny_extra - 1
        self.kms, self.kme, self.kts, self.kte = 1, nz, 1, nz
        self.ids = self.jds = self.kds = 1
        self.ide, self.jde, self.kde = self.nx_global, self.ny_global, nz
        self.halo_nz = nz; self.halo_size = halo
        self._update_with_halos(halo)
        self.ns_halo_nx = self.nx_global // self.ximages + 1 + nx_extra
        self.ew_halo_ny = self.ny_global // self.yimages + 1 + ny_extra
        return self

    # boundary flags as exchangeable_obj.f90:24-27
    @property
    def north_boundary(self): return self.yimg == self.yimages
    @property
    def south_boundary(self): return self.yimg == 1
    @property
    def east_boundary(self): return self.ximg == self.ximages
    @property
    def west_boundary(self): return self.ximg == 1

    def _update_with_halos(self, h):    # grid_obj.f90:228-255
        self.ims -= 0 if self.west_boundary else h
        self.ime += 0 if self.east_boundary else h
        self.jms -= 0 if self.south_boundary else h
        self.jme += 0 if self.north_boundary else h
        self.its = self.ims + (1 if self.west_boundary else h)
        self.ite = self.ime - (1 if self.east_boundary else h)
        self.jts = self.jms + (1 if self.south_boundary else h)
        self.jte = self.jme - (1 if self.north_boundary else h)
        self.nx = self.ime - self.ims + 1
        self.ny = self.jme - self.jms + 1

    def neighbors(self, image):
        """exchangeable_obj.f90:69-115: images (1-based) to the N,S,E,W or None on a boundary."""
        return dict(
            north=None if self.north_boundary else image + self.ximages,
            south=None if self.south_boundary else image - self.ximages,
            east=None if self.east_boundary else image + 1,
            west=None if self.west_boundary else image - 1)
