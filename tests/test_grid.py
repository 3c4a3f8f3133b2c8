"""G1: tile index algebra (src/objects/grid_obj.f90).  Known answers + the self-consistency
scenario of src/tests/test_caf_other_image_grids.f90 (nx=1024, ny=1234, nz=13)."""
import numpy as np
import pytest
from icar_amd.grid import grid_t, domain_decomposition, my_n, my_start


def test_decomposition_nearest_square():
    assert domain_decomposition(512, 512, 4) == (2, 2)
    assert domain_decomposition(1024, 1024, 8) == (2, 4)        # 8 GPUs -> 2x4 for square domains
    assert domain_decomposition(512, 512, 1) == (1, 1)
    assert domain_decomposition(2000, 500, 4) == (4, 1)
    assert domain_decomposition(300, 20, 6) == (6, 1)
    xs, ys = domain_decomposition(1024, 1234, 36)
    assert xs * ys == 36


def test_block_distribution_with_remainder():
    for n, nimg in [(1234, 7), (1024, 8), (10, 3), (13, 13)]:
        sizes = [my_n(n, me, nimg) for me in range(1, nimg + 1)]
        starts = [my_start(n, me, nimg) for me in range(1, nimg + 1)]
        assert sum(sizes) == n and starts[0] == 1
        for a, b, s in zip(starts[:-1], starts[1:], sizes[:-1]):
            assert b == a + s
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@pytest.mark.parametrize("nimg", [1, 2, 4, 6, 8, 36])
def test_tiles_cover_domain_and_halos_overlap_by_one(nimg):
    nx, ny, nz = 1024, 1234, 13
    own = np.zeros((ny + 2, nx + 2), int)
    for img in range(1, nimg + 1):
        g = grid_t().set_grid_dimensions(nx, ny, nz, nimg, img)
        assert (g.kms, g.kme, g.ids, g.ide, g.jds, g.jde) == (1, nz, 1, nx, 1, ny)
        # memory bounds = owned block + halo on interior edges
        ox0 = g.ims + (0 if g.west_boundary else g.halo_size); ox1 = g.ime - (0 if g.east_boundary else g.halo_size)
        oy0 = g.jms + (0 if g.south_boundary else g.halo_size); oy1 = g.jme - (0 if g.north_boundary else g.halo_size)
        own[oy0:oy1 + 1, ox0:ox1 + 1] += 1
        assert g.its == (g.ims + 1) and g.ite == g.ime - 1 and g.jts == g.jms + 1 and g.jte == g.jme - 1
        assert g.nx == g.ime - g.ims + 1 and g.ny == g.jme - g.jms + 1
        nb = g.neighbors(img)
        if nb["east"]:
            e = grid_t().set_grid_dimensions(nx, ny, nz, nimg, nb["east"])
            assert e.ims == g.ime - 2 * g.halo_size + 1 and e.jms == g.jms and e.jme == g.jme
        if nb["north"]:
            n = grid_t().set_grid_dimensions(nx, ny, nz, nimg, nb["north"])
            assert n.jms == g.jme - 2 * g.halo_size + 1 and n.ims == g.ims and n.ime == g.ime
    assert (own[1:ny + 1, 1:nx + 1] == 1).all() and own.sum() == nx * ny


def test_single_image_bounds():
    g = grid_t().set_grid_dimensions(100, 100, 30, 1, 1)
    assert (g.ims, g.ime, g.its, g.ite, g.jts, g.jte) == (1, 100, 2, 99, 2, 99)


def test_staggered_grids():
    gu = grid_t().set_grid_dimensions(512, 512, 40, 4, 1, nx_extra=1)
    g = grid_t().set_grid_dimensions(512, 512, 40, 4, 1)
    assert gu.ime == g.ime + 1 and gu.nx_global == 513


def test_grid_matches_reference_golden():
    """tests/golden/grid_tiles.npz: integer members of the reference's own grid_t (compiled grid_obj.f90) for every
    image of 11 decompositions x 5 domains x {mass, u, v} grids -- bit-exact (integers)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_tiles.npz"))
    rows, members = z["rows"], [str(m) for m in z["members"]]
    assert len(rows) > 2000
    for r in rows:
        nx, ny, nz, n, img, ex, ey = (int(v) for v in r[:7])
        g = grid_t().set_grid_dimensions(nx, ny, nz, n, img, nx_extra=ex, ny_extra=ey)
        for m, want in zip(members, r[7:]):
            if m in ("is2d", "is3d"):
                continue
            assert getattr(g, m) == int(want), (nx, ny, nz, n, img, ex, ey, m, getattr(g, m), int(want))
