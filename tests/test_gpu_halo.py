"""H1 on the GPU: the HIP pack/unpack kernels against the host-array double, and a two-tile
exchange emulated inside one process (two contexts on cuda:0, device buffers handed across)."""
import numpy as np
import pytest
import torch
from icar_amd.grid import grid_t
from host_tile import HostTile

pytestmark = pytest.mark.gpu


def mk_domain(g, fields):
    from icar_amd.domain import domain_t
    d = domain_t(g, device=0)
    for name, a in fields.items():
        d.set(name, a)
    return d


@pytest.mark.parametrize("halo", [1, 2])
def test_pack_unpack_match_host_double(halo):
    nimg = 4
    g = grid_t().set_grid_dimensions(70, 50, 7, nimg, 1, halo_width=halo)
    ny, nz, nx = g.jme - g.jms + 1, 7, g.ime - g.ims + 1
    rng = np.random.default_rng(5)
    host = {0: rng.standard_normal((ny, nz, nx)).astype(np.float32), 4: rng.standard_normal((ny, nz, nx)).astype(np.float32),
            7: rng.standard_normal((ny, nz, nx)).astype(np.float32)}
    d = mk_domain(g, {"water_vapor": host[0], "potential_temperature": host[4], "cloud_ice_number": host[7]})
    ht = HostTile(g, {k: v.copy() for k, v in host.items()})
    ids = [0, 4, 7]
    for direction in range(4):
        n = d.halo_count(direction, halo)
        assert n == ht.halo_count(direction, halo)
        gb = d.new_buffer(n * 3); hb = ht.new_buffer(n * 3)
        d.halo_pack(direction, halo, ids, gb); d.synchronize()
        ht.halo_pack(direction, halo, ids, hb)
        assert torch.equal(gb.cpu(), hb), f"pack dir {direction}"
        inbox = torch.from_numpy(rng.standard_normal(n * 3).astype(np.float32))
        d.halo_unpack(direction, halo, ids, inbox.cuda()); d.synchronize()
        ht.halo_unpack(direction, halo, ids, inbox)
    for fid, name in ((0, "water_vapor"), (4, "potential_temperature"), (7, "cloud_ice_number")):
        assert np.array_equal(d.get(name), ht.f[fid]), name
    d.close()


@pytest.mark.parametrize("halo", [1, 2])
@pytest.mark.parametrize("dirs", [[0, 1, 2, 3], [3, 0], [0, 1], [2, 3, 1]])
def test_pack_unpack_dirs_one_launch_equals_per_direction(halo, dirs):
    """icar_hip_halo_pack_dirs / _unpack_dirs == the per-direction calls in the reference's retrieve order N, S, E, W
    (exchangeable_obj.f90:138-151): E/W win the corner cells."""
    g = grid_t().set_grid_dimensions(70, 50, 7, 4, 1, halo_width=halo)
    ny, nz, nx = g.jme - g.jms + 1, 7, g.ime - g.ims + 1
    rng = np.random.default_rng(11)
    host = {0: rng.standard_normal((ny, nz, nx)).astype(np.float32), 5: rng.standard_normal((ny, nz, nx)).astype(np.float32)}
    d = mk_domain(g, {"water_vapor": host[0], "potential_temperature": host[5]})
    ids = [0, 4]
    ht = HostTile(g, {0: host[0].copy(), 4: host[5].copy()})
    gb = [d.new_buffer(d.halo_count(x, halo) * 2) for x in dirs]
    d.halo_pack_many(dirs, halo, ids, gb); d.synchronize()
    for x, b in zip(dirs, gb):
        hb = ht.new_buffer(ht.halo_count(x, halo) * 2)
        ht.halo_pack(x, halo, ids, hb)
        assert torch.equal(b.cpu(), hb), f"pack dir {x}"
    inbox = [torch.from_numpy(rng.standard_normal(b.numel()).astype(np.float32)) for b in gb]
    d.halo_unpack_many(dirs, halo, ids, [t.cuda() for t in inbox]); d.synchronize()
    for x, t in sorted(zip(dirs, inbox), key=lambda p: p[0]):
        ht.halo_unpack(x, halo, ids, t)
    assert np.array_equal(d.get("water_vapor"), ht.f[0])
    assert np.array_equal(d.get("potential_temperature"), ht.f[4])
    d.close()


def test_two_tiles_one_process_exchange():
    """West tile (image 1) and east tile (image 2) of a 1x2... 2x1 decomposition: my east faces land
    in the neighbour's west halo and vice versa; both end up equal to the global field."""
    nxg, nyg, nz = 64, 24, 5
    j, k, i = np.meshgrid(np.arange(nyg), np.arange(nz), np.arange(nxg), indexing="ij")
    G = (i + 100 * j + 10000 * k).astype(np.float32)
    tiles = []
    for img in (1, 2):
        g = grid_t().set_grid_dimensions(nxg, nyg, nz, 2, img)
        assert (g.ximages, g.yimages) == (2, 1)
        a = G[g.jms - 1:g.jme, :, g.ims - 1:g.ime].copy()
        if img == 1: a[:, :, -1] = -1
        else: a[:, :, 0] = -1
        tiles.append((g, mk_domain(g, {"water_vapor": a})))
    (g1, d1), (g2, d2) = tiles
    b12 = d1.new_buffer(d1.halo_count(2, 1)); b21 = d2.new_buffer(d2.halo_count(3, 1))
    d1.halo_pack(2, 1, [0], b12); d2.halo_pack(3, 1, [0], b21)       # put_east / put_west
    d1.synchronize(); d2.synchronize()
    d2.halo_unpack(3, 1, [0], b12); d1.halo_unpack(2, 1, [0], b21)    # retrieve_west_halo / retrieve_east_halo
    for g, d in tiles:
        assert np.array_equal(d.get("water_vapor"), G[g.jms - 1:g.jme, :, g.ims - 1:g.ime])
        d.close()


@pytest.mark.parametrize("halo", [1, 2])
def test_staggered_boxes_match_host_double(halo):
    """exchange_u / exchange_v faces (exchangeable_obj.f90:158-229): every send and receive box of
    icar_amd.halo.staggered_boxes through icar_hip_box_pack/unpack vs plain numpy slicing, on data_3d and dqdt_3d."""
    from icar_amd.halo import staggered_boxes
    g = grid_t().set_grid_dimensions(70, 50, 6, 4, 1, halo_width=halo)
    ny, nz, nx = g.jme - g.jms + 1, 6, g.ime - g.ims + 1
    rng = np.random.default_rng(9)
    mk = lambda s: rng.standard_normal(s).astype(np.float32)
    u, v, du, dv = mk((ny, nz, nx + 1)), mk((ny + 1, nz, nx)), mk((ny, nz, nx + 1)), mk((ny + 1, nz, nx))
    d = mk_domain(g, {"u": u, "v": v}); d.set_dqdt("u", du); d.set_dqdt("v", dv)
    ht = HostTile(g, {11: u.copy(), 12: v.copy()}, {11: du.copy(), 12: dv.copy()})
    fid = {"u": 11, "v": 12}
    for direction, (send, recv) in staggered_boxes(nx, ny, halo).items():
        for which in (0, 1):
            for kind, i0, ni, j0, nj in send:
                gb = d.new_buffer(ni * nj * nz); hb = ht.new_buffer(ni * nj * nz)
                d.box_pack(fid[kind], which, i0, ni, j0, nj, gb); d.synchronize()
                ht.box_pack(fid[kind], which, i0, ni, j0, nj, hb)
                assert torch.equal(gb.cpu(), hb), (direction, kind, which)
            for kind, i0, ni, j0, nj in recv:
                inbox = torch.from_numpy(mk(ni * nj * nz))
                d.box_unpack(fid[kind], which, i0, ni, j0, nj, inbox.cuda()); d.synchronize()
                ht.box_unpack(fid[kind], which, i0, ni, j0, nj, inbox)
    assert np.array_equal(d.get("u"), ht.f[11]) and np.array_equal(d.get("v"), ht.f[12])
    assert np.array_equal(d.get_dqdt("u"), ht.dq[11]) and np.array_equal(d.get_dqdt("v"), ht.dq[12])
    assert not np.array_equal(ht.f[11], u) and not np.array_equal(ht.dq[12], dv)
    with pytest.raises(Exception):
        d.box_pack(11, 0, 0, nx + 2, 0, 1, d.new_buffer((nx + 2) * nz))      # outside the field
    d.close()
