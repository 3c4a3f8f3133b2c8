"""HaloComm(loopback=True): edges without a neighbouring image wrap around to the tile's own opposite edge -- the
periodic ring of the reference's src/tests/test_mpdata.f90, which bench.py uses at N=1 so that the 1-GPU line runs the
same pack / unpack launches as every rank of an N-GPU run.  Host-array tile double, no GPU, no process group."""
import numpy as np
from icar_amd.grid import grid_t
from icar_amd.halo import HaloComm, DIR_NORTH, DIR_SOUTH, DIR_EAST, DIR_WEST
from host_tile import HostTile


def test_single_image_loopback_is_a_periodic_wrap():
    nx, ny, nz = 12, 9, 3
    g = grid_t().set_grid_dimensions(nx, ny, nz, 1, 1)
    rng = np.random.default_rng(0)
    a = rng.random((ny, nz, nx)).astype(np.float32); b = rng.random((ny, nz, nx)).astype(np.float32)
    a0, b0 = a.copy(), b.copy()
    tile = HostTile(g, {0: a, 4: b})
    comm = HaloComm(g, 1, loopback=True)
    assert not comm.peers and sorted(comm.loop) == [DIR_NORTH, DIR_SOUTH, DIR_EAST, DIR_WEST]
    comm.send(tile, [0, 4]); comm.retrieve(tile, [0, 4])
    for x, x0 in ((a, a0), (b, b0)):
        assert np.array_equal(x[1:-1, :, 1:-1], x0[1:-1, :, 1:-1])          # owned cells untouched
        assert np.array_equal(x[0, :, 1:-1], x0[ny - 2, :, 1:-1])            # south ring row <- my northern edge row
        assert np.array_equal(x[ny - 1, :, 1:-1], x0[1, :, 1:-1])            # north ring row <- my southern edge row
        assert np.array_equal(x[1:-1, :, 0], x0[1:-1, :, nx - 2])            # west ring column <- my eastern edge column
        assert np.array_equal(x[1:-1, :, nx - 1], x0[1:-1, :, 1])


def test_without_loopback_a_single_image_exchanges_nothing():
    g = grid_t().set_grid_dimensions(8, 8, 2, 1, 1)
    a = np.arange(8 * 2 * 8, dtype=np.float32).reshape(8, 2, 8); a0 = a.copy()
    comm = HaloComm(g, 1)
    comm.send(HostTile(g, {0: a}), [0]); comm.retrieve(HostTile(g, {0: a}), [0])
    assert np.array_equal(a, a0) and not comm.loop
