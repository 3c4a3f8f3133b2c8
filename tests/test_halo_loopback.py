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


class _FakeDomain:
    """records the order of the calls mp_and_halo makes (time_step.f90:512-526 + the second stream)"""
    def __init__(self):
        self.log = []; self.model_time_seconds = 0.0; self.mp_state = dict(last_model_time=-999.0)
    def __getattr__(self, name):
        if name in ("aux_fork", "aux_begin", "aux_end", "aux_join", "halo_send", "halo_retrieve"):
            return lambda: self.log.append(name)
        raise AttributeError(name)


def test_mp_and_halo_orders_strips_exchange_interior(monkeypatch):
    from icar_amd import time_step
    from icar_amd.options import options_t
    from icar_amd.constants import kMP_THOMPSON, kMP_WSM3
    calls = []
    from icar_amd import advection
    monkeypatch.setattr(time_step, "mp", lambda d, o, dt, halo=None, subset=None: d.log.append("mp_halo" if halo else "mp_subset"))
    monkeypatch.setattr(advection, "setup_winds", lambda d, o, dt: d.log.append("setup_winds"))
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    d = _FakeDomain()
    time_step.mp_and_halo(d, opt, 10.0)
    # the wind setup of the following advect() goes out on the main stream while the interior runs on the second one
    assert d.log == ["aux_fork", "mp_halo", "halo_send", "aux_begin", "mp_subset", "aux_end", "setup_winds", "aux_join", "halo_retrieve"]
    d = _FakeDomain()
    time_step.mp_and_halo(d, opt, 10.0, prepare_advection=False)
    assert d.log == ["aux_fork", "mp_halo", "halo_send", "aux_begin", "mp_subset", "aux_end", "aux_join", "halo_retrieve"]
    opt.physics.microphysics = 0                            # no microphysics: nothing to put on a second stream
    d = _FakeDomain()
    time_step.mp_and_halo(d, opt, 10.0)
    assert d.log == ["mp_halo", "halo_send", "mp_subset", "halo_retrieve"]
