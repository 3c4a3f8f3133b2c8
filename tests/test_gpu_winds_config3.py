"""BASELINE configs[3] at its literal size (about a minute and ~35 GB of HBM: runs whenever the GPU has 40 GB free; ICAR_CONFIG3_LUT=0 skips it): the linear-wind LUT of ONE
GPU's tile -- 512 x 256 x 40 of a 1024 x 1024 domain split 2 x 4, terrain padded to 1128 x 1128 for the FFTs, the reference's
default axes 24 dir x 6 spd x 5 N^2 = 720 entries (linear_winds.f90:596-830, :1180-1309).  Records the build time and the memory it
takes (gpurun_out/parity/winds_config3.jsonl -> profiles/r04_winds.json) and compares three entries, chosen at random, with
the numpy oracle (oracle/wind_oracle.py) evaluated at that size: every level of both components within 1e-5 of the entry's maximum."""
import os
import time
import numpy as np
import pytest
import torch
from icar_amd import linear_winds as LW
from icar_amd.domain import domain_t
from icar_amd.grid import grid_t
from icar_amd.options import options_t
from oracle import wind_oracle as W
from wind_case import terrain, lut_options
from util import parity_record

pytestmark = pytest.mark.gpu


def test_lut_of_one_tile_of_the_1024_domain():
    if os.environ.get("ICAR_CONFIG3_LUT") == "0":
        pytest.skip("ICAR_CONFIG3_LUT=0")
    if torch.cuda.mem_get_info()[0] < 40 * 2 ** 30:
        pytest.skip("needs 40 GB of free HBM (the LUT of configs[3]'s tile is 31.6 GiB)")
    nxg = nyg = 1024; nz = 40; dx = 2000.0; nimages, image = 8, 3
    opt = options_t()
    dz = np.array([50., 75., 125., 200., 300., 400.] + [500.] * 34, np.float32)[:nz]
    opt.parameters.dz_levels = dz
    g = grid_t().set_grid_dimensions(nxg, nyg, nz, nimages, image)
    d = domain_t(g, device=0, dx=dx)
    assert (g.ximages, g.yimages) == (2, 4) and d.nx in (513, 514) and d.ny in (257, 258)
    t = terrain(nxg, nyg, seed=7)
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time(); LW.setup_linwinds(d, opt, t, build=False); d.synchronize(); t_setup = time.time() - t0
    zc = np.cumsum(dz, dtype=np.float32) - dz / np.float32(2)
    zb, zt = LW.layer_bounds(zc, 0.0, dz)
    t0 = time.time(); LW.build_lut(d, zb, zt); d.synchronize(); t_lut = time.time() - t0
    used = free0 - torch.cuda.mem_get_info()[0]
    lt_o = opt.lt_options
    ncombo = lt_o.n_dir_values * lt_o.n_spd_values * lt_o.n_nsq_values
    # the oracle on the same global terrain
    tf, lt, buf = W.setup_linwinds(t.T.copy(), dx, lt_o.buffer)
    lo = lut_options(lt_o)
    dirv = W.linear_space(lo["dirmin"], lo["dirmax"], lo["n_dir_values"]); spdv = W.linear_space(lo["spdmin"], lo["spdmax"], lo["n_spd_values"])
    nsqv = W.linear_space(lo["nsqmin"], lo["nsqmax"], lo["n_nsq_values"])
    rng = np.random.default_rng(2026)
    i0, j0 = g.ims - 1, g.jms - 1
    worst = 0.0; picked = []
    for _ in range(3):
        k = int(rng.integers(1, lt_o.n_spd_values)); i = int(rng.integers(0, lt_o.n_dir_values)); j = int(rng.integers(0, lt_o.n_nsq_values))
        picked.append((k, i, j))
        u = W.calc_u(dirv[i], spdv[k]); v = W.calc_v(dirv[i], spdv[k]); nsq = np.float32(W._libm.expf(float(nsqv[j])))
        got_u = LW.lut_entry(d, 0, k, i, j); got_v = LW.lut_entry(d, 1, k, i, j)           # [ny(+1), nz, nx(+1)]
        for z in range(nz):
            up, vp = W.linear_perturbation_constz(u, v, nsq, zb[z], zt[z], lo["minimum_layer_size"], tf, lt)
            tu, tv = W.destagger(up, vp, buf, lt.nx, lt.ny)                                 # [nxg+1, nyg], [nxg, nyg+1]
            want_u = tu[i0:i0 + d.nx + 1, j0:j0 + d.ny].T; want_v = tv[i0:i0 + d.nx, j0:j0 + d.ny + 1].T
            su = max(float(abs(want_u).max()), 1e-30); sv = max(float(abs(want_v).max()), 1e-30)
            eu = float(abs(got_u[:, z, :] - want_u).max()) / su; ev = float(abs(got_v[:, z, :] - want_v).max()) / sv
            worst = max(worst, eu, ev)
            assert eu <= 1e-5 and ev <= 1e-5, (k, i, j, z, eu, ev)
        assert float(abs(got_u).max()) > 1e-3
    parity_record("winds_config3", "lut 512x256x40 tile of 1024x1024, fft 1128x1128, 720 entries",
                  {"lut": {"setup_s": t_setup, "lut_build_s": t_lut, "device_bytes": int(used), "entries": ncombo, "tile_memory": [d.nx, nz, d.ny],
                           "fft": [nxg + 2 * (lt_o.buffer + 2), nyg + 2 * (lt_o.buffer + 2)], "entries_checked": picked, "max_err_over_entry_max": worst}})
    print(f"config3 LUT: setup {t_setup:.2f} s, build {t_lut:.2f} s, {used / 2**30:.1f} GiB, worst error {worst:.2e} of the entry maximum")
    d.close()
