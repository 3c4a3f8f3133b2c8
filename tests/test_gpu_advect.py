"""HIP advection (rows A1-A5) vs the CPU oracle (bit-exact restatement of the compiled reference) on identical inputs,
through the C ABI.
  * upwind scheme and mpdata_order 1 (the donor-cell kernel, HBM-bound): BIT-EXACT (same operation order,
    -ffp-contract=off, IEEE division);
  * MPDATA's corrective iterations (the fused kernel, VALU-bound: 1-ulp reciprocals, fma): EVERY cell within 1e-5 of the
    local field scale (util.local_rel_err; north-star tolerance).  Measured: 2e-7 ... 3e-6 (profiles/r02_parity.json)."""
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.options import options_t
from icar_amd.advection import advect
from icar_amd.capi import lib, check
from icar_amd.constants import kADV_UPWIND, kADV_MPDATA
from util import SCALARS, MEMBER, KVAR, bits_equal, nbitdiff, single_image_domain, adv_args, assert_fields_close, local_rel_err, parity_record, roughen_winds, equals_reference_vector

pytestmark = pytest.mark.gpu


def run_case(oracle, scheme, nx, ny, nz, names, hill=1000.0, dens=False, order=2, fct=True, nsteps=2, noise=0.01, exact_mode=False, rough=0.0):
    c = ideal.make_case(nx, ny, nz, hill_height=hill, noise=noise, n_hydro=1)
    if rough: c = roughen_winds(c, oracle, rough)
    dt = ideal.cfl_dt(c)
    q = np.stack([c[n] for n in names]).copy()
    oracle.advect(scheme, q, *adv_args(c), dt, advect_density=dens, mpdata_order=order, fct=fct, nsteps=nsteps)
    d = single_image_domain(c)
    if exact_mode:
        check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
    opt = options_t()
    opt.physics.advection = scheme
    opt.parameters.advect_density = dens
    opt.adv_options.mpdata_order = order
    opt.adv_options.flux_corrected_transport = fct
    opt.advect_vars([KVAR[n] for n in names])
    for _ in range(nsteps):
        advect(d, opt, dt)
    out = {n: d.get(MEMBER[n]) for n in names}
    d.close()
    exact = (scheme == kADV_UPWIND) or order == 1 or exact_mode
    worst = 0.0; stats = {}
    for m, n in enumerate(names):
        assert np.abs(out[n] - c[n]).max() > 0, f"{n}: advection did nothing"
        if exact:
            assert bits_equal(out[n], q[m]), f"{n}: {nbitdiff(out[n], q[m])} cells differ, max|d|={np.abs(out[n]-q[m]).max()}"
            stats[n] = {"bitdiff_cells": 0, "cells": int(q[m].size)}
        else:
            err = assert_fields_close(out[n], q[m], n); worst = max(worst, err)
            stats[n] = {"max_local_rel": err, "bitdiff_cells": nbitdiff(out[n], q[m]), "cells": int(q[m].size),
                        "max_abs_over_max": float(np.abs(out[n].astype(np.float64) - q[m]).max() / max(float(np.abs(q[m]).max()), 1e-300))}
    parity_record("advect", f"{'upwind' if scheme == kADV_UPWIND else ('mpdata(exact)' if exact_mode else 'mpdata')} {nx}x{ny}x{nz} order{order} fct{int(fct)} dens{int(dens)} steps{nsteps}" + (f" rough{rough:g}" if rough else ""), stats)
    return worst


@pytest.mark.parametrize("dens", [False, True])
def test_upwind_bit_exact(oracle, dens):
    run_case(oracle, kADV_UPWIND, 70, 37, 12, ["water_vapor", "potential_temperature", "ice_number"], dens=dens)


@pytest.mark.parametrize("dens,fct,order", [(False, True, 2), (True, True, 2), (False, False, 2), (False, True, 1),
                                            (False, True, 3), (True, True, 3), (False, False, 3), (False, True, 4)])
def test_mpdata_vs_oracle(oracle, dens, fct, order):
    """adv_mpdata.f90:372-402 for mpdata_order 1 .. 4 (the oracle's iord loop is pinned to the compiled reference,
    tests/test_oracle_vs_ref.py::test_mpdata_order_matches_reference)."""
    # (the tolerance is a per-step bound: one step for the higher orders, whose extra iterations compound the rounding)
    run_case(oracle, kADV_MPDATA, 70, 37, 12, ["water_vapor", "cloud_water", "potential_temperature"],
             dens=dens, fct=fct, order=order, nsteps=2 if order <= 2 else 1)


@pytest.mark.parametrize("dens,fct,order", [(False, True, 2), (True, True, 2), (False, False, 2), (False, True, 3), (True, True, 3),
                                            (False, False, 3), (False, True, 4)])
def test_mpdata_exact_mode_bit_exact(oracle, dens, fct, order):
    """icar_hip_mpdata_exact(ctx, 1): the corrective iterations in the reference's operation order (mpdata_exact.hip) --
    EVERY cell of every scalar bit-identical to the CPU oracle (itself bit-identical to adv_mpdata.f90 compiled unmodified) over
    several steps, for mpdata_order 2 .. 4, with and without the limiter and advect_density."""
    run_case(oracle, kADV_MPDATA, 70, 37, 12, ["water_vapor", "cloud_water", "potential_temperature"],
             dens=dens, fct=fct, order=order, nsteps=3, exact_mode=True)


def test_mpdata_exact_mode_sizes(oracle):
    """ragged sizes, the smallest lines the limiter is defined for (3 cells), all 9 Thompson scalars, config[1]'s grid"""
    run_case(oracle, kADV_MPDATA, 5, 4, 3, ["water_vapor"], hill=0.0, nsteps=2, exact_mode=True)
    run_case(oracle, kADV_MPDATA, 129, 3, 5, ["water_vapor", "cloud_water"], nsteps=2, exact_mode=True)
    run_case(oracle, kADV_MPDATA, 66, 34, 10, SCALARS, nsteps=2, exact_mode=True)
    run_case(oracle, kADV_MPDATA, 256, 256, 40, ["water_vapor", "cloud_water", "rain"], nsteps=2, exact_mode=True)


def test_mpdata_exact_mode_full_size_every_cell(oracle):
    """512x512x40 (BASELINE metric size), the 9 Thompson scalars, two steps: every cell bit-identical to the CPU oracle; and the
    fused kernel (the default) within its 1e-5 of THIS result, i.e. the two device paths agree with each other as they do with
    the reference"""
    run_case(oracle, kADV_MPDATA, 512, 512, 40, SCALARS, nsteps=2, exact_mode=True)


@pytest.mark.parametrize("nx,ny,nz", [(61, 70, 41), (64, 40, 80), (200, 130, 40), (100, 100, 30), (70, 46, 7)])
def test_mpdata_level_and_chunk_layouts(oracle, nx, ny, nz):
    """Level counts that exercise every levels-per-thread variant of the fused kernel (1..5, 8 and 16 waves, a last
    wave with idle levels), several y chunks, XCD shares that do not divide evenly."""
    run_case(oracle, kADV_MPDATA, nx, ny, nz, ["water_vapor", "cloud_water", "rain", "potential_temperature"], nsteps=1)


@pytest.mark.parametrize("nx,ny,nz,dens,rough", [(70, 37, 12, False, 0.0), (70, 37, 12, True, 0.5), (130, 64, 40, False, 0.5), (61, 70, 41, True, 0.5),
                                                 (512, 256, 40, False, 0.5)])
def test_fused_kernel_donor_cell_pass_bit_identical(oracle, probe, nx, ny, nz, dens, rough):
    """The field after the donor-cell pass INSIDE the fused kernel (q2, never written to memory) is bit-identical to the reference's
    (adv_mpdata.f90:44-105): the limiter's factors next to the ring are all-or-nothing in whether q2 equals a local extremum
    (adv_mpdata_FCT_core.f90:80-113 with fin = fout = 0 there), so one ulp of q2 is a whole corrective flux at that cell (round 5:
    9.2e-6 of theta at one cell of the config-3 tile).  With the antidiffusive coefficients of the context zeroed
    (tests/support/mpdata_probe.hip) the kernel's output is q2; the oracle's is mpdata_order = 1."""
    from icar_amd.advection import setup_winds
    names = ["water_vapor", "cloud_water", "potential_temperature"]
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    if rough: c = roughen_winds(c, oracle, rough)
    dt = ideal.cfl_dt(c)
    q = np.stack([c[n] for n in names]).copy()
    oracle.advect(kADV_MPDATA, q, *adv_args(c), dt, advect_density=dens, mpdata_order=1, fct=True, nsteps=1)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.parameters.advect_density = dens
    opt.adv_options.mpdata_order = 2; opt.adv_options.flux_corrected_transport = True
    opt.advect_vars([KVAR[n] for n in names])
    d.configure(opt)
    setup_winds(d, opt, dt)
    assert probe.icar_probe_mpdata_zero_antidiffusion(d.ctx) == 0
    advect(d, opt, dt)                                  # (same scheme, dt and density: the coefficients are not rebuilt)
    for m, n in enumerate(names):
        got = d.get(MEMBER[n])
        assert np.abs(got - c[n]).max() > 0
        assert bits_equal(got, q[m]), f"{n}: q2 of {nbitdiff(got, q[m])} of {got.size} cells differs from the reference's donor-cell pass"
    d.close()


def test_mpdata_rough_winds_metric_size(oracle):
    """BASELINE metric size, the 9 Thompson scalars, winds with white noise of 0.5 m/s on u and v (w rebalanced): every cell within
    the tolerance -- and, by assert_fields_close's margin rule, within 0.3 of it.  """
    run_case(oracle, kADV_MPDATA, 512, 512, 40, SCALARS, nsteps=1, rough=0.5)


@pytest.mark.parametrize("dens,fct,order", [(False, True, 2), (True, True, 2), (False, True, 3)])
def test_mpdata_rough_winds(oracle, dens, fct, order):
    run_case(oracle, kADV_MPDATA, 130, 67, 40, ["water_vapor", "cloud_water", "potential_temperature"], dens=dens, fct=fct, order=order, nsteps=1, rough=0.5)


@pytest.mark.parametrize("name", ["adv_mpdata_rough_40x36x12", "adv_mpdata_rough_dens_order3_40x36x12", "adv_mpdata_dens_40x36x12"])
def test_device_against_the_compiled_references_vectors(name):
    """No oracle in between: the inputs and outputs of tests/golden/<name>.npz were written by the reference's own adv_mpdata.f90
    (compiled unmodified, tests/golden/make_golden.py).  The exact mode reproduces the stored field bit for bit, the fused kernel
    every cell to the tolerance -- incl. the two rough-wind fixtures (white noise on u, v: the limiter's all-or-nothing factor next
    to the ring decides cells there)."""
    import json, os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    p = json.loads(str(z["params"]))
    base = ideal.make_case(p["nx"], p["ny"], p["nz"], hill_height=p["hill"], noise=0.01, n_hydro=1)
    c = dict(base)
    for n in ["u", "v", "w", "density", "jacobian", "jacobian_u", "jacobian_v", "jacobian_w", "advection_dz", "dz_levels"] + p["vars"]:
        c[n] = np.ascontiguousarray(z["in_" + n])
    for exact_mode in (True, False):
        d = single_image_domain(c)
        if exact_mode: check(lib().icar_hip_mpdata_exact(d.ctx, 1), "mpdata_exact")
        opt = options_t(); opt.physics.advection = p["scheme"]; opt.parameters.advect_density = bool(p["dens"])
        opt.adv_options.mpdata_order = p["order"]; opt.adv_options.flux_corrected_transport = bool(p["fct"])
        opt.advect_vars([KVAR[n] for n in p["vars"]])
        for _ in range(p["nsteps"]):
            advect(d, opt, float(z["dt"]))
        for m, n in enumerate(p["vars"]):
            got = d.get(MEMBER[n])
            if exact_mode: assert equals_reference_vector(got, z["q"][m]), f"{n}: {nbitdiff(got, z['q'][m])} cells differ from the compiled reference's output"
            else: assert_fields_close(got, z["q"][m], n, record=("advect", f"fused kernel vs the compiled reference's vectors: {name}"))
        d.close()


def test_mpdata_all_thompson_scalars(oracle):
    """The 9 scalars Thompson advects (mp_driver.f90:128-131) in one batched launch."""
    run_case(oracle, kADV_MPDATA, 66, 34, 10, SCALARS, nsteps=1)


def test_mpdata_ragged_sizes(oracle):
    """Sizes that are not multiples of the 64x4 block and minimal tiles."""
    run_case(oracle, kADV_MPDATA, 5, 4, 3, ["water_vapor"], hill=0.0, nsteps=1)
    run_case(oracle, kADV_MPDATA, 129, 3, 5, ["water_vapor", "cloud_water"], nsteps=1)
    run_case(oracle, kADV_UPWIND, 3, 3, 2, ["water_vapor"], hill=0.0, nsteps=1)


def test_config1_upwind_100x100x30(oracle):
    """BASELINE config[0]: 100x100x30 ideal hill, upwind, the 5 scalars mp_simple advects."""
    run_case(oracle, kADV_UPWIND, 100, 100, 30, ["potential_temperature", "water_vapor", "cloud_water", "rain", "snow"],
             nsteps=2)


def test_boundary_ring_untouched(oracle):
    c = ideal.make_case(40, 20, 8, hill_height=500.0, noise=0.02)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.advect_vars(["water_vapor"])
    advect(d, opt, ideal.cfl_dt(c))
    q = d.get("water_vapor"); d.close()
    q0 = c["water_vapor"]
    assert bits_equal(q[0], q0[0]) and bits_equal(q[-1], q0[-1])
    assert bits_equal(q[:, :, 0], q0[:, :, 0]) and bits_equal(q[:, :, -1], q0[:, :, -1])


def test_full_size_properties():
    """512x512x40 (BASELINE metric size): conservation + monotonicity properties of MPDATA+FCT
    (flat terrain, uniform wind => interior mass is conserved up to boundary fluxes; FCT keeps the
    field within the initial global bounds)."""
    c = ideal.make_case(512, 512, 40, hill_height=0.0, noise=0.0)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.advect_vars(["water_vapor", "potential_temperature"])
    dt = ideal.cfl_dt(c)
    for _ in range(3):
        advect(d, opt, dt)
    q = d.get("water_vapor"); th = d.get("potential_temperature"); d.close()
    q0 = c["water_vapor"]; th0 = c["potential_temperature"]
    assert np.isfinite(q).all() and np.isfinite(th).all()
    assert q.min() >= q0.min() * (1 - 1e-6) and q.max() <= q0.max() * (1 + 1e-6)
    assert th.min() >= th0.min() * (1 - 1e-6) and th.max() <= th0.max() * (1 + 1e-6)
    # horizontally uniform theta is a fixed point of the scheme when w=0
    assert np.abs(th - th0).max() <= 2e-4 * th0.max()
    # a blob far from the boundary moves but keeps its mass (sum over a window that contains it)
    s0 = q0[64:-64, :, 64:-64].astype(np.float64).sum(); s1 = q[64:-64, :, 64:-64].astype(np.float64).sum()
    assert abs(s1 - s0) / s0 < 5e-3


def test_full_size_every_cell_vs_oracle(oracle):
    """512x512x40 (BASELINE metric size), hill case, the 9 Thompson scalars, one MPDATA step: EVERY cell of every scalar
    within 1e-5 of the local field scale of the CPU oracle's result (OpenMP restatement, bit-identical to the compiled
    reference); the boundary ring bit for bit.  Also records the measured deviation (printed; profiles/r02_parity.json is
    written by profiles/collect_parity.py from the same comparison)."""
    nx = ny = 512; nz = 40
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    dt = ideal.cfl_dt(c)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.advect_vars([KVAR[n] for n in SCALARS])
    advect(d, opt, dt)
    out = {n: d.get(MEMBER[n]) for n in SCALARS}
    d.close()
    q = np.stack([c[n] for n in SCALARS]).copy()
    oracle.advect(kADV_MPDATA, q, *adv_args(c), dt)
    stats = {}
    for k, n in enumerate(SCALARS):
        err = assert_fields_close(out[n], q[k], n)
        stats[n] = {"max_local_rel": err, "bitdiff_cells": nbitdiff(out[n], q[k]), "cells": int(q[k].size),
                    "max_abs_over_max": float(np.abs(out[n].astype(np.float64) - q[k]).max() / max(float(np.abs(q[k]).max()), 1e-300))}
        print(f"{n}: max |d| / local scale = {err:.2e}, cells differing in any bit: {nbitdiff(out[n], q[k])} of {q[k].size}")
    parity_record("advect", "mpdata 512x512x40 order2 fct1 dens0 steps1 (every cell, 9 scalars)", stats)
    for k, n in enumerate(SCALARS):
        assert bits_equal(out[n][0], q[k][0]) and bits_equal(out[n][-1], q[k][-1])
        assert bits_equal(out[n][:, :, 0], q[k][:, :, 0]) and bits_equal(out[n][:, :, -1], q[k][:, :, -1])


@pytest.mark.parametrize("fct", [True, False])
def test_mpdata_sparse_fields(oracle, fct):
    """Hydrometeor-like fields: zero almost everywhere with small blobs that straddle the 58-cell x tiles, the level groups
    of a wave and the y chunks of the fused kernel, plus one all-zero field.  Every cell within 1e-5 of the local scale;
    a cell whose neighbourhood is all zero must come out EXACTLY zero (local_rel_err treats it that way): the reciprocal
    arithmetic must not leak anything into empty air."""
    nx, ny, nz = 200, 45, 20
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.01, n_hydro=1)
    rng = np.random.default_rng(7)
    names = ["water_vapor", "cloud_water", "rain", "snow", "cloud_ice"]
    blobs = {"cloud_water": [(57, 5, 7), (116, 16, 4), (190, 30, 0)], "rain": [(1, 1, 1), (198, 43, 18), (64, 24, 9)],
             "snow": [(100, 20, 10)], "cloud_ice": []}
    for n, bl in blobs.items():
        a = np.zeros((ny, nz, nx), np.float32)
        for (i, j, k) in bl:
            sl = (slice(max(j - 1, 0), j + 2), slice(max(k - 1, 0), k + 2), slice(max(i - 2, 0), i + 3))
            a[sl] = np.float32(1e-4) * (1 + rng.random(a[sl].shape)).astype(np.float32)
        c[n] = a
    dt = ideal.cfl_dt(c)
    q = np.stack([c[n] for n in names]).copy()
    d = single_image_domain(c)
    opt = options_t(); opt.physics.advection = kADV_MPDATA; opt.adv_options.flux_corrected_transport = fct
    opt.advect_vars([KVAR[n] for n in names])
    for step in range(3):
        oracle.advect(kADV_MPDATA, q, *adv_args(c), dt, fct=fct, nsteps=1)
        advect(d, opt, dt)
        for m, n in enumerate(names):
            got = d.get(MEMBER[n])
            assert_fields_close(got, q[m], f"step {step} {n}")
            assert np.array_equal(got == 0, q[m] == 0) or np.abs(got[(got == 0) != (q[m] == 0)]).max() < 1e-30, n
        q = np.stack([d.get(MEMBER[n]) for n in names])          # keep the two runs on the same state
    assert not d.get(MEMBER["cloud_ice"]).any()                  # an all-zero field stays all zero
    d.close()


@pytest.mark.parametrize("fct", [True, False])
def test_periodic_ring_step_function_scenario(oracle, fct):
    """The scenario of the reference's src/tests/test_mpdata.f90::test_3d (print-only there): a 3 x 3 x 100 ring, Courant
    number 0.25 along y, a step from 1 to 2, wrap-around after every advect3d, MPDATA order 2.  The wrap is done on the
    device with the halo faces themselves (my north face -> my south halo and vice versa = the periodic self-exchange).
    Device vs oracle within 1e-5 after 2 trips round the ring (792 steps: the two runs are re-synchronised every 50 steps
    so that the bound is on the scheme's arithmetic, not on 800 steps of error growth), with and without FCT; with FCT the
    step stays inside [1, 2] and is still a step."""
    nx, ny, nz, cfl = 3, 100, 3, 0.25
    f32 = np.float32
    one = lambda s: np.ones(s, f32)
    c = dict(nx=nx, ny=ny, nz=nz, dx=f32(1.0), dz_levels=one(nz), u=np.zeros((ny, nz, nx + 1), f32), v=np.full((ny + 1, nz, nx), cfl, f32),
             w=np.zeros((ny, nz, nx), f32), density=one((ny, nz, nx)), jacobian=one((ny, nz, nx)), jacobian_u=one((ny, nz, nx + 1)),
             jacobian_v=one((ny + 1, nz, nx)), jacobian_w=one((ny, nz, nx)), advection_dz=one((ny, nz, nx)))
    q = np.ones((ny, nz, nx), f32)
    q[ny // 2:] = 2.0                                            # i > ny/2 (1-based) -> 2
    q[0] = q[ny - 2]; q[ny - 1] = q[1]
    c["water_vapor"] = q.copy()
    d = single_image_domain(c)
    opt = options_t()
    opt.physics.advection = kADV_MPDATA; opt.adv_options.mpdata_order = 2; opt.adv_options.flux_corrected_transport = fct
    opt.advect_vars(["water_vapor"])
    nsteps = 2 * int((ny - 2) / cfl)
    nb, sb = d.new_buffer(d.halo_count(0, 1)), d.new_buffer(d.halo_count(1, 1))
    qo = q[None].copy()
    for it in range(nsteps):
        advect(d, opt, 1.0)                                      # dt = dx = 1: V_m = v * dt / dx = the Courant number itself
        d.halo_pack(0, 1, [0], nb); d.halo_pack(1, 1, [0], sb)  # rows ny-2 and 1 ...
        d.halo_unpack(1, 1, [0], nb); d.halo_unpack(0, 1, [0], sb)   # ... into rows 0 and ny-1
        oracle.advect(2, qo, *adv_args(c), 1.0, mpdata_order=2, fct=fct)
        qo[0, 0] = qo[0, ny - 2]; qo[0, ny - 1] = qo[0, 1]
        if it % 50 == 49 or it == nsteps - 1:
            got = d.get("water_vapor")
            assert_fields_close(got, qo[0], f"step {it}")
            qo[0] = got
    got = d.get("water_vapor")
    d.close()
    line = got[1:-1, 0, 1].astype(np.float64)
    assert abs(line.mean() - q[1:-1, 0, 1].mean()) < 1e-3       # the ring keeps its mass (the wrap rows are copies, not fluxes)
    if fct:
        assert line.min() >= 1.0 - 1e-6 and line.max() <= 2.0 + 1e-6
        assert line.max() > 1.9 and line.min() < 1.1             # ... and still has a step after two trips


def test_mpdata_fct_is_sign_preserving_and_bounded_on_a_sharp_blob(oracle):
    """adv_mpdata_FCT_core.f90 guarantees no new extrema; the fused kernel's limiter uses 1-ulp reciprocals, so its betas can
    exceed the exact ones by an ulp.  A box of ones in a field of zeros (the sharpest gradient there is), 12 steps of MPDATA +
    FCT in a sheared 3-D flow over a hill: the field must stay >= 0 EXACTLY (mixing ratios feed square roots and logarithms in
    the microphysics) wherever the oracle's does, and its maximum may exceed the ORACLE's maximum of the same step (the flow over the
    hill converges: the scheme itself raises the maximum by a few per mille) by no more than rounding (2e-6)."""
    nx, ny, nz = 70, 50, 20
    c = ideal.make_case(nx, ny, nz, hill_height=900.0, noise=0.0)
    q0 = np.zeros((ny, nz, nx), np.float32); q0[18:30, 4:12, 20:34] = 1.0
    fields = ["water_vapor", "cloud_water"]
    c["water_vapor"] = q0.copy(); c["cloud_water"] = (q0 * np.float32(3e-4)).astype(np.float32)
    opt = options_t(); opt.physics.advection = kADV_MPDATA
    opt.parameters.dz_levels = c["dz_levels"]; opt.parameters.dx = float(c["dx"])
    opt.advect_vars(fields)
    d = single_image_domain(c)
    dt = 0.9 * ideal.cfl_dt(c)
    s = np.stack([c[n] for n in fields]).copy()
    for it in range(12):
        advect(d, opt, dt)
        oracle.advect(2, s, *adv_args(c), dt)
        for m, n in enumerate(fields):
            got = d.get(MEMBER[n])
            assert s[m].min() >= 0.0
            assert got.min() >= 0.0, f"step {it} {n}: min {got.min()!r} (oracle min {s[m].min()!r})"
            assert got.max() <= float(s[m].max()) * (1 + 2e-6), f"step {it} {n}: max {got.max()!r} > oracle {s[m].max()!r}"
            s[m][...] = got                                                # the next step from equal inputs
    assert 0.2 < float(d.get("water_vapor").max()) < 1.1                 # the blob has moved and spread, not vanished
    d.close()


def test_config3_whole_domain_1024_every_cell_vs_oracle(oracle):
    """configs[3]'s whole 1024 x 1024 x 40 domain on one GPU (the largest tile the 32-bit buffer offsets of the fused kernel
    were sized for: 42 M cells x 11 coefficient arrays = 1.8 GB < 2 GiB): three scalars, one MPDATA step with advect_density,
    every cell within 1e-5 of the local scale of the CPU oracle, and the upwind scheme bit for bit."""
    nx = ny = 1024; nz = 40
    names = ["water_vapor", "potential_temperature", "cloud_water"]
    assert run_case(oracle, kADV_MPDATA, nx, ny, nz, names, dens=True, nsteps=1) <= 1e-5
    run_case(oracle, kADV_UPWIND, nx, ny, nz, names[:2], nsteps=1)
