"""HIP Thompson microphysics (rows M2-M4) vs the CPU oracle through the C ABI.

* lookup tables: every table of thompson_init is BIT-IDENTICAL to the oracle's (which is bit-identical to the compiled
  reference, tests/test_oracle_vs_ref.py): the O(1e10)-term FP64 collection integrals run on the GPU in the reference's
  summation order without FMA contraction.
* column physics: compared with the oracle in math-mode 0 -- the host's libm, i.e. what the compiled reference calls and what
  the device restates bit for bit (icar_amd/csrc/glibc_flt32.h) -- every field of every case, incl. every column of
  512 x 512 x 40: BIT-IDENTICAL (asserted: 0 differing cells).  
  The measured values are recorded by every run (gpurun_out/parity -> profiles/r0*_parity.json)."""
import ctypes
import numpy as np
import pytest
from icar_amd import ideal
from icar_amd.capi import lib, check
from icar_amd.options import options_t
from icar_amd.microphysics import mp, mp_init
from icar_amd.constants import kMP_THOMPSON
from util import single_image_domain, parity_record, field_stats, bits_equal

pytestmark = pytest.mark.gpu
TABLES = ["tcg_racg", "tmr_racg", "tcr_gacr", "tmg_gacr", "tnr_racg", "tnr_gacr", "tcs_racs1", "tmr_racs1", "tcs_racs2",
          "tmr_racs2", "tcr_sacr1", "tms_sacr1", "tcr_sacr2", "tms_sacr2", "tnr_racs1", "tnr_racs2", "tnr_sacr1", "tnr_sacr2",
          "tpi_qcfz", "tni_qcfz", "tpi_qrfz", "tpg_qrfz", "tni_qrfz", "tnr_qrfz", "tps_iaus", "tni_iaus", "tpi_ide", "t_Efrw", "t_Efsw"]
FIELDS = {"water_vapor": "water_vapor", "cloud_water": "cloud_water_mass", "rain": "rain_mass", "cloud_ice": "cloud_ice_mass",
          "snow": "snow_mass", "graupel": "graupel_mass", "ice_number": "cloud_ice_number", "rain_number": "rain_number",
          "potential_temperature": "potential_temperature"}


def device_table(d, name):
    n = ctypes.c_size_t()
    check(lib().icar_hip_thompson_table(d.ctx, name.encode(), None, ctypes.c_size_t(0), ctypes.byref(n)), "table size")
    out = np.empty(n.value, np.float64)
    check(lib().icar_hip_thompson_table(d.ctx, name.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), None), "table")
    return out


def test_lookup_tables_bit_identical(th_oracle):
    c = ideal.make_case(8, 8, 4)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    for name in TABLES:
        a = device_table(d, name); b = th_oracle.thompson_table(name)
        assert a.shape == b.shape, name
        nb = int((a.view(np.int64) != b.view(np.int64)).sum())
        assert nb == 0, f"{name}: {nb} of {a.size} entries differ, max rel {np.abs(a-b).max()/max(np.abs(b).max(),1e-300):.2e}"
    d.close()


def run_case(oracle, nx, ny, nz, steps, cool, moist, dt, mode, uniform_dz=None, mp_options=None):
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, uniform_dz=uniform_dz)
    c["water_vapor"] = (c["water_vapor"] * np.float32(moist)).astype(np.float32)
    s = {k: c[k].copy() for k in list(FIELDS) + ["exner", "pressure", "dz_mass"]}
    acc = {k: np.zeros((ny, nx), np.float64) for k in ("rain", "snow", "graupel")}
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    if mp_options is not None:
        opt.mp_options = mp_options
    mp_init(opt, d)
    oracle.set_math_mode(mode)
    try:
        for _ in range(steps):
            r = np.zeros((ny, nx), np.float32); rv = r.copy(); sn = r.copy(); gr = r.copy(); sr = r.copy()
            oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                            s["rain_number"], s["potential_temperature"], s["exner"], s["pressure"], s["dz_mass"], dt, r, rv, sn, gr, sr,
                            1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
            acc["rain"] += r; acc["snow"] += sn; acc["graupel"] += gr
            s["potential_temperature"] -= np.float32(cool)
            mp(d, opt, dt)
            d.model_time_seconds += dt
            d.set("potential_temperature", d.get("potential_temperature") - np.float32(cool))
    finally:
        oracle.set_math_mode(0)
    out = {k: d.get(m) for k, m in FIELDS.items()}
    out["acc_rain"] = d.get("accumulated_precipitation"); out["acc_snow"] = d.get("accumulated_snowfall"); out["acc_graupel"] = d.get("graupel")
    d.close()
    ref = dict(s); ref["acc_rain"] = acc["rain"]; ref["acc_snow"] = acc["snow"]; ref["acc_graupel"] = acc["graupel"]
    return out, ref


def check_close(out, ref, rtol, frac_allowed, label, abs_allowed=None):
    """Per field: the fraction of cells beyond rtol must not exceed frac_allowed and max |d| / max|field| must not exceed
    abs_allowed; the measured values go to gpurun_out/parity/ (-> profiles/r02_parity.json), the bounds are <= 2x them."""
    report = []; stats = {}
    for k in list(FIELDS) + ["acc_rain", "acc_snow", "acc_graupel"]:
        st = field_stats(out[k], ref[k], rtol); stats[k] = st
        report.append(f"{k}: bitdiff {st['bitdiff_frac']:.2e}, beyond-rtol {st['beyond_rtol_frac']:.2e}, max|d|/max {st['max_abs_over_max']:.2e}")
    parity_record("thompson", label, stats)
    print(f"[{label}]\n  " + "\n  ".join(report))
    import util
    for k, st in stats.items():
        util.COUNTS["bit_exact_fields" if (frac_allowed == 0.0 and abs_allowed == 0.0) else "tolerance_fields"] += 1
        assert st["beyond_rtol_frac"] <= frac_allowed, f"[{label}] {k}: {st}"
        if abs_allowed is not None:
            assert st["max_abs_over_max"] <= abs_allowed, f"[{label}] {k}: {st}"


# (frac_allowed, abs_allowed) per oracle math mode: <= 2x the values measured on MI355X (profiles/r02_parity.json)
#   mode 1 (the device's definition of the float transcendentals): measured 0 differing bits in every case -> asserted exact
#   mode 0 (the reference's libm): measured <= 2.1e-3 of the cells beyond rtol 1e-5 (long_dt; 5e-5 / 6e-5 in the other cases),
#           max |d| <= 1.3e-5 of the field maximum; full-size subset: no cell beyond rtol, max |d| 5.6e-6 of the maximum
FULL_SIZE_BOUNDS = {0: (0.0, 0.0)}
EXACT = dict(frac_allowed=0.0, abs_allowed=0.0)

CASES = {"warm_mixed": dict(nx=70, ny=20, nz=30, steps=10, cool=1.0, moist=1.6, dt=40.0),
         "cold_graupel": dict(nx=66, ny=18, nz=40, steps=20, cool=2.0, moist=2.0, dt=60.0),
         "long_dt": dict(nx=40, ny=12, nz=40, steps=12, cool=3.0, moist=2.5, dt=130.0)}


@pytest.mark.parametrize("case", list(CASES))
def test_thompson_bit_exact_vs_reference_math(th_oracle, case):
    out, ref = run_case(th_oracle, mode=0, **CASES[case])
    if case == "cold_graupel":
        assert ref["snow"].max() > 1e-4 and ref["graupel"].max() > 1e-5 and ref["cloud_ice"].max() > 1e-6
    assert ref["rain"].max() > 1e-5 and ref["acc_rain"].max() > 0
    check_close(out, ref, rtol=1e-5, label=case + "/mode0", **EXACT)


def test_thompson_quiet_columns_beside_active_ones(th_oracle):
    """Columns with nothing to do (:1363) beside active ones, bit for bit against the oracle (the dry half of the domain keeps
    its inputs except for what the column routine does before it returns)."""
    nx, ny, nz = 140, 11, 40
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.01)
    qv = c["water_vapor"].copy(); qv[:, :, : nx // 2] *= np.float32(0.05); qv[:, :, nx // 2:] *= np.float32(2.0)
    c["water_vapor"] = qv.astype(np.float32)
    c["cloud_water"][:, 3, 5:9] = np.float32(5e-13)          # below R1: zeroed even where the column returns early
    s = {k: c[k].copy() for k in list(FIELDS) + ["exner", "pressure", "dz_mass"]}
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    th_oracle.set_math_mode(0)
    for _ in range(4):
        r = np.zeros((ny, nx), np.float32); rv = r.copy(); sn = r.copy(); gr = r.copy(); sr = r.copy()
        th_oracle.thompson(s["water_vapor"], s["cloud_water"], s["rain"], s["cloud_ice"], s["snow"], s["graupel"], s["ice_number"],
                           s["rain_number"], s["potential_temperature"], s["exner"], s["pressure"], s["dz_mass"], 60.0, r, rv, sn, gr, sr,
                           1, nx, 1, ny, 1, nz, 2, nx - 1, 2, ny - 1, 1, nz)
        s["potential_temperature"] -= np.float32(1.5)
        mp(d, opt, 60.0); d.model_time_seconds += 60.0
        d.set("potential_temperature", d.get("potential_temperature") - np.float32(1.5))
    out = {k: d.get(m) for k, m in FIELDS.items()}
    d.close()
    dry = out["cloud_water"].max(axis=1) == 0.0                      # (ny, nx): columns without any cloud water
    assert dry.any() and (~dry).any() and out["cloud_water"][1:-1, 3, 5:9].max() == 0.0
    for k in out:
        assert bits_equal(out[k], s[k]), k


def test_thompson_excludes_last_global_row_and_column(th_oracle):
    """i_end = min(ite, ide-1), j_end = min(jte, jde-1)."""
    nx, ny, nz = 20, 12, 20
    c = ideal.make_case(nx, ny, nz, hill_height=500.0)
    c["water_vapor"] = (c["water_vapor"] * np.float32(2.0)).astype(np.float32)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    g = d.grid
    check(lib().icar_hip_thompson(d.ctx, ctypes.c_float(30.0), 1, nx, 1, ny, 1, nz, g.ids, g.ide, g.jds, g.jde, g.kds, g.kde), "thompson")
    qc = d.get("cloud_water_mass"); d.close()
    assert qc[:-1, :, :-1].max() > 0 and qc[-1].max() == 0 and qc[:, :, -1].max() == 0


@pytest.mark.parametrize("nz", [3, 12, 56, 100])
def test_thompson_other_level_counts_vs_oracle(th_oracle, nz):
    """nz=12: 21 columns per 256-thread block; nz=56: one column per wave; nz=100: 5 columns per 512-thread block (more levels than
    a level mask holds: the exchanges scan flag words); nz=3: 85 columns per block, more than a wave has lanes (a wave then holds
    one level of SOME columns: the per-wave minima of the others stay at the neutral element the kernel starts them with)."""
    out, ref = run_case(th_oracle, mode=0, nx=30 if nz > 3 else 200, ny=10, nz=nz, steps=8, cool=2.0, moist=2.0, dt=60.0,
                        uniform_dz=150.0 if nz == 100 else (2500.0 if nz == 3 else None))
    if nz > 3:
        assert ref["rain"].max() > 1e-6
    check_close(out, ref, rtol=1e-5, label=f"nz{nz}/mode0", **EXACT)


def test_halo_strips_in_one_launch_equal_four_launches():
    """icar_hip_thompson_tiles (process_halo's four strips in one launch) == four icar_hip_thompson calls."""
    from icar_amd.microphysics import mp_tiles
    nx, ny, nz = 50, 31, 40
    c = ideal.make_case(nx, ny, nz, hill_height=800.0, noise=0.01)
    c["water_vapor"] = (c["water_vapor"] * np.float32(2.2)).astype(np.float32)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    outs = []
    for batched in (True, False):
        d = single_image_domain(c); mp_init(opt, d); g = d.grid
        for step in range(3):
            if batched:
                tiles = mp_tiles(g.its, g.ite, g.jts, g.jte, halo=1)
                arr = ((ctypes.c_int * 4) * 4)(*[(ctypes.c_int * 4)(*t) for t in tiles])
                check(lib().icar_hip_thompson_tiles(d.ctx, ctypes.c_float(60.0), 4, arr, g.kts, g.kte,
                                                    g.ids, g.ide, g.jds, g.jde, g.kds, g.kde), "thompson_tiles")
            else:
                for (a, b, cc, dd) in mp_tiles(g.its, g.ite, g.jts, g.jte, halo=1):
                    check(lib().icar_hip_thompson(d.ctx, ctypes.c_float(60.0), a, b, cc, dd, g.kts, g.kte,
                                                  g.ids, g.ide, g.jds, g.jde, g.kds, g.kde), "thompson")
            d.model_time_seconds += 60.0
            d.set("potential_temperature", d.get("potential_temperature") - np.float32(2.0))
        outs.append({k: d.get(m) for k, m in FIELDS.items()} | {"acc": d.get("accumulated_precipitation")})
        d.close()
    ring = np.zeros((ny, nx), bool); ring[1, 1:-1] = ring[-2, 1:-1] = True; ring[1:-1, 1] = ring[1:-1, -2] = True
    assert (outs[0]["cloud_water"][:, 5, :][ring] > 0).any(), "the strips must have done some work"
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    inner = outs[0]["cloud_water"][2:-2, :, 2:-2]
    assert inner.max() == 0.0, "only the halo ring is processed"


def test_thompson_full_size_every_column_bit_exact(th_oracle):
    """The BASELINE tile (512 x 512 x 40): EVERY column of two microphysics calls, device vs the CPU oracle in the reference's own
    math, bit for bit (the column subset above only samples 3000 of the 260 100)."""
    out, ref = run_case(th_oracle, mode=0, nx=512, ny=512, nz=40, steps=2, cool=1.5, moist=1.8, dt=60.0)
    assert ref["rain"].max() > 1e-5 and ref["cloud_water"].max() > 1e-5 and ref["acc_rain"].max() > 0
    check_close(out, ref, rtol=1e-5, label="full_size_every_column/mode0", **EXACT)


@pytest.mark.parametrize("mode", [0])
def test_thompson_full_size_budget_and_column_subset_vs_oracle(th_oracle, mode):
    """BASELINE size (512x512x40): (a) every species stays non-negative and finite, (b) the column water budget closes to within
    1 % (microphysics only moves water between species / levels / the surface), (c) 3000 random columns, re-run by the CPU
    oracle as a small domain of their own (the scheme is column-local), agree with the device to the usual tolerance."""
    nx = ny = 512; nz = 40; dt = 60.0; steps = 3
    c = ideal.make_case(nx, ny, nz, hill_height=1000.0, noise=0.01, n_hydro=1)
    c["water_vapor"] = (c["water_vapor"] * np.float32(1.8)).astype(np.float32)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    rng = np.random.default_rng(11)
    jj = rng.integers(1, ny - 2, 3000); ii = rng.integers(1, nx - 2, 3000)          # inside its..ite-1 / jts..jte-1 (F7)
    keys = list(FIELDS)
    sub = {k: np.ascontiguousarray(np.stack([c[k][jj, :, ii].T] * 3, axis=0)[:, :, :]) for k in keys + ["exner", "pressure", "dz_mass"]}
    # sub[k]: (3, nz, 3000); pad one column on each side so that its=2..ite=n-1 covers all of them
    sub = {k: np.ascontiguousarray(np.pad(v, ((0, 0), (0, 0), (1, 1)), mode="edge")) for k, v in sub.items()}
    n = sub["pressure"].shape[2]

    t0 = c["potential_temperature"].astype(np.float64) * c["exner"]
    rho0 = 0.622 * c["pressure"] / (287.04 * t0 * (c["water_vapor"].astype(np.float64) + 0.622))      # the scheme's own density, initial

    def water_path(f):
        # conversions conserve the total mixing ratio of a level exactly; fall fluxes are mass fluxes of the CURRENT density,
        # which drifts from rho0 by the latent heating (<1 %), so the budget closes to a small fraction of what fell
        q = sum(f[k].astype(np.float64) for k in ("water_vapor", "cloud_water", "rain", "cloud_ice", "snow", "graupel"))
        return (q * rho0 * c["dz_mass"]).sum(axis=1)
    before = water_path(c)
    th_oracle.set_math_mode(mode)
    try:
        for _ in range(steps):
            mp(d, opt, dt); d.model_time_seconds += dt
            z = [np.zeros((3, n), np.float32) for _ in range(5)]
            th_oracle.thompson(sub["water_vapor"], sub["cloud_water"], sub["rain"], sub["cloud_ice"], sub["snow"], sub["graupel"],
                               sub["ice_number"], sub["rain_number"], sub["potential_temperature"], sub["exner"], sub["pressure"],
                               sub["dz_mass"], dt, *z, 1, n, 1, 3, 1, nz, 2, n - 1, 2, 2, 1, nz)
    finally:
        th_oracle.set_math_mode(0)
    out = {k: d.get(m) for k, m in FIELDS.items()}
    precip = d.get("accumulated_precipitation")
    d.close()
    for k, a in out.items():
        assert np.isfinite(a).all(), k
        if k != "potential_temperature":
            assert a.min() >= 0.0, f"{k}: negative values"
    assert out["rain"].max() > 1e-5 and out["cloud_water"].max() > 1e-5 and precip.max() > 0
    after = water_path(out)
    inner = (slice(1, ny - 1), slice(1, nx - 1))
    resid = np.abs(after[inner] + precip[inner] - before[inner])                        # kg m-2 ; precipitation is in mm = kg m-2
    print(f"column water budget: max residual {resid.max():.3e} kg m-2 of {before[inner].max():.3e}, max precipitation {precip.max():.3e}")
    # the scheme itself (== the CPU oracle == the reference, bit for bit) closes this budget only to ~2e-3 of the column
    # water per step in this deliberately messy state (every species present at every level); the bound catches gross
    # errors such as a surface flux counted twice, the column subset below is the sharp check
    assert resid.max() <= 0.01 * before[inner].max(), f"column water budget residual {resid.max():.3e} of {before[inner].max():.3e}"
    got = {k: out[k][jj, :, ii].T for k in keys}                                          # (nz, 3000)
    ref = {k: sub[k][1, :, 1:-1] for k in keys}
    stats = {k: field_stats(got[k], ref[k], 1e-5) for k in keys}
    parity_record("thompson", f"full_size_subset/mode{mode}", stats)
    frac_allowed, abs_allowed = FULL_SIZE_BOUNDS[mode]
    for k, st in stats.items():
        assert st["beyond_rtol_frac"] <= frac_allowed and st["max_abs_over_max"] <= abs_allowed, f"mode {mode} {k}: {st}"


def test_table_cache_files_byte_identical_to_the_reference(tmp_path):
    """icar_amd.thompson_cache writes qr_acr_qg_mpt.dat / qr_acr_qs_mpt.dat / freezeH2O_mpt.dat from the DEVICE tables in the
    reference's Fortran unformatted-sequential layout; their SHA-256 must equal the digests of the files the compiled
    reference wrote (tests/golden/thompson_cache_sha256.json) -- i.e. a reference run can consume them as its own."""
    import hashlib, json, os
    from icar_amd import thompson_cache as tc
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "thompson_cache_sha256.json")))["files"]
    c = ideal.make_case(8, 8, 4)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    tc.write_caches(d, str(tmp_path))
    d.close()
    for f, g in gold.items():
        p = os.path.join(str(tmp_path), f)
        assert os.path.getsize(p) == g["bytes"], f
        assert hashlib.sha256(open(p, "rb").read()).hexdigest() == g["sha256"], f"{f} differs from the reference's file"
    back = tc.read_caches(str(tmp_path))
    assert len(back) == 24 and back["tcg_racg"].size == 28 * 28 * 37 * 37


def test_non_default_mp_options(oracle):
    """A second mp_options set (all 18 parameters changed, both efficiency-table flags on; the oracle is pinned to the reference
    for it by tests/test_oracle_thompson.py::test_non_default_mp_options_tables_and_columns): the device tables are bit-identical
    to the oracle's and the cold-graupel column case agrees like the default set does."""
    from icar_amd.options import mp_options_type
    mpo = mp_options_type(Nt_c=50.e6, TNO=4.0, am_s=0.08, rho_g=400.0, av_s=35.0, bv_s=0.5, fv_s=80.0, av_g=400.0, bv_g=0.85, av_i=1800.0,
                          Ef_si=0.07, Ef_rs=0.9, Ef_rg=0.7, Ef_ri=0.9, C_cubes=0.4, C_sqrd=0.25, mu_r=1.0, t_adjust=1.0, Ef_rw_l=True, Ef_sw_l=True)
    p, f = mpo.as_arrays()
    oracle.thompson_init(p, f)
    try:
        c = ideal.make_case(8, 8, 4)
        d = single_image_domain(c)
        opt = options_t(); opt.physics.microphysics = kMP_THOMPSON; opt.mp_options = mpo
        mp_init(opt, d)
        for name in TABLES:
            a = device_table(d, name); b = oracle.thompson_table(name)
            nb = int((a.view(np.int64) != b.view(np.int64)).sum())
            assert nb == 0, f"{name}: {nb} of {a.size} entries differ"
        d.close()
        out, ref = run_case(oracle, mode=0, mp_options=mpo, **CASES["cold_graupel"])
        assert ref["snow"].max() > 1e-4 and ref["graupel"].max() > 1e-5
        check_close(out, ref, rtol=1e-5, label="alt/mode0", **EXACT)
    finally:
        po, fo = options_t().mp_options.as_arrays()
        oracle.thompson_init(po, fo)


def test_decade_index_fast_form_equals_reference_loop(probe):
    """The table indices (mp_thompson.f90:1562-1627) are integer-exact rows: the level code takes the decade from the hardware log2
    and one division by the tabulated 10.**n, and runs the reference's loop (nint(log10 r), 10.**n by repeated squaring, the
    [1, 10) test) only near a power of ten.  Both forms, value by value: log-uniform random arguments over every decade the scheme
    can produce, every REAL(4) within 200 ulps of a power of ten, and the table's own grid values."""
    import ctypes
    from icar_amd.capi import lib, check
    c = ideal.make_case(12, 6, 12)
    d = single_image_domain(c)
    opt = options_t(); opt.physics.microphysics = kMP_THOMPSON
    mp_init(opt, d)
    rng = np.random.default_rng(5)
    r4 = [10.0 ** rng.uniform(-13.5, 9.5, 2_000_000)]
    for e in range(-13, 10):
        p = np.float32(10.0) ** np.float32(e)
        for base in (p, np.float32(float(10.0 ** e))):
            bits = np.array([base], np.float32).view(np.int32)[0]
            r4.append(np.arange(bits - 200, bits + 201, dtype=np.int32).view(np.float32).astype(np.float64))
        r4.append(p * np.arange(1, 10, dtype=np.float64))                      # the decade's grid points 1e.., 2e.., ...
        f = np.linspace(1.5e-4, 3.5e-4, 400)                                   # either side of where the fast form hands over to the loop
        r4.append(10.0 ** (e + f)); r4.append(10.0 ** (e - f))
    r4 = np.ascontiguousarray(np.concatenate(r4).astype(np.float32))
    r8 = np.ascontiguousarray(np.concatenate([10.0 ** rng.uniform(0.5, 13.5, 1_000_000),
                                              np.concatenate([np.nextafter(10.0 ** e, np.inf) * (1 + np.arange(-50, 51) * 2.0 ** -40) for e in range(1, 14)]),
                                              np.concatenate([10.0 ** e * np.arange(1, 10) for e in range(1, 14)])]))
    for arr, is4 in ((r4, True), (r8, False)):
        for n2 in (-12, -6, 0, 2):
            fast = np.zeros(arr.size, np.int32); slow = np.zeros(arr.size, np.int32)
            p4 = arr.ctypes.data_as(ctypes.c_void_p) if is4 else None
            p8 = None if is4 else arr.ctypes.data_as(ctypes.c_void_p)
            assert probe.icar_probe_dec_index(p4, p8, arr.size, n2, 0, fast.ctypes.data_as(ctypes.c_void_p)) == 0
            assert probe.icar_probe_dec_index(p4, p8, arr.size, n2, 1, slow.ctypes.data_as(ctypes.c_void_p)) == 0
            bad = np.flatnonzero(fast != slow)
            assert bad.size == 0, (is4, n2, bad.size, arr[bad[:5]], fast[bad[:5]], slow[bad[:5]])
    d.close()


def test_fp64_transcendentals_of_the_level_code(oracle, probe):
    """The DOUBLE PRECISION log / exp / x**y of the level code are the C library's log / exp / pow restated (icar_amd/csrc/glibc_dbl64.h:
    glibc 2.35's FMA builds, operation by operation): on the device, bit for bit against the host libm on millions of arguments
    per function -- the scheme's ranges, every binade, arguments next to 1, arbitrary bit patterns, the special values.  (Until
    round 4 these were FP64 polynomials of our own, < 1 ulp of the double: one REAL(4) ulp away from the reference in ~1e-7 of the
    cells of a step.)  The same header is checked on the CPU in tests/test_glibc_dbl64_host.py."""
    import ctypes
    from icar_amd.capi import lib, check
    from util import parity_record
    c = ideal.make_case(12, 6, 12)
    d = single_image_domain(c)
    rng = np.random.default_rng(11)

    def run(op, x, y=None):
        x = np.ascontiguousarray(x, np.float64); out = np.zeros(x.size, np.float64)
        yp = None if y is None else np.ascontiguousarray(y, np.float64).ctypes.data_as(ctypes.c_void_p)
        assert probe.icar_probe_math(op, x.size, x.ctypes.data_as(ctypes.c_void_p), yp, out.ctypes.data_as(ctypes.c_void_p)) == 0, "math_probe"
        return out

    def differ(a, b):
        return ~((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b)))

    n = 2_000_000
    anybits = lambda m: rng.integers(0, 2 ** 63, m, dtype=np.uint64).view(np.float64) * rng.choice([-1.0, 1.0], m)
    special = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 3.0, -3.0, 1e-320, -1e-320, 2.0 ** -1022, 2.0 ** 1023, -2.0 ** 1023, np.inf, -np.inf,
                        np.nan, 2.0 ** -70, -2.0 ** -70, 2.0 ** 70, 1.5, 2.5, 1024.0, -1075.0, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0), 709.0, -745.0, 1e-20])
    stats = {}
    import warnings
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    try:
        xl = np.concatenate([(10.0 ** rng.uniform(-37.5, 38.0, n)).astype(np.float32).astype(np.float64), 10.0 ** rng.uniform(-307, 308, n),
                             1.0 + rng.uniform(-0.07, 0.07, n // 2), 1.0 + rng.uniform(-1e-9, 1e-9, n // 10), np.abs(anybits(n // 2)), anybits(n // 10), special])
        xe = np.concatenate([rng.uniform(-87.3, 88.7, n).astype(np.float32).astype(np.float64), rng.uniform(-750.0, 715.0, n),
                             rng.uniform(-1.0, 1.0, n // 2) * 2.0 ** -rng.integers(0, 70, n // 2), anybits(n // 10), special])
        for op, name, x in ((0, "log", xl), (1, "exp", xe)):
            got, want = run(op, x), oracle.libm_d(op, x)
            bad = differ(got, want)
            stats[name] = {"n": int(x.size), "differ": int(bad.sum())}
            assert not bad.any(), f"{name}: {bad.sum()} of {x.size} differ from libm, first x = {x[bad][0]!r}: {got[bad][0]!r} vs {want[bad][0]!r}"
        # x**y: the scheme's use (REAL(4) and DOUBLE PRECISION positive bases, moderate exponents), quarter-integer exponents, results near
        # over- / underflow, arbitrary bit patterns, the grid of special values
        xb = np.concatenate([(10.0 ** rng.uniform(-12.0, 12.0, n)).astype(np.float32).astype(np.float64), 10.0 ** rng.uniform(-40.0, 40.0, n),
                             10.0 ** rng.uniform(-20.0, 20.0, n // 2), 2.0 ** rng.uniform(-1074, 1024, n // 2), anybits(n // 4), np.abs(anybits(n // 4))])
        yb = np.concatenate([rng.uniform(-6.0, 6.0, n).astype(np.float32).astype(np.float64), rng.uniform(-12.0, 12.0, n),
                             0.25 * rng.integers(-48, 49, n // 2), rng.uniform(-1.1, 1.1, n // 2), anybits(n // 4), rng.uniform(-0.5, 0.5, n // 4) * 2.0 ** rng.integers(-10, 14, n // 4)])
        yb[2 * n + n // 2: 3 * n] *= 1075.0 / np.maximum(1.0, np.abs(np.log2(xb[2 * n + n // 2: 3 * n])))
        sx, sy = np.meshgrid(special, special)
        xb = np.concatenate([xb, sx.ravel()]); yb = np.concatenate([yb, sy.ravel()])
        got, want = run(2, xb, yb), oracle.libm_d(2, xb, yb)
        bad = differ(got, want)
        stats["pow"] = {"n": int(xb.size), "differ": int(bad.sum())}
        assert not bad.any(), f"pow: {bad.sum()} of {xb.size} differ from libm, first ({xb[bad][0]!r}, {yb[bad][0]!r}): {got[bad][0]!r} vs {want[bad][0]!r}"
        parity_record("thompson", "DOUBLE PRECISION log / exp / pow of the level code vs the host libm (bit patterns)", stats)
    finally:
        d.close()
