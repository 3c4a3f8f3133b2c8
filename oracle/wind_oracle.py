"""oracle/wind_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (numpy, FP64 complex / FP32 real exactly where the reference uses them) of the
linear-theory wind LUT build, SURVEY.md section 8 row W3:

    fftshift / ifftshift            src/utilities/fftshift.f90:95-117, 217-239  (F9: single-precision temp)
    add_buffer_topo                 src/physics/linear_winds.f90:351-418
    initialize_linear_theory_data   src/physics/linear_winds.f90:426-499
    setup_linwinds (terrain FFT)    src/physics/linear_winds.f90:1180-1225
    linear_perturbation_at_height   src/physics/linear_winds.f90:181-237
    linear_perturbation_constz      src/physics/linear_winds.f90:239-276
    linear_perturbation_varyingz    src/physics/linear_winds.f90:280-344
    initialize_spatial_winds        src/physics/linear_winds.f90:596-830   (LUT loop, destagger :766-772)
    linear_space / calc_u / calc_v  src/utilities/array_utilities.f90:215-237, atm_utilities.f90:373-391

Pinned by execution (tests/test_oracle_helpers_vs_ref.py, vs the unmodified utility modules in oracle/_ref):
linear_space, calc_u, calc_v.
PARITY UNPINNED for everything else: linear_winds.f90 cannot be built in this image (it needs FFTW3's fftw3.f03 and the
NetCDF/coarray-dependent domain object) and the reference's own test for it
(tests/test_caf_linear_winds_setup.f90) is a smoke test without expected values.  Third-party
arithmetic: FFTW3 (system package, unpinned; CI image Ubuntu 20.04 libfftw3 3.3.8); its documented
transform -- unnormalised, sign -1 forward / +1 backward, fftw_plan_dft_2d(ny,nx) == full 2-D DFT of
the Fortran (nx,ny) array -- is restated with numpy.fft (pocketfft, FP64).  The only known-answer
material is the tests/test_fftshift.f90 scenario (n=5, x(i,j)=i+50j, shift then unshift == identity),
checked in tests/test_oracle_winds.py.

Arrays are indexed [i, j] like the Fortran (i,j) (0-based); real(4) quantities are np.float32 and
every float32 operation is done in float32 in the reference's order.
"""
import ctypes
import math
import numpy as np

_libm = ctypes.CDLL("libm.so.6")           # sin/cos/exp of REAL(4) scalars: glibc's float functions, as gfortran/flang call them
for _n in ("sinf", "cosf", "expf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]

f32 = np.float32
PI = f32(3.1415927)            # icar_constants.f90:395
SMALL_VALUE = f32(1e-15)       # linear_winds.f90:112


# ---------------------------------------------------------------- fftshift.f90
def _shift_index(n):
    """ii(i) = mod(i + (n+1)/2, n), 0 -> n   (1-based), returned 0-based."""
    i = np.arange(1, n + 1)
    ii = (i + (n + 1) // 2) % n
    ii[ii == 0] = n
    return ii - 1


def _to_single(a):
    """complex(8) -> default complex temp -> complex(8): the F9 quirk of fftshift2cc / ifftshift2cc."""
    return a.astype(np.complex64).astype(np.complex128)


def fftshift2cc(a):                       # fftshift.f90:95-117   tmp(ii,jj) = f(i,j)
    nx, ny = a.shape
    out = np.empty_like(a)
    out[np.ix_(_shift_index(nx), _shift_index(ny))] = a
    return _to_single(out)


def ifftshift2cc(a):                      # fftshift.f90:217-239  tmp(i,j) = f(ii,jj)
    nx, ny = a.shape
    return _to_single(a[np.ix_(_shift_index(nx), _shift_index(ny))])


def fftshift2r(a):                        # fftshift.f90:141-163 (real arrays, no precision change)
    nx, ny = a.shape
    out = np.empty_like(a)
    out[np.ix_(_shift_index(nx), _shift_index(ny))] = a
    return out


def ifftshift2r(a):
    nx, ny = a.shape
    return a[np.ix_(_shift_index(nx), _shift_index(ny))].copy()


# ---------------------------------------------------------------- add_buffer_topo
def _seqsum(bt, xs, xe, ys, ye):
    """Fortran SUM over bt(xs:xe, ys:ye) (1-based inclusive), column-major sequential order."""
    acc = 0.0
    sub = bt[xs - 1:xe, ys - 1:ye].real
    for jj in range(sub.shape[1]):
        for ii in range(sub.shape[0]):
            acc += float(sub[ii, jj])
    return acc


def add_buffer_topo(terrain, smooth_window, buffer):
    """linear_winds.f90:351-418.  terrain: float32 [nx_t, ny_t].  Returns complex128 [nx_t+2b, ny_t+2b]."""
    terrain = np.asarray(terrain, f32)
    b = int(buffer)
    tx, ty = terrain.shape
    nx, ny = tx + 2 * b, ty + 2 * b
    bt = np.full((nx, ny), complex(float(terrain.min()), 0.0), np.complex128)
    bt[b:nx - b, b:ny - b] = terrain
    for i in range(1, b + 1):
        weight = f32(i) / (f32(b) * f32(2))
        pos = b - i
        omw = f32(1) - weight
        bt[pos, b:ny - b] = terrain[0, :] * omw + terrain[tx - 1, :] * weight
        bt[nx - pos - 1, b:ny - b] = terrain[0, :] * weight + terrain[tx - 1, :] * omw
    for i in range(1, b + 1):
        weight = f32(i) / (f32(b) * f32(2))
        pos = b - i
        omw = float(f32(1) - weight); w = float(weight)
        lo = bt[:, b].copy(); hi = bt[:, ny - b - 1].copy()
        bt[:, pos] = lo * omw + hi * w
        # the second statement reads buffer_topo(:,buffer+1) / (:,ny-buffer) again; neither was modified
        bt[:, ny - pos - 1] = lo * w + hi * omw
    if smooth_window > 0:
        for j in range(1, b + 1):
            window = min(j, smooth_window)
            for i in range(1, nx + 1):
                xs = max(1, i - window); xe = min(nx, i + window)
                ys = max(1, b - j + 1 - window); ye = min(ny, b - j + 1 + window)
                bt[i - 1, b - j] = _seqsum(bt, xs, xe, ys, ye) / ((xe - xs + 1) * (ye - ys + 1))
                ys = max(1, ny - (b - j) - window); ye = min(ny, ny - (b - j) + window)
                bt[i - 1, ny - (b - j) - 1] = _seqsum(bt, xs, xe, ys, ye) / ((xe - xs + 1) * (ye - ys + 1))
            for i in range(1, ny + 1):
                xs = max(1, b - j + 1 - window); xe = min(nx, b - j + 1 + window)
                ys = max(1, i - window); ye = min(ny, i + window)
                bt[b - j, i - 1] = _seqsum(bt, xs, xe, ys, ye) / ((xe - xs + 1) * (ye - ys + 1))
                xs = max(1, nx - (b - j) - window); xe = min(nx, nx - (b - j) + window)
                bt[nx - (b - j) - 1, i - 1] = _seqsum(bt, xs, xe, ys, ye) / ((xe - xs + 1) * (ye - ys + 1))
    return bt


# ---------------------------------------------------------------- FFTW restated
def fft2_forward(a):        # fftw_plan_dft_2d(ny,nx,..,FFTW_FORWARD): unnormalised, exp(-i...)
    return np.fft.fft2(a)


def fft2_backward(a):       # FFTW_BACKWARD: unnormalised, exp(+i...)
    return np.fft.ifft2(a) * (a.shape[0] * a.shape[1])


class lt_data_t:
    """linear_theory_type (src/main/data_structures.f90:184-195): k, l, kl are real(4) [nx, ny]."""

    def __init__(self, nx, ny, dx):                       # linear_winds.f90:426-470
        dx = f32(dx)
        offset = PI / dx
        gain = f32(2) * offset / f32(nx - 1)
        k1 = (np.arange(nx, dtype=f32) * gain - offset).astype(f32)
        gain = f32(2) * offset / f32(ny - 1)
        l1 = (np.arange(ny, dtype=f32) * gain - offset).astype(f32)
        self.k1, self.l1 = k1, l1
        self.k = np.repeat(k1[:, None], ny, axis=1)
        self.l = np.repeat(l1[None, :], nx, axis=0)
        kl = (self.k * self.k + self.l * self.l).astype(f32)
        kl[kl == 0] = SMALL_VALUE
        self.kl = kl
        self.nx, self.ny = nx, ny


def setup_linwinds(global_terrain, dx, buffer, smooth_window=5):
    """linear_winds.f90:1180-1225.  Returns (terrain_frequency complex128 [fftnx,fftny], lt_data, total buffer)."""
    first = add_buffer_topo(global_terrain, smooth_window, buffer)
    second = add_buffer_topo(first.real.astype(f32), 0, 2)
    nx, ny = second.shape
    tf = fft2_forward(second)
    tf = tf / float(nx * ny)
    tf = fftshift2cc(tf)
    return tf, lt_data_t(nx, ny, dx), buffer + 2


def linear_perturbation_at_height(U, V, Nsq, z, hhat, lt):
    """linear_winds.f90:181-237.  Returns complex128 (u_perturb, v_perturb) [fftnx, fftny]."""
    U, V, Nsq, z = f32(U), f32(V), f32(Nsq), f32(z)
    if U == 0 and V == 0:
        zero = np.zeros((lt.nx, lt.ny), np.complex128)
        return zero, zero.copy()
    sig = (U * lt.k + V * lt.l).astype(f32)
    sig[sig == 0] = SMALL_VALUE
    denom = (sig * sig).astype(f32).astype(np.float64)           # sig**2 is real(4), stored complex(8)
    msq = (float(Nsq) / denom) * lt.kl.astype(np.float64)         # real >= 0, imaginary part 0
    mr = np.sqrt(msq)                                             # m = sqrt(msq)
    mr = np.where(sig < 0, -mr, mr)                               # m * sign(sig)
    # real(msq) < 0 never happens (Nsq, sig**2, kl > 0): the evanescent branch :212-214 is dead code,
    # kept here for the restatement's completeness
    neg = msq < 0
    if neg.any():
        raise AssertionError("evanescent branch reached: complex m is not restated")
    theta = mr * float(z)                                         # imaginary_number*m*z = (-0, mr*z)
    er, ei = np.cos(theta), np.sin(theta)
    a, b = -hhat.imag, hhat.real                                  # imaginary_number * fourier_terrain
    tr = a * er - b * ei
    ti = a * ei + b * er
    q = lt.kl.astype(np.float64) / ((0.0 - mr) * sig.astype(np.float64))   # kl / ((0-m)*sig), imaginary 0
    ir, ii = tr / q, ti / q
    kd, ld = lt.k.astype(np.float64), lt.l.astype(np.float64)
    uhat = (kd * ir) + 1j * (kd * ii)
    vhat = (ld * ir) + 1j * (ld * ii)
    uhat = ifftshift2cc(uhat)
    vhat = ifftshift2cc(vhat)
    return fft2_backward(uhat), fft2_backward(vhat)


def n_steps_of(z_bottom, z_top, minimum_step):
    z_bottom, z_top, minimum_step = f32(z_bottom), f32(z_top), f32(minimum_step)
    return max(1, int(math.ceil(float((z_top - z_bottom) / minimum_step))))


def linear_perturbation_constz(U, V, Nsq, z_bottom, z_top, minimum_step, hhat, lt):
    """linear_winds.f90:239-276."""
    U, V = f32(U), f32(V)
    if U == 0 and V == 0:
        zero = np.zeros((lt.nx, lt.ny), np.complex128)
        return zero, zero.copy()
    z_bottom, z_top = f32(z_bottom), f32(z_top)
    n_steps = n_steps_of(z_bottom, z_top, minimum_step)
    step_size = (z_top - z_bottom) / f32(n_steps)
    ua = np.zeros((lt.nx, lt.ny), np.complex128); va = np.zeros_like(ua)
    current_z = z_bottom + step_size / f32(2)
    for _ in range(n_steps):
        up, vp = linear_perturbation_at_height(U, V, Nsq, current_z, hhat, lt)
        ua = ua + up; va = va + vp
        current_z = current_z + step_size
    return ua / float(n_steps), va / float(n_steps)


def linear_perturbation_varyingz(U, V, Nsq, z_bottom, z_top, minimum_step, hhat, lt, buffer):
    """linear_winds.f90:280-344.  z_bottom/z_top float32 [nx_g, ny_g]; buffer = the module's total buffer."""
    U, V = f32(U), f32(V)
    if U == 0 and V == 0:
        zero = np.zeros((lt.nx, lt.ny), np.complex128)
        return zero, zero.copy()
    z_bottom = np.asarray(z_bottom, f32); z_top = np.asarray(z_top, f32)
    start_z = z_bottom.min(); end_z = z_top.max()
    b0 = buffer - 1                                               # internal_z(buffer:buffer+n-1) 1-based
    izt = np.full((lt.nx, lt.ny), end_z, f32); izt[b0:b0 + z_top.shape[0], b0:b0 + z_top.shape[1]] = z_top
    izb = np.full((lt.nx, lt.ny), start_z, f32); izb[b0:b0 + z_bottom.shape[0], b0:b0 + z_bottom.shape[1]] = z_bottom
    layer_count = np.zeros((lt.nx, lt.ny), f32)
    step_size = min(f32(minimum_step), (z_top - z_bottom).min())
    half = step_size / f32(2)
    current_z = start_z + half
    ua = np.zeros((lt.nx, lt.ny), np.complex128); va = np.zeros_like(ua)
    zero = f32(0)
    while current_z < end_z:
        up, vp = linear_perturbation_at_height(U, V, Nsq, current_z, hhat, lt)
        frac = (np.maximum(zero, ((np.minimum(half, current_z - izb) + np.minimum(zero, izt - current_z))
                                  + np.minimum(half, izt - current_z)) + np.minimum(zero, current_z - izb)) / step_size).astype(f32)
        layer_count = layer_count + frac
        ua = ua + up * frac.astype(np.float64); va = va + vp * frac.astype(np.float64)
        current_z = current_z + step_size
    lc = layer_count.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ua / lc, va / lc


# ---------------------------------------------------------------- LUT build
def linear_space(vmin, vmax, n):                                   # array_utilities.f90:215-237
    i = np.arange(1, n + 1, dtype=f32)
    return ((i - f32(1.0)) / f32(f32(n) - f32(1.0)) * (f32(vmax) - f32(vmin)) + f32(vmin)).astype(f32)


def calc_u(direction, magnitude):                                  # atm_utilities.f90:373-379
    return f32(f32(_libm.sinf(float(f32(direction)))) * f32(magnitude))


def calc_v(direction, magnitude):                                  # atm_utilities.f90:385-391
    return f32(f32(_libm.cosf(float(f32(direction)))) * f32(magnitude))


def destagger(up, vp, buffer, fftnx, fftny):
    """linear_winds.f90:766-772 (buffer = total buffer).  Returns float32 temporary_u [nxg+1, nyg], temporary_v [nxg, nyg+1]."""
    b = buffer
    # u_perturb(buffer:fftnx-buffer, 1+buffer:fftny-buffer) + u_perturb(1+buffer:fftnx-buffer+1, same)
    tu = (up[b - 1:fftnx - b, b:fftny - b] + up[b:fftnx - b + 1, b:fftny - b]).real.astype(f32) / f32(2)
    tv = (vp[b:fftnx - b, b - 1:fftny - b] + vp[b:fftnx - b, b:fftny - b + 1]).real.astype(f32) / f32(2)
    return tu.astype(f32), tv.astype(f32)


def build_lut(hhat, lt, buffer, z_bottom, z_top, opt, varying=None):
    """initialize_spatial_winds loop :702-783 for one image (num_images()==1).

    z_bottom/z_top: float32 [nz] layer bounds (constant-z branch :750-762) or, with varying=True,
    [nz][nx_g, ny_g] arrays (space_varying_dz branch :743-748).
    opt: dict with dirmin,dirmax,spdmin,spdmax,nsqmin,nsqmax,n_dir_values,n_spd_values,n_nsq_values,minimum_layer_size.
    Returns (u_LUT [n_spd,n_dir,n_nsq,nxg+1,nz,nyg], v_LUT [..., nxg, nz, nyg+1]) float32, value tables.
    """
    dirv = linear_space(opt["dirmin"], opt["dirmax"], opt["n_dir_values"])
    nsqv = linear_space(opt["nsqmin"], opt["nsqmax"], opt["n_nsq_values"])
    spdv = linear_space(opt["spdmin"], opt["spdmax"], opt["n_spd_values"])
    nd, nn, ns = len(dirv), len(nsqv), len(spdv)
    nz = len(z_bottom)
    fftnx, fftny = lt.nx, lt.ny
    nxg, nyg = fftnx - 2 * buffer, fftny - 2 * buffer
    ulut = np.zeros((ns, nd, nn, nxg + 1, nz, nyg), f32)
    vlut = np.zeros((ns, nd, nn, nxg, nz, nyg + 1), f32)
    for ijk in range(nd * ns * nn):
        ik = ijk // nn
        j = ijk % nn
        i = ik // ns
        k = ik % ns
        u = calc_u(dirv[i], spdv[k])
        v = calc_v(dirv[i], spdv[k])
        nsq = f32(_libm.expf(float(nsqv[j])))
        for z in range(nz):
            if varying:
                up, vp = linear_perturbation_varyingz(u, v, nsq, z_bottom[z], z_top[z], opt["minimum_layer_size"], hhat, lt, buffer)
            else:
                up, vp = linear_perturbation_constz(u, v, nsq, z_bottom[z], z_top[z], opt["minimum_layer_size"], hhat, lt)
            tu, tv = destagger(up, vp, buffer, fftnx, fftny)
            ulut[k, i, j, :, z, :] = tu
            vlut[k, i, j, :, z, :] = tv
    return ulut, vlut, dirv, spdv, nsqv
