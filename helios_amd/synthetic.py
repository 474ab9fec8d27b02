"""Synthetic opacity tables and planet columns of the shapes named in BASELINE.json.

There is no network for the real HDF5 k-tables, so benchmarks, smoke() and most tests run on
synthetic inputs built exactly as SURVEY.md §8(d) prescribes (seeded numpy Generator, fp64).  All
arrays are produced in the reference's flat layouts because that is what the
reference's readers hand to its `Store` (source/read.py:1041-1103):

    k-table      [y + ny*x + ny*nbin*p + ny*nbin*npress*t]
    Rayleigh     [x + nbin*p + nbin*npress*t]
    mean mass    [p + npress*t]
"""
import numpy as np
from numpy.polynomial.legendre import leggauss

from . import phys_const as pc


def wavelength_grid(nbin):
    """log-spaced bin edges 0.3 - 500 micron in cm (mimics the fixed-resolution grid of
    ktable/source_ktable/build_individual_opacities.py:128-139)."""
    x = np.arange(nbin + 1, dtype=np.float64)
    interwave = 0.3e-4 * (500.0 / 0.3) ** (x / nbin)
    wave = 0.5 * (interwave[1:] + interwave[:-1])
    deltawave = interwave[1:] - interwave[:-1]
    return interwave, wave, deltawave


def gauss_points(ny):
    """Gauss-Legendre abscissae mapped to (0,1) and weights summing to 2 (SURVEY.md Q6;
    source/host_functions.py:222, ktable/source_ktable/build_individual_opacities.py:221-223)."""
    if ny == 1:
        return np.array([0.5]), np.array([2.0])
    yy, ww = leggauss(ny)
    return 0.5 * yy + 0.5, ww.astype(np.float64)


def tp_grid(ntemp=30, npress=20):
    """uniform-in-T, uniform-in-log10(P) table nodes (the kernels assume uniformity, Q5)."""
    return np.linspace(100.0, 3000.0, ntemp), np.logspace(0.0, 9.0, npress)


def _smooth5(a):
    k = np.ones(5) / 5.0
    pad = np.concatenate([a[:2][::-1], a, a[-2:][::-1]])
    return np.convolve(pad, k, mode="valid")


def ktable_factors(rng, nbin, ny, ktemp, kpress, gauss_y):
    """the two factors of the synthetic k-distribution table: kxy[x][y] (flat, y fastest) and ftp[t][p] (flat, p fastest) with
    kappa[t][p][x][y] = kxy[x][y] * ftp[t][p] -- what the device needs to form the table itself
    (RTBatch.set_species_separable); same random draws as ktable()"""
    a = _smooth5(rng.uniform(-6.0, -1.0, nbin))
    b = rng.uniform(1.0, 4.0, nbin)
    # log10 kappa[y,x,p,t] = a(x) + 3.5 * y^b(x) + 0.3 (log10 p - 6) - 0.2 T/1000
    yx = a[None, :] + 3.5 * gauss_y[:, None] ** b[None, :]                      # [y, x]
    lp = 0.3 * (np.log10(kpress) - 6.0)                                         # [p]
    lt = -0.2 * (ktemp / 1000.0)                                                # [t]
    base = np.ascontiguousarray(10.0 ** yx.T)                                   # [x, y]
    ftp = np.array([[10.0 ** (lp[p] + lt[t]) for p in range(len(kpress))] for t in range(len(ktemp))])
    return base.reshape(-1), ftp.reshape(-1)


def ktable(rng, nbin, ny, ktemp, kpress, gauss_y):
    """monotone-in-y synthetic k-distribution table, flat in reference order."""
    base, ftp = ktable_factors(rng, nbin, ny, ktemp, kpress, gauss_y)
    out = np.empty((len(ktemp) * len(kpress), nbin * ny), dtype=np.float64)     # [t][p][x][y]
    for tp in range(out.shape[0]):
        out[tp] = base * ftp[tp]
    return out.reshape(-1)


def rayleigh_table(wave, ntemp, npress):
    sig = 1e-27 * (1e-4 / wave) ** 4
    return np.ascontiguousarray(np.broadcast_to(sig, (ntemp, npress, len(wave)))).reshape(-1)


def meanmass_table(ntemp, npress, mu=2.3):
    return np.full(ntemp * npress, mu * pc.AMU)


def pressure_levels(p_boa, p_toa, nlayer):
    """source/host_functions.py:714-724"""
    lev = [p_boa * (p_toa / p_boa) ** (i / (2 * nlayer - 1)) for i in range(2 * nlayer)]
    p_lay = [lev[i] for i in range(1, 2 * nlayer, 2)]
    p_int = [lev[i] for i in range(0, 2 * nlayer, 2)]
    p_int.append(p_toa * (p_toa / p_boa) ** (1 / (2 * nlayer - 1)))
    return np.array(p_lay), np.array(p_int)


def cloud_arrays(nbin, nlayer, wave, p_lay, p_int, rng, decks=((1e5, 1.0), (1e3, 0.4)),
                 sigma0=1e-27, ratio=0.5):
    """two synthetic cloud decks (config 5): cross sections per gas particle ~ f_i sigma0 (lambda/1um)^-1
    with a vertical profile that falls off above the deck base with a scale-height ratio
    (param.dat:83-85 semantics; source/clouds.py:122-176 builds the real ones)."""
    spec = sigma0 * (wave / 1e-4) ** -1.0

    def profile(p):
        f = np.zeros_like(p)
        for p_base, amp in decks:
            above = p <= p_base
            f = f + np.where(above, amp * (p / p_base) ** (1.0 / ratio), 0.0)
        return f

    f_lay, f_int = profile(p_lay), profile(p_int)
    out = {}
    for name, f, n in (("lay", f_lay, nlayer), ("int", f_int, nlayer + 1)):
        out["abs_cross_all_clouds_" + name] = (0.3 * f[:, None] * spec[None, :]).reshape(-1)
        out["scat_cross_all_clouds_" + name] = (0.7 * f[:, None] * spec[None, :]).reshape(-1)
        out["g_0_all_clouds_" + name] = np.full(n * nbin, 0.6)
    return out


def species_set(rng, nspecies, nbin, ny, ktemp, kpress, gauss_y):
    """`nspecies` absorbers (molar weight U(2,64), constant VMR log-uniform in [1e-8,1e-2]) plus H2/He
    filler as Rayleigh scatterers (config 3-5)."""
    out = []
    for s in range(nspecies):
        out.append(dict(name="SPEC%02d" % s, absorbing="yes", scattering="no",
                        weight=float(rng.uniform(2.0, 64.0)),
                        vmr=float(10.0 ** rng.uniform(-8.0, -2.0)),
                        opacity_pretab=ktable(rng, nbin, ny, ktemp, kpress, gauss_y)))
    return out
