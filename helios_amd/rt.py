"""Python face of the fused fast path (include/helios_hip.h section 4, helios_amd/csrc/rt_fused.hip).

`RTBatch` owns the device-resident state of a batch of independent atmosphere columns and exposes
the two calls the iteration loop needs -- `step(iter)` (refreshing opacities every 10th iteration
exactly as source/computation.py:860 does) and `get(name)` (arrays in the REFERENCE's layouts).
`helios_amd.computation.Compute.radiation_loop` drives one RTBatch per `Store`.
"""
import ctypes

import numpy as np

from . import _lib


class RtDims(ctypes.Structure):
    _fields_ = [("nbin", ctypes.c_int32), ("ny", ctypes.c_int32), ("nlayer", ctypes.c_int32),
                ("ncol", ctypes.c_int32), ("ntemp", ctypes.c_int32), ("npress", ctypes.c_int32),
                ("plancktable_dim", ctypes.c_int32), ("plancktable_step", ctypes.c_int32),
                ("nspecies", ctypes.c_int32), ("reserved", ctypes.c_int32 * 7)]


class RtFlags(ctypes.Structure):
    _fields_ = [("scat", ctypes.c_int32), ("dir_beam", ctypes.c_int32), ("clouds", ctypes.c_int32),
                ("scat_corr", ctypes.c_int32), ("geom_zenith_corr", ctypes.c_int32),
                ("smooth", ctypes.c_int32), ("real_star", ctypes.c_int32),
                ("planet_type_gas", ctypes.c_int32), ("kcoeff_mixing_ro", ctypes.c_int32),
                ("debug", ctypes.c_int32), ("iso", ctypes.c_int32), ("singlewalk", ctypes.c_int32),
                ("matrix", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3),
                ("epsi", ctypes.c_double), ("epsi2", ctypes.c_double), ("g_0", ctypes.c_double),
                ("i2s_transition", ctypes.c_double), ("w_0_limit", ctypes.c_double),
                ("w_0_scat_limit", ctypes.c_double), ("delta_tau_limit", ctypes.c_double),
                ("reserved_d", ctypes.c_double * 9)]


class RtColumn(ctypes.Structure):
    _fields_ = [("g", ctypes.c_double), ("a", ctypes.c_double), ("R_planet", ctypes.c_double),
                ("R_star", ctypes.c_double), ("T_star", ctypes.c_double), ("f_factor", ctypes.c_double),
                ("mu_star", ctypes.c_double), ("F_intern", ctypes.c_double),
                ("rad_convergence_limit", ctypes.c_double), ("physical_tstep", ctypes.c_double),
                ("adapt_interval", ctypes.c_int32), ("foreplay", ctypes.c_int32),
                ("no_atmo", ctypes.c_int32), ("reserved_i", ctypes.c_int32),
                ("reserved_d", ctypes.c_double * 4)]


def _check_struct_sizes(l):
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    l.hx_rt_struct_sizes(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    got = (ctypes.sizeof(RtDims), ctypes.sizeof(RtFlags), ctypes.sizeof(RtColumn))
    if got != (a.value, b.value, c.value):
        raise _lib.HeliosHipError("hx_rt_* struct layout mismatch: python %r, library %r"
                                  % (got, (a.value, b.value, c.value)))


def _dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class RTBatch(object):
    """device-resident radiative-transfer state of `ncol` columns sharing grids and opacity tables"""

    def __init__(self, ctx, nbin, ny, nlayer, ncol, ntemp, npress, plancktable_dim, plancktable_step,
                 flags, columns, nspecies=0):
        self.ctx = ctx
        self._l = _lib.lib()
        _check_struct_sizes(self._l)
        self.nbin, self.ny, self.nlayer, self.ncol = int(nbin), int(ny), int(nlayer), int(ncol)
        self.ninterface = self.nlayer + 1
        self.nspecies = int(nspecies)
        self.dims = RtDims(nbin=nbin, ny=ny, nlayer=nlayer, ncol=ncol, ntemp=ntemp, npress=npress,
                           plancktable_dim=plancktable_dim, plancktable_step=plancktable_step,
                           nspecies=nspecies)
        self.flags = RtFlags()
        for k, v in flags.items():
            setattr(self.flags, k, v)
        cols = (RtColumn * ncol)()
        for i, c in enumerate(columns):
            for k, v in c.items():
                setattr(cols[i], k, v)
        self._cols = cols
        h = ctypes.c_void_p()
        ctx.check(self._l.hx_rt_create(ctx.handle, ctypes.byref(self.dims), ctypes.byref(self.flags), cols,
                                       ctypes.byref(h)), "hx_rt_create")
        self.handle = h

    def _ck(self, rc, what):
        self.ctx.check(rc, what)

    # ---- inputs ----------------------------------------------------------------------------------
    def set_grid(self, opac_interwave, opac_deltawave, opac_wave, gauss_y, gauss_weight, ktemp, kpress):
        arrs = [_f64(a) for a in (opac_interwave, opac_deltawave, opac_wave, gauss_y, gauss_weight, ktemp, kpress)]
        self._ck(self._l.hx_rt_set_grid(self.handle, *[_dp(a) for a in arrs]), "hx_rt_set_grid")

    def set_premixed_tables(self, opac_k, opac_scat_cross, opac_meanmass):
        arrs = [_f64(a) for a in (opac_k, opac_scat_cross, opac_meanmass)]
        self._ck(self._l.hx_rt_set_premixed_tables(self.handle, *[_dp(a) for a in arrs]),
                 "hx_rt_set_premixed_tables")

    def set_species(self, s, opacity_pretab, scat_cross, weight, is_h2o=0, is_cia=0, in_mu=1):
        p, q = _f64(opacity_pretab), _f64(scat_cross)
        self._ck(self._l.hx_rt_set_species(self.handle, s, _dp(p), _dp(q), float(weight), int(is_h2o),
                                           int(is_cia), int(in_mu)), "hx_rt_set_species")

    def set_species_separable(self, s, kxy, ftp, scat_cross, weight, is_h2o=0, is_cia=0, in_mu=1):
        """a synthetic species table kappa[t][p][x][y] = kxy[x][y] * ftp[t][p] (synthetic.ktable_factors), formed on the device"""
        a, b, q = _f64(kxy), _f64(ftp), _f64(scat_cross)
        self._ck(self._l.hx_rt_set_species_separable(self.handle, s, _dp(a), _dp(b), _dp(q), float(weight), int(is_h2o),
                                                     int(is_cia), int(in_mu)), "hx_rt_set_species_separable")

    def set_premixed_separable(self, kxy, ftp, opac_scat_cross, opac_meanmass):
        arrs = [_f64(a) for a in (kxy, ftp, opac_scat_cross, opac_meanmass)]
        self._ck(self._l.hx_rt_set_premixed_separable(self.handle, *[_dp(a) for a in arrs]), "hx_rt_set_premixed_separable")

    def set_species_vmr_table(self, s, vmr_pretab):
        """the species' mixing ratio on the opacity tables' (T, P) grid, [p + npress * t]: its profile follows the
        temperatures on the device at every refresh (None: back to the profiles of set_column_vmr)"""
        t = _f64(vmr_pretab)
        self._ck(self._l.hx_rt_set_species_vmr_table(self.handle, s, _dp(t)), "hx_rt_set_species_vmr_table")

    def set_column_vmr_table(self, col, s, vmr_pretab):
        """the same for ONE column of the batch (col < 0: all): columns of a sweep may come with their own chemistry"""
        t = _f64(vmr_pretab)
        self._ck(self._l.hx_rt_set_column_vmr_table(self.handle, int(col), s, _dp(t)), "hx_rt_set_column_vmr_table")

    def set_column_profile(self, col, p_lay, p_int, T_lay, surf_albedo=None, starflux=None):
        arrs = [_f64(a) for a in (p_lay, p_int, T_lay, surf_albedo, starflux)]
        self._ck(self._l.hx_rt_set_column_profile(self.handle, col, *[_dp(a) for a in arrs]),
                 "hx_rt_set_column_profile")

    def set_column_vmr(self, col, vmr_lay, vmr_int):
        a, b = _f64(vmr_lay), _f64(vmr_int)
        self._ck(self._l.hx_rt_set_column_vmr(self.handle, col, _dp(a), _dp(b)), "hx_rt_set_column_vmr")

    def set_column_clouds(self, col, abs_lay, abs_int, scat_lay, scat_int, g0_lay, g0_int):
        arrs = [_f64(a) for a in (abs_lay, abs_int, scat_lay, scat_int, g0_lay, g0_int)]
        self._ck(self._l.hx_rt_set_column_clouds(self.handle, col, *[_dp(a) for a in arrs]),
                 "hx_rt_set_column_clouds")

    def set_column_heating(self, col, F_add_heat_lay, F_add_heat_sum):
        a, b = _f64(F_add_heat_lay), _f64(F_add_heat_sum)
        self._ck(self._l.hx_rt_set_column_heating(self.handle, col, _dp(a), _dp(b)), "hx_rt_set_column_heating")

    def set_temperatures(self, col, T_lay):
        a = _f64(T_lay)
        self._ck(self._l.hx_rt_set_temperatures(self.handle, col, _dp(a)), "hx_rt_set_temperatures")

    def set_convergence_limit(self, col, limit):
        self._ck(self._l.hx_rt_set_convergence_limit(self.handle, col, float(limit)),
                 "hx_rt_set_convergence_limit")

    def set_state(self, col, name, array):
        a = np.ascontiguousarray(array)
        self._ck(self._l.hx_rt_set_state(self.handle, col, name.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                         a.nbytes), "hx_rt_set_state(%s)" % name)

    def keep_down_fluxes(self, on=True):
        """also keep the down-flux tiles so that F_down_wg / Fc_down_wg can be read back (tests)"""
        self.set_state(-1, "keep_down", np.array([1 if on else 0], np.int32))

    # ---- compute ---------------------------------------------------------------------------------
    def build_planck_table(self, energy_correction=1):
        self._ck(self._l.hx_rt_build_planck_table(self.handle, int(energy_correction)),
                 "hx_rt_build_planck_table")

    def refresh(self):
        self._ck(self._l.hx_rt_refresh(self.handle), "hx_rt_refresh")

    def step(self, itervalue, step_temperature=True):
        self._ck(self._l.hx_rt_step(self.handle, int(itervalue), 1 if step_temperature else 0), "hx_rt_step")

    def run(self, itervalue, nsteps):
        self._ck(self._l.hx_rt_run(self.handle, int(itervalue), int(nsteps)), "hx_rt_run")

    def set_kappa_table(self, entr_temp, entr_press, entr_kappa, entr_c_p):
        a = [_f64(v) for v in (entr_temp, entr_press, entr_kappa, entr_c_p)]
        self._ck(self._l.hx_rt_set_kappa_table(self.handle, _dp(a[0]), len(a[0]), _dp(a[1]), len(a[1]), _dp(a[2]),
                                               _dp(a[3])), "hx_rt_set_kappa_table")

    def kappa_cp_refresh(self):
        """kappa and c_p of every column from the table at the current temperatures"""
        self._ck(self._l.hx_rt_kappa_cp_refresh(self.handle), "hx_rt_kappa_cp_refresh")

    def conv_adjust(self, itervalue):
        self._ck(self._l.hx_rt_conv_adjust(self.handle, int(itervalue)), "hx_rt_conv_adjust")

    def conv_advance(self, itervalue):
        self._ck(self._l.hx_rt_conv_advance(self.handle, int(itervalue)), "hx_rt_conv_advance")

    def conv_run(self, itervalue, nsteps):
        self._ck(self._l.hx_rt_conv_run(self.handle, int(itervalue), int(nsteps)), "hx_rt_conv_run")

    def converged_layers(self):
        out = np.zeros(self.ncol, np.int32)
        self._ck(self._l.hx_rt_converged_layers(self.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
                 "hx_rt_converged_layers")
        return out

    # ---- outputs ---------------------------------------------------------------------------------
    _SHAPES = None

    def _shape(self, name):
        X, Y, L, I = self.nbin, self.ny, self.nlayer, self.ninterface
        wg = Y * X * I
        table = {
            "T_lay": (L + 1, np.float64), "T_int": (I, np.float64), "T_store": (L + 1, np.float64),
            "delta_t_prefactor": (L + 1, np.float64), "abort": (L + 1, np.int32),
            "F_up_band": (X * I, np.float64), "F_down_band": (X * I, np.float64),
            "F_dir_band": (X * I, np.float64), "F_up_tot": (I, np.float64), "F_down_tot": (I, np.float64),
            "F_net": (I, np.float64), "F_net_diff": (L, np.float64),
            "planckband_lay": (X * (L + 2), np.float64), "planckband_int": (X * I, np.float64),
            "opac_wg_lay": (Y * X * L, np.float64), "opac_wg_int": (wg, np.float64),
            "scat_cross_lay": (X * L, np.float64), "scat_cross_int": (X * I, np.float64),
            "meanmolmass_lay": (L, np.float64), "meanmolmass_int": (I, np.float64),
            "g_0_tot_lay": (X * L, np.float64), "g_0_tot_int": (X * I, np.float64),
            "delta_z_lay": (L, np.float64), "z_lay": (L, np.float64),
            "F_up_wg": (wg, np.float64), "F_down_wg": (wg, np.float64), "Fc_up_wg": (wg, np.float64),
            "Fc_down_wg": (wg, np.float64), "F_dir_wg": (wg, np.float64), "Fc_dir_wg": (wg, np.float64),
            "iters_done": (1, np.int32), "done": (1, np.int32), "flux_launch_policy": (2, np.float64), "graph_replays": (3, np.float64), "graph_builds": (2, np.float64),
            "conv_layer": (L + 1, np.int32), "conv_unstable": (L + 1, np.int32), "marked_red": (L + 1, np.int32),
            "kappa_lay": (L, np.float64), "kappa_int": (L + 1, np.float64), "c_p_lay": (L, np.float64),
            "F_smooth_sum": (L, np.float64), "F_add_heat_lay": (L, np.float64), "F_add_heat_sum": (L, np.float64),
            "planck_grid": ((self.dims.plancktable_dim + 1) * X, np.float64),
            "vmr_lay": (max(1, self.dims.nspecies) * I, np.float64), "vmr_int": (max(1, self.dims.nspecies) * I, np.float64),
        }
        return table[name]

    def get(self, name, col=0):
        n, dt = self._shape(name)
        out = np.zeros(n, dt)
        self._ck(self._l.hx_rt_get(self.handle, int(col), name.encode(), out.ctypes.data_as(ctypes.c_void_p),
                                   out.nbytes), "hx_rt_get(%s)" % name)
        return out

    def device_ptr(self, name, col=0):
        p = ctypes.c_void_p()
        self._ck(self._l.hx_rt_device_ptr(self.handle, int(col), name.encode(), ctypes.byref(p)),
                 "hx_rt_device_ptr(%s)" % name)
        return p

    def traffic_model(self):
        v = [ctypes.c_double() for _ in range(4)]
        self._ck(self._l.hx_rt_traffic_model(self.handle, *[ctypes.byref(x) for x in v]), "hx_rt_traffic_model")
        return dict(step_algorithmic=v[0].value, step_actual=v[1].value,
                    refresh_algorithmic=v[2].value, refresh_actual=v[3].value)

    def profile(self, enable=True):
        self._ck(self._l.hx_rt_profile(self.handle, 1 if enable else 0), "hx_rt_profile")

    def profile_read(self, kernel):
        ms, n = ctypes.c_double(), ctypes.c_int()
        self._ck(self._l.hx_rt_profile_read(self.handle, kernel.encode(), ctypes.byref(ms), ctypes.byref(n)),
                 "hx_rt_profile_read")
        return ms.value, n.value

    def close(self):
        if self.handle:
            self._l.hx_rt_destroy(self.handle)
            self.handle = None


def batch_from_case(ctx, c, ncol=1, nspecies=0, columns=None):
    """build an RTBatch from a dict-like problem description that uses the reference's Store attribute
    names (tests/cases.py, helios_amd.quantities.Store): every column gets the same inputs, except for the
    per-column parameters given in `columns` (a list of ncol dicts overriding g, a, F_intern, ...)."""
    flags = dict(scat=int(c.scat), dir_beam=int(c.dir_beam), clouds=int(c.clouds),
                 scat_corr=int(c.scat_corr), geom_zenith_corr=int(c.geom_zenith_corr),
                 smooth=int(c.smooth), real_star=int(c.real_star),
                 planet_type_gas=0 if c.get("planet_type", "gas") == "rocky" else 1,
                 kcoeff_mixing_ro=0 if c.get("kcoeff_mixing", "RO") == "correlated-k" else 1, iso=int(c.get("iso", 0)),
                 singlewalk=int(c.get("singlewalk", 0)), matrix=1 if c.get("flux_calc_method", "iteration") == "matrix" else 0,
                 epsi=float(c.epsi), epsi2=float(c.epsi2), g_0=float(c.g_0),
                 i2s_transition=float(c.i2s_transition), w_0_limit=float(c.w_0_limit),
                 w_0_scat_limit=float(c.w_0_scat_limit), delta_tau_limit=float(c.delta_tau_limit))
    col = dict(g=float(c.g), a=float(c.a), R_planet=float(c.R_planet), R_star=float(c.R_star),
               T_star=float(c.T_star), f_factor=float(c.f_factor), mu_star=float(c.mu_star),
               F_intern=float(c.F_intern), rad_convergence_limit=float(c.rad_convergence_limit),
               physical_tstep=float(c.physical_tstep), adapt_interval=int(c.adapt_interval),
               foreplay=int(c.foreplay), no_atmo=int(c.no_atmo))
    cols = [col] * ncol
    if columns is not None:
        cols = [dict(col, **{k: v for k, v in cp.items() if k in col}) for cp in columns]
    rt = RTBatch(ctx, c.nbin, c.ny, c.nlayer, ncol, c.ntemp, c.npress, c.plancktable_dim,
                 c.plancktable_step, flags, cols, nspecies=nspecies)
    try:
        rt.set_grid(c.opac_interwave, c.opac_deltawave, c.opac_wave, c.gauss_y, c.gauss_weight, c.ktemp, c.kpress)
        if nspecies == 0:
            if c.get("opac_k_factors") is not None:     # a synthetic table given by its two factors: formed on the device
                rt.set_premixed_separable(c.opac_k_factors[0], c.opac_k_factors[1], c.opac_scat_cross, c.opac_meanmass)
            else:
                rt.set_premixed_tables(c.opac_k, c.opac_scat_cross, c.opac_meanmass)
        rt.set_column_profile(-1, c.p_lay, c.p_int, c.T_lay, c.surf_albedo, c.starflux)
        if c.clouds:
            rt.set_column_clouds(-1, c.abs_cross_all_clouds_lay, c.abs_cross_all_clouds_int,
                                 c.scat_cross_all_clouds_lay, c.scat_cross_all_clouds_int,
                                 c.g_0_all_clouds_lay, c.g_0_all_clouds_int)
        rt.set_state(-1, "c_p_lay", np.asarray(c.c_p_lay, np.float64))
    except Exception:
        rt.close()         # a refused input must not leave the batch's device memory behind
        raise
    return rt
