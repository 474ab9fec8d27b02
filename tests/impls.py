"""Adapters that give every implementation the stage signatures of oracle/helios_oracle.h."""
import numpy as np


def port_impl():
    import oracle
    return oracle.port


class RefImpl(object):
    """the reference's own kernels (oracle/_ref) behind the oracle's signatures"""

    def __init__(self, backend=None):
        if backend is None:
            import oracle
            if oracle.ref is None:
                raise RuntimeError("oracle/_ref/libhelios_ref.so not built")
            backend = oracle.ref
        self.r = backend
        r = self.r
        # identical argument lists
        self.planck_table = r.plancktable
        self.corr_inc_energy = r.corr_inc_energy
        self.planck_interpol_layer = r.planck_interpol_layer
        self.planck_interpol_interface = r.planck_interpol_interface
        self.opac_interpol = r.opac_interpol
        self.meanmolmass_interpol = r.meanmolmass_interpol
        self.kappa_interpol = r.kappa_interpol
        self.cp_interpol = r.cp_interpol
        self.entropy_interpol = r.entropy_interpol
        self.phase_number_interpol = r.phase_number_interpol
        self.opac_species_interpol = r.opac_species_interpol
        self.add_to_mixed_opac = r.add_to_mixed_opac
        self.calc_h2o_scat = r.calc_h2o_scat
        self.add_to_mixed_scat = r.add_to_mixed_scat
        self.calc_total_g0 = r.calc_total_g_0_of_gas_and_clouds
        self.fdir_iso = r.fdir_iso
        self.fdir_noniso = r.fdir_noniso
        self.integrate_flux = r.integrate_flux_double
        self.integrate_optdepth_transmission_iso = r.integrate_optdepth_transmission_iso
        self.integrate_optdepth_transmission_noniso = r.integrate_optdepth_transmission_noniso
        self.calc_contr_func_iso = r.calc_contr_func_iso
        self.calc_contr_func_noniso = r.calc_contr_func_noniso
        self.calc_mean_opacities = r.calc_mean_opacities

    def temp_inter(self, T_lay, T_int, ninterface):
        self.r.temp_inter(T_lay, T_int, ninterface, 0)

    def calc_trans_iso(self, *a):
        self.r.calc_trans_iso(*(a[:-1] + (0, a[-1])))       # debug = 0 before i2s_transition

    def calc_trans_noniso(self, *a):
        self.r.calc_trans_noniso(*(a[:-1] + (0, a[-1])))

    def calc_delta_z(self, T_lay, p_int, mmm, dz, g, nlayer):
        self.r.calc_delta_z(T_lay, p_int, p_int, mmm, dz, g, nlayer)  # p_lay is unused there

    def fband_iso(self, *a):
        a = list(a)
        # oracle: ..., g_0_tot_lay, g_0, Rstar, ...      ref: ..., g_0_tot_lay, g_0, singlewalk, Rstar, ...
        a.insert(13, 0)
        a.insert(len(a) - 1, 0)                              # debug
        self.r.fband_iso(*a)

    def fband_noniso(self, *a):
        a = list(a)
        a.insert(28, 0)                                      # singlewalk after g_0
        a.insert(len(a) - 1, 0)                              # debug
        self.r.fband_noniso(*a)

    def fband_matrix_iso(self, *a):
        a = list(a)
        a.insert(21, 0)                                      # singlewalk after g_0
        a.insert(len(a) - 1, 0)                              # debug
        self.r.fband_matrix_iso(*a)

    def fband_matrix_noniso(self, *a):
        a = list(a)
        a.insert(37, 0)                                      # singlewalk after g_0
        a.insert(len(a) - 1, 0)                              # debug
        self.r.fband_matrix_noniso(*a)

    def rad_temp_iter(self, F_down_tot, F_up_tot, F_net, F_net_diff, T_lay, p_lay, p_int, abrt,
                      T_store, pref, F_add_lay, F_add_sum, F_smooth, F_smooth_sum, c_p, mmm,
                      itervalue, foreplay, g, nlayer, physical_tstep, local_limit, adapt, smooth,
                      dim, step, F_intern, no_atmo):
        tint = np.zeros(nlayer + 1)
        self.r.rad_temp_iter(F_down_tot, F_up_tot, F_net, F_net_diff, T_lay, p_lay, tint, p_int,
                             abrt, T_store, pref, F_add_lay, F_add_sum, F_smooth, F_smooth_sum, c_p,
                             mmm, itervalue, 0.0, foreplay, g, nlayer, physical_tstep, local_limit,
                             adapt, smooth, dim, step, F_intern, no_atmo)

    def conv_temp_iter(self, F_net, F_net_diff, T_lay, p_lay, p_int, T_store, pref, marked_red,
                       F_add_lay, F_smooth, F_smooth_sum, nlayer, itervalue, adapt, smooth, F_intern):
        z = np.zeros(nlayer + 1)
        self.r.conv_temp_iter(z, z.copy(), F_net, F_net_diff, T_lay, p_lay, p_int, T_store, pref,
                              marked_red, F_add_lay, F_smooth, F_smooth_sum, nlayer, itervalue,
                              adapt, smooth, F_intern)

    def integrate_beamflux(self, F_dir_tot, F_dir_band, deltalambda, nbin, ninterface):
        self.r.integrate_beamflux(F_dir_tot, F_dir_band, deltalambda, np.zeros(1), nbin, ninterface)


class HipBackend(object):
    """libhelios_hip.so's per-stage entry points (hx_<kernel>, reference argument order) called with
    numpy arrays: every array argument is uploaded, the kernel runs on the GPU, and every array is
    downloaded again into the caller's numpy array (so in/out semantics match the CPU checkers)."""

    def __init__(self, ctx=None):
        import ctypes
        from helios_amd import _lib
        from helios_amd.device import Context
        self.ctx = ctx or Context(0)
        self._l = _lib.lib()
        self._protos = _lib.prototypes()
        self._ct = ctypes

    def __getattr__(self, name):
        ct = self._ct
        hx_name = "hx_" + {"integrate_flux_double": "integrate_flux"}.get(name, name)
        if hx_name not in self._protos:
            raise AttributeError(name)
        fn = getattr(self._l, hx_name)
        _res, argtypes, argnames = self._protos[hx_name]

        def call(*args):
            if len(args) != len(argtypes) - 1:
                raise TypeError("%s expects %d arguments, got %d" % (hx_name, len(argtypes) - 1, len(args)))
            conv, arrays = [self.ctx.handle], []
            for a, t in zip(args, argtypes[1:]):
                if isinstance(a, np.ndarray):
                    d = self.ctx.to_gpu(a)
                    arrays.append((a, d))
                    conv.append(ct.cast(d.ptr, t))
                elif a is None:
                    conv.append(None)
                else:
                    conv.append(a)
            self.ctx.check(fn(*conv), hx_name)
            for a, d in arrays:
                if a.flags.writeable:
                    a[...] = d.get()
                d.free()

        call.__name__ = hx_name
        return call


def hip_impl(ctx=None):
    return RefImpl(HipBackend(ctx))
