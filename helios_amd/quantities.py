"""State blackboard: host scalars/arrays plus their `dev_*` device counterparts.

Counterpart of the reference's `Store` (source/quantities.py:29-670).  The attribute names are the
de-facto ABI between the reader, the compute driver and the writers, so they are kept: scalars are
typed numpy scalars (`np.int32` / `np.float64`), host arrays are flat C-contiguous fp64/int32 arrays in
the reference's layouts, and every `dev_<name>` is a
`helios_amd.device.DeviceArray` (get() / set()) instead of a PyCUDA gpuarray.

Differences that stay invisible to callers: device arrays are allocated once and overwritten (the
reference re-assigns `dev_*` on every refresh and leaves the old buffers to the garbage collector,
source/computation.py:368, :878, :1461-1498); "layer" wg arrays are sized ny*nbin*ninterface exactly
as the reference over-allocates them (source/quantities.py:406-407).
"""
import numpy as np

from .device import Context

# single variables filled by the reader (source/quantities.py:36-134)
_SCALARS = """iso nlayer ninterface p_toa p_boa singlewalk scat diffusivity convection epsi epsi2 f_factor
T_intern ntemp npress entr_ntemp entr_npress g_0 planet g a R_planet R_star T_star T_eff_final model name
foreplay realtime_plot prec fl_prec nr_bytes iter_value ny nbin nlayer_nbin nlayer_plus2_nbin
ninterface_nbin nlayer_wg_nbin ninterface_wg_nbin wg_nbin nplanck_grid dir_beam dir_angle mu_star w_0_limit
w_0_scat_limit delta_tau_limit rad_convergence_limit global_limit n_plot energy_correction input_dampara
dampara F_intern adapt_interval smooth geom_zenith_corr scat_corr input_kappa_value approx_f planet_type
debug i2s_transition flux_calc_method clouds add_heating physical_tstep runtime_limit
force_start_tp_from_file plancktable_dim plancktable_step kcoeff_mixing opacity_mixing coupling
coupling_full_output coupling_speed_up coupling_iter_nr coupl_tp_write_interval coupl_convergence_limit
max_nr_iterations""".split()

# host input arrays copied to the device once (source/quantities.py:463-497)
_INPUT_ARRAYS = """p_lay p_int delta_colmass delta_col_upper delta_col_lower ktemp kpress entr_temp entr_press
opac_k gauss_y gauss_weight opac_wave opac_deltawave opac_interwave opac_scat_cross opac_meanmass entr_kappa
entr_c_p entr_phase_number entr_entropy c_p_lay kappa_lay starflux T_lay surf_albedo
abs_cross_all_clouds_lay scat_cross_all_clouds_lay g_0_all_clouds_lay""".split()
_INPUT_ARRAYS_NONISO = """abs_cross_all_clouds_int scat_cross_all_clouds_int g_0_all_clouds_int kappa_int""".split()


class Store(object):
    """stores parameters, quantities and arrays used by the radiative-transfer driver"""

    def __init__(self, ctx=None):
        self._ctx = ctx
        for n in _SCALARS:
            setattr(self, n, None)
        # defaults the reference sets in __init__
        self.real_star = np.int32(0)
        self.star_corr_factor = np.int32(1)   # never updated by the reference either
        self.tau_lw = 1
        self.F_sens = 0
        self.kappa_file_format = np.int32(0)
        self.relaxed_criterion_trigger = 0
        self.no_atmo_mode = np.int32(0)
        self.fl_prec = np.float64
        self.prec = "double"
        self.nr_bytes = 8
        # host-only lists/arrays
        self.T_restart = []
        self.conv_unstable = None
        self.F_net_conv = []
        self.F_ratio = []
        self.marked_red = None
        self.converged = None
        self.add_heat_dens = None
        self.species_list = []
        self.crit_relaxation_numbers = []
        self.conv_layer = None
        for n in _INPUT_ARRAYS + _INPUT_ARRAYS_NONISO:
            setattr(self, n, [])
            setattr(self, "dev_" + n, None)
        self.abort = None
        self.rt = None          # helios_amd.rt.RTBatch of the fused path, created by Compute

    # ---------------------------------------------------------------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context(0)   # raises when libhelios_hip.so / a GPU is missing: no CPU fallback
        return self._ctx

    def convert_input_list_to_array(self):
        """source/quantities.py:366-398"""
        for n in _INPUT_ARRAYS + _INPUT_ARRAYS_NONISO:
            v = getattr(self, n)
            if v is None:
                v = []
            setattr(self, n, np.ascontiguousarray(np.array(v, self.fl_prec).reshape(-1)))

    def dimensions(self):
        """source/quantities.py:400-409"""
        self.ninterface = np.int32(self.nlayer + 1)
        self.nlayer_nbin = np.int32(self.nlayer * self.nbin)
        self.nlayer_plus2_nbin = np.int32((self.nlayer + 2) * self.nbin)
        self.ninterface_nbin = np.int32(self.ninterface * self.nbin)
        self.ninterface_wg_nbin = np.int32(self.ninterface * self.ny * self.nbin)
        self.nlayer_wg_nbin = np.int32(self.ninterface * self.ny * self.nbin)
        self.wg_nbin = np.int32(self.ny * self.nbin)
        self.nplanck_grid = np.int32((self.plancktable_dim + 1) * self.nbin)

    # name -> (size attribute, dtype) of every zero-initialised array (source/quantities.py:411-461)
    def _zero_specs(self):
        f, i32 = self.fl_prec, np.int32
        nl, ni = int(self.nlayer), int(self.ninterface)
        lb, ib, wg = int(self.nlayer_nbin), int(self.ninterface_nbin), int(self.ninterface_wg_nbin)
        specs = {}
        for n in ("F_up_band", "F_down_band", "F_dir_band", "scat_cross_int", "planckband_int", "g_0_tot_int"):
            specs[n] = (ib, f)
        for n in ("F_up_wg", "F_down_wg", "F_dir_wg", "Fc_up_wg", "Fc_down_wg", "Fc_dir_wg", "opac_wg_lay",
                  "opac_wg_int"):
            specs[n] = (wg, f)
        for n in ("F_up_tot", "F_down_tot", "F_dir_tot", "F_net", "meanmolmass_int", "T_int"):
            specs[n] = (ni, f)
        for n in ("opac_band_lay", "scat_cross_lay", "trans_band", "delta_tau_band", "trans_weight_band",
                  "contr_func_band", "g_0_tot_lay", "delta_tau_all_clouds"):
            specs[n] = (lb, f)
        for n in ("F_net_diff", "meanmolmass_lay", "planck_opac_T_pl", "ross_opac_T_pl", "planck_opac_T_star",
                  "ross_opac_T_star", "entropy_lay", "phase_number_lay", "delta_z_lay", "z_lay",
                  "F_add_heat_lay", "F_add_heat_sum", "F_smooth", "F_smooth_sum"):
            specs[n] = (nl, f)
        specs["planckband_lay"] = (int(self.nlayer_plus2_nbin), f)
        specs["abort"] = (nl + 1, i32)
        specs["scat_trigger"] = (int(self.wg_nbin), i32)
        return specs

    def create_zero_arrays(self):
        for n, (size, dt) in self._zero_specs().items():
            setattr(self, n, np.zeros(size, dt))
        self.conv_layer = np.zeros(int(self.nlayer) + 1, np.int32)
        self.marked_red = np.zeros(int(self.nlayer) + 1, np.int32)

    def copy_host_to_device(self):
        """source/quantities.py:463-549"""
        ctx = self.ctx
        names = list(_INPUT_ARRAYS) + (list(_INPUT_ARRAYS_NONISO) if self.iso == 0 else [])
        for n in names:
            v = getattr(self, n)
            if v is None or len(v) == 0:
                v = np.zeros(1, self.fl_prec)
            setattr(self, "dev_" + n, ctx.to_gpu(np.asarray(v, self.fl_prec)))
        for n in self._zero_specs():
            setattr(self, "dev_" + n, ctx.to_gpu(getattr(self, n)))
        self.dev_conv_layer = ctx.to_gpu(self.conv_layer)
        self.dev_marked_red = ctx.to_gpu(self.marked_red)

    def allocate_on_device(self):
        """device-only work arrays (source/quantities.py:593-665); zero-filled here, whereas the
        reference leaves them uninitialised and relies on first-use writes"""
        ctx = self.ctx
        wg = int(self.nlayer_wg_nbin)
        lb = int(self.nlayer_nbin)
        self.dev_delta_t_prefactor = ctx.zeros(int(self.nlayer) + 1)
        self.dev_T_store = ctx.zeros(int(self.nlayer) + 1)
        self.dev_planckband_grid = ctx.zeros(int(self.nplanck_grid))
        for n in ("delta_tau_wg", "trans_wg", "w_0", "M_term", "N_term", "P_term", "G_plus", "G_minus"):
            setattr(self, "dev_" + n, ctx.zeros(wg))
        if self.opacity_mixing == "on-the-fly":
            self.dev_opac_spec_wg_lay = ctx.zeros(wg)
            self.dev_scat_cross_spec_lay = ctx.zeros(lb)
            self.dev_vmr_spec_lay = ctx.zeros(int(self.nlayer))
        if self.iso == 0:
            for stem in ("delta_tau_wg", "trans_wg", "M", "N", "P", "w_0", "G_plus", "G_minus"):
                for half in ("upper", "lower"):
                    setattr(self, "dev_%s_%s" % (stem, half), ctx.zeros(wg))
            self.dev_delta_tau_all_clouds_upper = ctx.zeros(lb)
            self.dev_delta_tau_all_clouds_lower = ctx.zeros(lb)
            if self.opacity_mixing == "on-the-fly":
                self.dev_opac_spec_wg_int = ctx.zeros(int(self.ninterface_wg_nbin))
                self.dev_scat_cross_spec_int = ctx.zeros(int(self.ninterface_nbin))
                self.dev_vmr_spec_int = ctx.zeros(int(self.ninterface))
        if self.flux_calc_method == "matrix":                      # source/quantities.py:652-665
            slabs = wg if self.iso == 1 else 2 * wg
            rows = int(self.wg_nbin) * (2 * int(self.ninterface) if self.iso == 1 else 4 * int(self.ninterface) - 2)
            for n in ("alpha", "beta", "source_term_down", "source_term_up"):
                setattr(self, "dev_" + n, ctx.zeros(slabs))
            self.dev_c_prime = ctx.zeros(rows)
            self.dev_d_prime = ctx.zeros(rows)

    def copy_device_to_host(self):
        """source/quantities.py:551-591"""
        for n in ("delta_colmass", "F_up_band", "F_down_band", "F_dir_band", "F_up_tot", "F_down_tot",
                  "F_dir_tot", "opac_band_lay", "scat_cross_lay", "F_net", "F_net_diff", "p_lay", "p_int",
                  "T_lay", "planckband_lay", "planck_opac_T_pl", "ross_opac_T_pl", "planck_opac_T_star",
                  "ross_opac_T_star", "trans_band", "delta_tau_band", "meanmolmass_lay", "c_p_lay",
                  "kappa_lay", "entropy_lay", "phase_number_lay", "trans_weight_band", "contr_func_band",
                  "g_0_tot_lay", "delta_z_lay", "z_lay", "delta_tau_all_clouds", "F_add_heat_sum",
                  "F_smooth_sum"):
            dev = getattr(self, "dev_" + n, None)
            if dev is not None:
                setattr(self, n, dev.get())
        if self.iso == 0:
            self.planckband_int = self.dev_planckband_int.get()
            if self.dev_kappa_int is not None:
                self.kappa_int = self.dev_kappa_int.get()
