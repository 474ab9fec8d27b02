"""Compute driver: one method per kernel group plus the two iteration loops.

Counterpart of the reference's `Compute` (source/computation.py).  Every per-stage method keeps its
reference name and calls the matching `hx_<kernel>` entry point of libhelios_hip.so with the `Store`'s
device arrays in the reference kernel's argument order -- no block/grid arguments, no synchronize after
each launch (calls are ordered on one stream).  `radiation_loop` / `convection_loop` keep the reference's
control flow but run the iteration body through the fused path (hx_rt_*) whenever the
configuration allows (non-isothermal layers, iterative flux solver -- the defaults), polling the
device-side convergence count instead of copying flags back every iteration.
"""
import ctypes

import os

import numpy as np

from . import _lib
from . import host_functions as hsfunc
from . import phys_const as pc
from .device import Context
from .rt import RTBatch


def _i(v):
    return int(v)


def _f(v):
    return float(v)


class Compute(object):
    """computational core: launches the HIP kernels of libhelios_hip.so"""

    def __init__(self, ctx=None):
        self._l = _lib.lib()          # raises HeliosHipError when the extension is missing
        # one process drives one GPU: HELIOS_DEVICE selects it (multi-GPU launchers set it from LOCAL_RANK)
        self.ctx = ctx or Context(int(os.environ.get("HELIOS_DEVICE", "0")))
        self.use_fused = True

    def _call(self, name, *args):
        self.ctx.check(getattr(self._l, name)(self.ctx.handle, *args), name)

    def _ctx_of(self, quant):
        if quant._ctx is None:
            quant._ctx = self.ctx
        return quant._ctx

    # ---- set-up ------------------------------------------------------------------------------------
    def construct_planck_table(self, quant):
        self._call("hx_plancktable", quant.dev_planckband_grid.d, quant.dev_opac_interwave.d,
                   quant.dev_opac_deltawave.d, _i(quant.nbin), _f(quant.T_star), _i(quant.plancktable_dim),
                   _i(quant.plancktable_step))

    def correct_incident_energy(self, quant):
        if quant.energy_correction == 1 and quant.T_star > 10:
            self._call("hx_corr_inc_energy", quant.dev_planckband_grid.d, quant.dev_starflux.d,
                       quant.dev_opac_deltawave.d, _i(quant.real_star), _i(quant.nbin), _f(quant.T_star),
                       _i(quant.plancktable_dim))
            self.print_energy_correction()

    def print_energy_correction(self):
        """the message the reference's kernel prints from the device (kernels.cu:453-457)"""
        corr = self.ctx.diag()["energy_correction"]
        if corr > 1:
            print("\nEnergy budget corrected (increased) by %.2f percent." % (100.0 * (corr - 1.0)))
        elif 0 < corr < 1:
            print("\nEnergy budget corrected (decreased) by %.2f percent." % (100.0 * (1.0 - corr)))

    def report_diagnostics(self, quant, always=False):
        """debugging feedback: what the reference's kernels print line by line when `debug = 1` (negative fluxes,
        kernels.cu:1458 ff.; limited G functions, :227) arrives here as counts since the last report; the
        malfunction message of the random-overlap re-binning (:3385) is reported whenever it occurred"""
        if not (always or _i(quant.debug or 0) == 1):
            return None
        d = self.ctx.diag()
        self.ctx.diag_reset()
        if d["negative_down_flux"] or d["negative_up_flux"]:
            print("WARNING WARNING WARNING WARNING -- negative flux found: %d downward, %d upward values !!!"
                  % (d["negative_down_flux"], d["negative_up_flux"]))
        if d["g_limited"]:
            print("WARNING: G_functions are being artificially limited!!! (%d values)" % d["g_limited"])
        if d["ro_rebin_skipped"]:
            print("ERROR ERROR ERROR: Rebinning algorithm in k-table RO method is malfunctioning. "
                  "Please double-check source code!!! (%d Gauss points)" % d["ro_rebin_skipped"])
        return d

    # ---- every iteration -----------------------------------------------------------------------------
    def interpolate_temperatures(self, quant):
        self._call("hx_temp_inter", quant.dev_T_lay.d, quant.dev_T_int.d, _i(quant.ninterface),
                   _i(quant.iter_value or 0))

    def interpolate_planck(self, quant):
        self._call("hx_planck_interpol_layer", quant.dev_T_lay.d, quant.dev_planckband_lay.d,
                   quant.dev_planckband_grid.d, quant.dev_starflux.d, _i(quant.real_star), _i(quant.nlayer),
                   _i(quant.nbin), _i(quant.plancktable_dim), _i(quant.plancktable_step))
        if quant.iso == 0:
            self._call("hx_planck_interpol_interface", quant.dev_T_int.d, quant.dev_planckband_int.d,
                       quant.dev_planckband_grid.d, _i(quant.ninterface), _i(quant.nbin),
                       _i(quant.plancktable_dim), _i(quant.plancktable_step))

    # ---- opacity refresh -----------------------------------------------------------------------------
    def interpolate_opacities_and_scattering_cross_sections(self, quant):
        for T, p, opac, scat, n in self._levels(quant, "dev_T_{}", "dev_p_{}", "dev_opac_wg_{}",
                                                 "dev_scat_cross_{}"):
            self._call("hx_opac_interpol", T.d, quant.dev_ktemp.d, p.d, quant.dev_kpress.d, quant.dev_opac_k.d,
                       opac.d, quant.dev_opac_scat_cross.d, scat.d, _i(quant.npress), _i(quant.ntemp),
                       _i(quant.ny), _i(quant.nbin), n)

    def _levels(self, quant, *patterns):
        """(arrays..., count) for layers and -- with non-isothermal layers -- interfaces"""
        out = [tuple(getattr(quant, p.format("lay")) for p in patterns) + (_i(quant.nlayer),)]
        if quant.iso == 0:
            out.append(tuple(getattr(quant, p.format("int")) for p in patterns) + (_i(quant.ninterface),))
        return out

    def interpolate_meanmolmass(self, quant):
        for T, p, mmm, n in self._levels(quant, "dev_T_{}", "dev_p_{}", "dev_meanmolmass_{}"):
            self._call("hx_meanmolmass_interpol", T.d, quant.dev_ktemp.d, mmm.d, quant.dev_opac_meanmass.d, p.d,
                       quant.dev_kpress.d, _i(quant.npress), _i(quant.ntemp), n)

    @staticmethod
    def _kappa_from_table(quant):
        """`kappa value = file` or `water_atmo`: the reference tests "is the value a string" (computation.py:202)"""
        return quant.convection == 1 and isinstance(quant.input_kappa_value, str) and len(quant.entr_kappa) > 1

    def interpolate_kappa_and_cp(self, quant):
        """only with a kappa/c_p table; a constant kappa needs no kernel (computation.py:199-250)"""
        if not self._kappa_from_table(quant):
            return
        for T, p, kap, n in self._levels(quant, "dev_T_{}", "dev_p_{}", "dev_kappa_{}"):
            self._call("hx_kappa_interpol", T.d, quant.dev_entr_temp.d, p.d, quant.dev_entr_press.d, kap.d,
                       quant.dev_entr_kappa.d, _i(quant.entr_npress), _i(quant.entr_ntemp), n)
        self._call("hx_cp_interpol", quant.dev_T_lay.d, quant.dev_entr_temp.d, quant.dev_p_lay.d,
                   quant.dev_entr_press.d, quant.dev_c_p_lay.d, quant.dev_entr_c_p.d, _i(quant.entr_npress),
                   _i(quant.entr_ntemp), _i(quant.nlayer))

    def interpolate_entropy(self, quant):
        """layer entropy from the kappa file (diagnostic column of one output file; computation.py:252-271)"""
        if self._kappa_from_table(quant):
            self._call("hx_entropy_interpol", quant.dev_T_lay.d, quant.dev_entr_temp.d, quant.dev_p_lay.d,
                       quant.dev_entr_press.d, quant.dev_entropy_lay.d, quant.dev_entr_entropy.d,
                       _i(quant.entr_npress), _i(quant.entr_ntemp), _i(quant.nlayer))

    def interpolate_phase_state(self, quant):
        """water phase number, `water_atmo` tables only (computation.py:273-292)"""
        if quant.input_kappa_value == "water_atmo":
            self._call("hx_phase_number_interpol", quant.dev_T_lay.d, quant.dev_entr_temp.d, quant.dev_p_lay.d,
                       quant.dev_entr_press.d, quant.dev_phase_number_lay.d, quant.dev_entr_phase_number.d,
                       _i(quant.entr_npress), _i(quant.entr_ntemp), _i(quant.nlayer))

    def calc_total_g_0_of_gas_and_clouds(self, quant):
        for scat, g_cl, s_cl, g_tot, n in self._levels(quant, "dev_scat_cross_{}", "dev_g_0_all_clouds_{}",
                                                       "dev_scat_cross_all_clouds_{}", "dev_g_0_tot_{}"):
            self._call("hx_calc_total_g_0_of_gas_and_clouds", scat.d, g_cl.d, s_cl.d, g_tot.d, _f(quant.g_0),
                       _i(quant.nbin), n)

    def calculate_transmission(self, quant):
        q = quant
        q.dev_scat_trigger.fill_zero()      # the reference re-uploads host zeros (computation.py:368)
        tail = (_f(q.g_0), _f(q.epsi), _f(q.epsi2), _f(q.mu_star), _f(q.w_0_limit), _f(q.w_0_scat_limit),
                _i(q.scat), _i(q.nbin), _i(q.ny), _i(q.nlayer), _i(q.clouds), _i(q.scat_corr), _i(q.debug or 0),
                _f(q.i2s_transition))
        if q.iso == 1:
            self._call("hx_calc_trans_iso", q.dev_trans_wg.d, q.dev_delta_tau_wg.d, q.dev_M_term.d,
                       q.dev_N_term.d, q.dev_P_term.d, q.dev_G_plus.d, q.dev_G_minus.d, q.dev_delta_colmass.d,
                       q.dev_opac_wg_lay.d, q.dev_meanmolmass_lay.d, q.dev_scat_cross_lay.d,
                       q.dev_abs_cross_all_clouds_lay.d, q.dev_scat_cross_all_clouds_lay.d,
                       q.dev_delta_tau_all_clouds.d, q.dev_w_0.d, q.dev_g_0_tot_lay.d, q.dev_scat_trigger.i, *tail)
        else:
            self._call("hx_calc_trans_noniso", q.dev_trans_wg_upper.d, q.dev_trans_wg_lower.d,
                       q.dev_delta_tau_wg_upper.d, q.dev_delta_tau_wg_lower.d, q.dev_M_upper.d, q.dev_M_lower.d,
                       q.dev_N_upper.d, q.dev_N_lower.d, q.dev_P_upper.d, q.dev_P_lower.d, q.dev_G_plus_upper.d,
                       q.dev_G_plus_lower.d, q.dev_G_minus_upper.d, q.dev_G_minus_lower.d,
                       q.dev_delta_col_upper.d, q.dev_delta_col_lower.d, q.dev_opac_wg_lay.d,
                       q.dev_opac_wg_int.d, q.dev_meanmolmass_lay.d, q.dev_meanmolmass_int.d,
                       q.dev_scat_cross_lay.d, q.dev_scat_cross_int.d, q.dev_abs_cross_all_clouds_lay.d,
                       q.dev_abs_cross_all_clouds_int.d, q.dev_scat_cross_all_clouds_lay.d,
                       q.dev_scat_cross_all_clouds_int.d, q.dev_delta_tau_all_clouds_upper.d,
                       q.dev_delta_tau_all_clouds_lower.d, q.dev_w_0_upper.d, q.dev_w_0_lower.d,
                       q.dev_g_0_tot_lay.d, q.dev_g_0_tot_int.d, q.dev_scat_trigger.i, *tail)

    def calculate_delta_z(self, quant):
        self._call("hx_calc_delta_z", quant.dev_T_lay.d, quant.dev_p_int.d, quant.dev_p_lay.d,
                   quant.dev_meanmolmass_lay.d, quant.dev_delta_z_lay.d, _f(quant.g), _i(quant.nlayer))

    def calculate_direct_beamflux(self, quant):
        q = quant
        tail = (q.dev_z_lay.d, _f(q.mu_star), _f(q.R_planet), _f(q.R_star), _f(q.a), _i(q.dir_beam),
                _i(q.geom_zenith_corr), _i(q.ninterface), _i(q.nbin), _i(q.ny))
        if q.iso == 1:
            self._call("hx_fdir_iso", q.dev_F_dir_wg.d, q.dev_planckband_lay.d, q.dev_delta_tau_wg.d, *tail)
        else:
            self._call("hx_fdir_noniso", q.dev_F_dir_wg.d, q.dev_Fc_dir_wg.d, q.dev_planckband_lay.d,
                       q.dev_delta_tau_wg_upper.d, q.dev_delta_tau_wg_lower.d, *tail)

    # ---- flux solve ----------------------------------------------------------------------------------
    def populate_spectral_flux_iteratively(self, quant):
        q = quant
        nscat_step = 3 if q.singlewalk == 0 else 1000          # computation.py:531-537
        for _ in range(nscat_step * _i(q.scat) + 1):
            if q.iso == 1:
                self._call("hx_fband_iso", q.dev_F_down_wg.d, q.dev_F_up_wg.d, q.dev_F_dir_wg.d,
                           q.dev_planckband_lay.d, q.dev_w_0.d, q.dev_M_term.d, q.dev_N_term.d, q.dev_P_term.d,
                           q.dev_G_plus.d, q.dev_G_minus.d, q.dev_surf_albedo.d, q.dev_g_0_tot_lay.d, _f(q.g_0),
                           _i(q.singlewalk), _f(q.R_star), _f(q.a), _i(q.ninterface), _i(q.nbin),
                           _f(q.f_factor), _f(q.mu_star), _i(q.ny), _f(q.epsi), _i(q.dir_beam), _i(q.clouds),
                           _i(q.scat_corr), _i(q.debug or 0), _f(q.i2s_transition))
            else:
                self._call("hx_fband_noniso", q.dev_F_down_wg.d, q.dev_F_up_wg.d, q.dev_Fc_down_wg.d,
                           q.dev_Fc_up_wg.d, q.dev_F_dir_wg.d, q.dev_Fc_dir_wg.d, q.dev_planckband_lay.d,
                           q.dev_planckband_int.d, q.dev_w_0_upper.d, q.dev_w_0_lower.d,
                           q.dev_delta_tau_wg_upper.d, q.dev_delta_tau_wg_lower.d,
                           q.dev_delta_tau_all_clouds_upper.d, q.dev_delta_tau_all_clouds_lower.d,
                           q.dev_M_upper.d, q.dev_M_lower.d, q.dev_N_upper.d, q.dev_N_lower.d, q.dev_P_upper.d,
                           q.dev_P_lower.d, q.dev_G_plus_upper.d, q.dev_G_plus_lower.d, q.dev_G_minus_upper.d,
                           q.dev_G_minus_lower.d, q.dev_surf_albedo.d, q.dev_g_0_tot_lay.d, q.dev_g_0_tot_int.d,
                           _f(q.g_0), _i(q.singlewalk), _f(q.R_star), _f(q.a), _i(q.ninterface), _i(q.nbin),
                           _f(q.f_factor), _f(q.mu_star), _i(q.ny), _f(q.epsi), _f(q.delta_tau_limit),
                           _i(q.dir_beam), _i(q.clouds), _i(q.scat_corr), _i(q.debug or 0), _f(q.i2s_transition))

    def solve_for_spectral_fluxes_via_matrix(self, quant):
        """source/computation.py:625-710: one tridiagonal solve per spectral point instead of the sweeps"""
        q = quant
        work = (q.dev_alpha.d, q.dev_beta.d, q.dev_source_term_down.d, q.dev_source_term_up.d,
                q.dev_c_prime.d, q.dev_d_prime.d, q.dev_scat_trigger.i)
        if q.iso == 1:
            self._call("hx_fband_matrix_iso", q.dev_F_down_wg.d, q.dev_F_up_wg.d, q.dev_F_dir_wg.d,
                       q.dev_planckband_lay.d, q.dev_w_0.d, q.dev_M_term.d, q.dev_N_term.d, q.dev_P_term.d,
                       q.dev_G_plus.d, q.dev_G_minus.d, q.dev_g_0_tot_lay.d, *work, q.dev_trans_wg.d,
                       q.dev_surf_albedo.d, _f(q.g_0), _i(q.singlewalk), _f(q.R_star), _f(q.a),
                       _i(q.ninterface), _i(q.nbin), _f(q.f_factor), _f(q.mu_star), _i(q.ny), _f(q.epsi),
                       _i(q.dir_beam), _i(q.clouds), _i(q.scat_corr), _i(q.debug or 0), _f(q.i2s_transition))
        else:
            self._call("hx_fband_matrix_noniso", q.dev_F_down_wg.d, q.dev_F_up_wg.d, q.dev_Fc_down_wg.d,
                       q.dev_Fc_up_wg.d, q.dev_F_dir_wg.d, q.dev_Fc_dir_wg.d, q.dev_planckband_lay.d,
                       q.dev_planckband_int.d, q.dev_w_0_upper.d, q.dev_w_0_lower.d,
                       q.dev_delta_tau_wg_upper.d, q.dev_delta_tau_wg_lower.d,
                       q.dev_delta_tau_all_clouds_upper.d, q.dev_delta_tau_all_clouds_lower.d,
                       q.dev_M_upper.d, q.dev_M_lower.d, q.dev_N_upper.d, q.dev_N_lower.d, q.dev_P_upper.d,
                       q.dev_P_lower.d, q.dev_G_plus_upper.d, q.dev_G_plus_lower.d, q.dev_G_minus_upper.d,
                       q.dev_G_minus_lower.d, q.dev_g_0_tot_lay.d, q.dev_g_0_tot_int.d, *work,
                       q.dev_trans_wg_upper.d, q.dev_trans_wg_lower.d, q.dev_surf_albedo.d, _f(q.g_0),
                       _i(q.singlewalk), _f(q.R_star), _f(q.a), _i(q.ninterface), _i(q.nbin), _f(q.f_factor),
                       _f(q.mu_star), _i(q.ny), _f(q.epsi), _f(q.delta_tau_limit), _i(q.dir_beam),
                       _i(q.clouds), _i(q.scat_corr), _i(q.debug or 0), _f(q.i2s_transition))

    def integrate_flux(self, quant):
        q = quant
        self._call("hx_integrate_flux", q.dev_opac_deltawave.d, q.dev_F_down_tot.d, q.dev_F_up_tot.d,
                   q.dev_F_net.d, q.dev_F_down_wg.d, q.dev_F_up_wg.d, q.dev_F_dir_wg.d, q.dev_F_down_band.d,
                   q.dev_F_up_band.d, q.dev_F_dir_band.d, q.dev_gauss_weight.d, _i(q.nbin), _i(q.ninterface),
                   _i(q.ny))

    # ---- temperature steps -----------------------------------------------------------------------------
    def rad_temp_iteration(self, quant):
        q = quant
        self._call("hx_rad_temp_iter", q.dev_F_down_tot.d, q.dev_F_up_tot.d, q.dev_F_net.d, q.dev_F_net_diff.d,
                   q.dev_T_lay.d, q.dev_p_lay.d, q.dev_T_int.d, q.dev_p_int.d, q.dev_abort.i, q.dev_T_store.d,
                   q.dev_delta_t_prefactor.d, q.dev_F_add_heat_lay.d, q.dev_F_add_heat_sum.d, q.dev_F_smooth.d,
                   q.dev_F_smooth_sum.d, q.dev_c_p_lay.d, q.dev_meanmolmass_lay.d, _i(q.iter_value),
                   _f(q.f_factor), _i(q.foreplay), _f(q.g), _i(q.nlayer), _f(q.physical_tstep),
                   _f(q.rad_convergence_limit), _i(q.adapt_interval), _i(q.smooth), _i(q.plancktable_dim),
                   _i(q.plancktable_step), _f(q.F_intern), _i(q.no_atmo_mode))

    def conv_temp_iteration(self, quant):
        q = quant
        self._call("hx_conv_temp_iter", q.dev_F_down_tot.d, q.dev_F_up_tot.d, q.dev_F_net.d, q.dev_F_net_diff.d,
                   q.dev_T_lay.d, q.dev_p_lay.d, q.dev_p_int.d, q.dev_T_store.d, q.dev_delta_t_prefactor.d,
                   q.dev_marked_red.i, q.dev_F_add_heat_lay.d, q.dev_F_smooth.d, q.dev_F_smooth_sum.d,
                   _i(q.nlayer), _i(q.iter_value), _i(q.adapt_interval), _i(q.smooth), _f(q.F_intern))

    # ---- on-the-fly opacity mixing -----------------------------------------------------------------------
    def interpolate_species_opac(self, quant):
        for T, p, spec, n in self._levels(quant, "dev_T_{}", "dev_p_{}", "dev_opac_spec_wg_{}"):
            self._call("hx_opac_species_interpol", T.d, quant.dev_ktemp.d, p.d, quant.dev_kpress.d,
                       quant.dev_opacity_spec_pretab.d, spec.d, _i(quant.npress), _i(quant.ntemp), _i(quant.ny),
                       _i(quant.nbin), n)

    def add_to_mixed_opacity(self, quant, mass_spec, s):
        mass = _f(mass_spec * pc.AMU)
        # CIA pairs and the correlated-k setting never use random overlap (computation.py:1343-1348)
        ro_method = 0 if (quant.kcoeff_mixing == "correlated-k" or "CIA" in quant.species_list[s].name) else 1
        for vmr, spec, mix, mmm, n in self._levels(quant, "dev_vmr_spec_{}", "dev_opac_spec_wg_{}",
                                                   "dev_opac_wg_{}", "dev_meanmolmass_{}"):
            self._call("hx_add_to_mixed_opac", vmr.d, spec.d, mix.d, mmm.d, quant.dev_gauss_weight.d,
                       quant.dev_gauss_y.d, mass, int(s), ro_method, _i(quant.ny), _i(quant.nbin), n)

    def calculate_H2O_Rayleigh_scattering(self, quant, s):
        mass = _f(quant.species_list[s].weight * pc.AMU)
        for T, p, sc, vmr, n in self._levels(quant, "dev_T_{}", "dev_p_{}", "dev_scat_cross_spec_{}",
                                             "dev_vmr_spec_{}"):
            self._call("hx_calc_h2o_scat", T.d, p.d, quant.dev_opac_wave.d, sc.d, vmr.d, mass, _i(quant.nbin), n)

    def add_to_mixed_scat_cross_sect(self, quant):
        for vmr, sc, tot, n in self._levels(quant, "dev_vmr_spec_{}", "dev_scat_cross_spec_{}",
                                            "dev_scat_cross_{}"):
            self._call("hx_add_to_mixed_scat", vmr.d, sc.d, tot.d, _i(quant.nbin), n)

    def calculate_total_opacity_and_scat_cross_sections_from_species(self, quant):
        """species loop of computation.py:1454-1501.  Species tables are uploaded ONCE and stay resident
        (`sp.dev_opacity_pretab`); the reference re-uploads each table on every refresh."""
        ctx = self._ctx_of(quant)
        for s, sp in enumerate(quant.species_list):
            quant.dev_vmr_spec_lay.set(sp.vmr_layer)
            if quant.iso == 0:
                quant.dev_vmr_spec_int.set(sp.vmr_interface)
            if sp.absorbing == "yes":
                if getattr(sp, "dev_opacity_pretab", None) is None:
                    sp.dev_opacity_pretab = ctx.to_gpu(sp.opacity_pretab)
                quant.dev_opacity_spec_pretab = sp.dev_opacity_pretab
                self.interpolate_species_opac(quant)
                self.add_to_mixed_opacity(quant, sp.weight, s)
            if sp.scattering == "yes":
                if sp.name == "H2O":
                    self.calculate_H2O_Rayleigh_scattering(quant, s)
                else:
                    quant.dev_scat_cross_spec_lay.set(sp.scat_cross_sect_layer)
                    if quant.iso == 0:
                        quant.dev_scat_cross_spec_int.set(sp.scat_cross_sect_interface)
                self.add_to_mixed_scat_cross_sect(quant)

    # ---- post-loop diagnostics ---------------------------------------------------------------------------
    def integrate_optdepth_transmission(self, quant):
        q = quant
        if q.iso == 1:
            self._call("hx_integrate_optdepth_transmission_iso", q.dev_trans_wg.d, q.dev_trans_band.d,
                       q.dev_delta_tau_wg.d, q.dev_delta_tau_band.d, q.dev_gauss_weight.d, _i(q.nbin),
                       _i(q.nlayer), _i(q.ny))
        else:
            self._call("hx_integrate_optdepth_transmission_noniso", q.dev_trans_wg_upper.d, q.dev_trans_wg_lower.d,
                       q.dev_trans_band.d, q.dev_delta_tau_wg_upper.d, q.dev_delta_tau_wg_lower.d,
                       q.dev_delta_tau_band.d, q.dev_gauss_weight.d, q.dev_delta_tau_all_clouds.d,
                       q.dev_delta_tau_all_clouds_upper.d, q.dev_delta_tau_all_clouds_lower.d, _i(q.nbin),
                       _i(q.nlayer), _i(q.ny))

    def calculate_contribution_function(self, quant):
        q = quant
        if q.iso == 1:
            self._call("hx_calc_contr_func_iso", q.dev_trans_wg.d, q.dev_trans_weight_band.d,
                       q.dev_contr_func_band.d, q.dev_gauss_weight.d, q.dev_planckband_lay.d, _f(q.epsi),
                       _i(q.nbin), _i(q.nlayer), _i(q.ny))
        else:
            self._call("hx_calc_contr_func_noniso", q.dev_trans_wg_upper.d, q.dev_trans_wg_lower.d,
                       q.dev_trans_weight_band.d, q.dev_contr_func_band.d, q.dev_gauss_weight.d,
                       q.dev_planckband_lay.d, _f(q.epsi), _i(q.nbin), _i(q.nlayer), _i(q.ny))

    def calculate_mean_opacities(self, quant):
        q = quant
        self._call("hx_calc_mean_opacities", q.dev_planck_opac_T_pl.d, q.dev_ross_opac_T_pl.d,
                   q.dev_planck_opac_T_star.d, q.dev_ross_opac_T_star.d, q.dev_opac_wg_lay.d,
                   q.dev_abs_cross_all_clouds_lay.d, q.dev_meanmolmass_lay.d, q.dev_planckband_lay.d,
                   q.dev_opac_interwave.d, q.dev_opac_deltawave.d, q.dev_T_lay.d, q.dev_gauss_weight.d,
                   q.dev_gauss_y.d, q.dev_opac_band_lay.d, _i(q.nlayer), _i(q.nbin), _i(q.ny), _f(q.T_star))

    def integrate_beamflux(self, quant):
        q = quant
        self._call("hx_integrate_beamflux", q.dev_F_dir_tot.d, q.dev_F_dir_band.d, q.dev_opac_deltawave.d,
                   q.dev_gauss_weight.d, _i(q.nbin), _i(q.ninterface))

    # =====================================================================================================
    # iteration loops
    # =====================================================================================================
    def _refresh_stagewise(self, quant):
        """the every-10th-iteration block of computation.py:860-879, one stage at a time"""
        if quant.opacity_mixing == "premixed":
            self.interpolate_opacities_and_scattering_cross_sections(quant)
            self.interpolate_meanmolmass(quant)
        else:
            hsfunc.calculate_vmr_for_all_species(quant)
            hsfunc.calculate_meanmolecularmass(quant)
            hsfunc.nullify_opac_scat_arrays(quant)
            self.calculate_total_opacity_and_scat_cross_sections_from_species(quant)
        if quant.clouds == 1:
            self.calc_total_g_0_of_gas_and_clouds(quant)
        self.calculate_transmission(quant)
        self.calculate_delta_z(quant)
        quant.delta_z_lay = quant.dev_delta_z_lay.get()
        quant.p_lay = quant.dev_p_lay.get()
        hsfunc.calculate_height_z(quant)
        quant.dev_z_lay.set(quant.z_lay)
        self.calculate_direct_beamflux(quant)

    @staticmethod
    def _stop_for_coupling_output(quant, it, nxt):
        """coupled runs may ask for the current T-P profile every n iterations (computation.py:967-971): end the chunk
        where the reference would write it"""
        n = int(getattr(quant, "coupl_tp_write_interval", 0) or 0)
        if quant.coupling == 1 and n > 0:
            due = it + 1 + (n - 1 - (it + 1) % n) % n       # first iteration count > it with count % n == n - 1
            nxt = min(nxt, due)
        return nxt

    @staticmethod
    def _coupling_output(quant, it, write, read):
        n = int(getattr(quant, "coupl_tp_write_interval", 0) or 0)
        if quant.coupling == 1 and n > 0 and it % n == n - 1 and write is not None:
            quant.T_lay = quant.rt.get("T_lay") if quant.rt is not None else quant.dev_T_lay.get()
            write.write_tp_for_coupling(quant, read)

    def _refresh_additional_heating(self, quant):
        """heating flux of the layers from the heating density and the current layer heights, every 10th iteration
        (computation.py:913-918, :1127-1132)"""
        if quant.add_heating == 1 and quant.iter_value % 10 == 0:
            quant.delta_z_lay = quant.dev_delta_z_lay.get()
            hsfunc.calc_add_heating_flux(quant)
            quant.dev_F_add_heat_lay.set(np.asarray(quant.F_add_heat_lay, np.float64))
            quant.dev_F_add_heat_sum.set(np.asarray(quant.F_add_heat_sum, np.float64))

    def _fused_supported(self, quant):
        # isothermal layers halve the segments per layer: 2048 of them fit the sweeps' tiles (32 rows x 64 lanes)
        # (`flux calculation method = matrix` runs in the same device-resident loop: hx_rt_flags.matrix; any number of
        # absorbers: the species loop on chip takes them in blocks of 48, csrc/rt_fused.hip refresh_species)
        why = None
        if quant.flux_calc_method not in ("iteration", "matrix"):
            why = "flux calculation method %r" % (quant.flux_calc_method,)
        elif _i(quant.nlayer) > (2048 if quant.iso == 1 else 1024):
            why = "%d layers (the sweeps' tiles hold 1024, 2048 isothermal ones)" % _i(quant.nlayer)
        if why is not None and self.use_fused and not getattr(self, "_told_stagewise", False):
            # not silent: the same kernels' per-stage entry points, every array through HBM at every stage -- measured
            # 5-10 times the device-resident loop's time per iteration
            self._told_stagewise = True
            print("\nhelios_amd: " + why + " -- outside the device-resident loop's limits; this run uses the per-stage "
                  "kernels (still on the GPU, same results, about 5-10 times slower per iteration).")
        return self.use_fused and why is None

    @staticmethod
    def _rt_flags(q):
        return dict(scat=_i(q.scat), dir_beam=_i(q.dir_beam), clouds=_i(q.clouds), scat_corr=_i(q.scat_corr),
                    geom_zenith_corr=_i(q.geom_zenith_corr), smooth=_i(q.smooth), real_star=_i(q.real_star),
                    planet_type_gas=1 if q.planet_type == "gas" else 0,
                    kcoeff_mixing_ro=0 if q.kcoeff_mixing == "correlated-k" else 1, iso=_i(q.iso),
                    singlewalk=_i(q.singlewalk), matrix=1 if q.flux_calc_method == "matrix" else 0, epsi=_f(q.epsi),
                    epsi2=_f(q.epsi2), g_0=_f(q.g_0), i2s_transition=_f(q.i2s_transition),
                    w_0_limit=_f(q.w_0_limit), w_0_scat_limit=_f(q.w_0_scat_limit),
                    delta_tau_limit=_f(q.delta_tau_limit), debug=_i(q.debug or 0))

    @staticmethod
    def _rt_column(q):
        return dict(g=_f(q.g), a=_f(q.a), R_planet=_f(q.R_planet), R_star=_f(q.R_star), T_star=_f(q.T_star),
                    f_factor=_f(q.f_factor), mu_star=_f(q.mu_star), F_intern=_f(q.F_intern),
                    rad_convergence_limit=_f(q.rad_convergence_limit), physical_tstep=_f(q.physical_tstep),
                    adapt_interval=_i(q.adapt_interval), foreplay=_i(q.foreplay), no_atmo=_i(q.no_atmo_mode))

    def _make_rt(self, quant):
        """device-resident fused state for this Store (one column)"""
        return self.make_rt_batch([quant])

    def make_rt_batch(self, quants):
        """device-resident fused state of several Stores that share wavelength grid, opacity tables and the physics
        switches and differ in their per-column parameters (planet, star, orbit, internal flux, albedo, clouds, start
        profile): column c of the batch belongs to quants[c] (`quant.rt`, `quant.rt_col`)"""
        q = quants[0]
        on_the_fly = q.opacity_mixing == "on-the-fly"
        nspecies = len(q.species_list) if on_the_fly else 0
        if on_the_fly:
            # before anything is allocated (the tables of a batch are gigabytes; a sweep goes on with its next batch after an
            # error): the columns of one batch share the species list and the kind of their mixing ratios
            for s, sp in enumerate(q.species_list):
                if getattr(sp, "source_for_vmr", "") != "FastChem":
                    continue
                for c, qc in enumerate(quants):
                    spc = qc.species_list[s] if s < len(qc.species_list) else None
                    if spc is None or getattr(spc, "source_for_vmr", "") != "FastChem" or getattr(spc, "vmr_pretab", None) is None:
                        raise ValueError("make_rt_batch: species %r is FastChem-tabulated in column 0 but not in column %d; "
                                         "columns of one batch share the species list and the kind of their mixing ratios"
                                         % (getattr(sp, "name", s), c))
        rt = RTBatch(self._ctx_of(q), _i(q.nbin), _i(q.ny), _i(q.nlayer), len(quants), _i(q.ntemp), _i(q.npress),
                     _i(q.plancktable_dim), _i(q.plancktable_step), self._rt_flags(q),
                     [self._rt_column(c) for c in quants], nspecies=nspecies)
        try:
            self._fill_rt_batch(rt, quants, on_the_fly)
        except Exception:
            rt.close()        # the device state of a batch that could not be set up is given back at once
            for qc in quants:
                if getattr(qc, "rt", None) is rt:
                    qc.rt, qc.rt_col = None, None
            raise
        return rt

    def _fill_rt_batch(self, rt, quants, on_the_fly):
        q = quants[0]
        rt.set_grid(q.opac_interwave, q.opac_deltawave, q.opac_wave, q.gauss_y, q.gauss_weight, q.ktemp, q.kpress)
        if on_the_fly:
            for s, sp in enumerate(q.species_list):
                scat = None
                if sp.scattering == "yes" and sp.name != "H2O":
                    scat = np.asarray(sp.scat_cross_sect_layer)[:_i(q.nbin)]
                rt.set_species(s, sp.opacity_pretab if sp.absorbing == "yes" else None, scat, sp.weight,
                               is_h2o=2 if (sp.scattering == "yes" and sp.name == "H2O") else 0,
                               is_cia=1 if "CIA" in sp.name else 0, in_mu=1 if hsfunc._counts_for_mu(sp) else 0)
                # calculate_vmr_for_all_species on the device: a FastChem species hands over its (T, P) table once and its
                # profile is interpolated from the device's temperatures at every refresh (host_functions.py:874-910).
                # One table per COLUMN: the Stores of a sweep each read their own FastChem directory (read.py:577-606)
                if getattr(sp, "source_for_vmr", "") == "FastChem":      # (checked for every column in make_rt_batch)
                    for c, qc in enumerate(quants):
                        rt.set_column_vmr_table(c, s, np.asarray(qc.species_list[s].vmr_pretab, np.float64).reshape(-1))
        else:
            rt.set_premixed_tables(q.opac_k, q.opac_scat_cross, q.opac_meanmass)
        if self._kappa_from_table(q):
            rt.set_kappa_table(q.entr_temp, q.entr_press, q.entr_kappa, q.entr_c_p)
        for c, qc in enumerate(quants):
            qc.rt, qc.rt_col = rt, c
            rt.set_column_profile(c, qc.p_lay, qc.p_int, qc.T_lay, qc.surf_albedo,
                                  qc.starflux if len(np.atleast_1d(qc.starflux)) == _i(qc.nbin) else None)
            if qc.clouds == 1:
                rt.set_column_clouds(c, qc.abs_cross_all_clouds_lay, qc.abs_cross_all_clouds_int,
                                     qc.scat_cross_all_clouds_lay, qc.scat_cross_all_clouds_int,
                                     qc.g_0_all_clouds_lay, qc.g_0_all_clouds_int)
            if qc.c_p_lay is not None and len(np.atleast_1d(qc.c_p_lay)) == _i(qc.nlayer):
                rt.set_state(c, "c_p_lay", np.asarray(qc.c_p_lay, np.float64))
            if qc.add_heating == 1:      # the flux follows the layer heights on the device (every refresh)
                rt.set_state(c, "add_heat_dens", np.asarray(qc.add_heat_dens, np.float64))
        rt.build_planck_table(1 if (q.energy_correction == 1 and q.T_star > 10) else 0)
        if _i(q.debug or 0) == 1:
            rt.keep_down_fluxes(True)     # so that the negative-flux count covers the downward fluxes too

    @staticmethod
    def _runtime_limit_iteration(quant):
        """number of completed iterations after which the reference's condition3 ends a time-stepped run: the smallest
        n with n * physical_tstep >= runtime_limit (computation.py:942 tests (iter + 1) * tstep < limit every iteration)"""
        ts, lim = float(quant.physical_tstep), float(quant.runtime_limit)
        n = max(1, int(np.ceil(lim / ts)))
        while n > 1 and (n - 1) * ts >= lim:
            n -= 1
        while n * ts < lim:
            n += 1
        return n

    def _push_vmr(self, quant):
        """hand the per-species VMR profiles to the fused state, once per loop: the profiles of species with a constant
        or file-given mixing ratio never change, those of FastChem species are re-interpolated on the device at every
        refresh (`RTBatch.set_species_vmr_table`) -- the values sent here for them are overwritten there"""
        hsfunc.calculate_vmr_for_all_species(quant)
        L, I = _i(quant.nlayer), _i(quant.ninterface)
        vl = np.array([np.asarray(sp.vmr_layer, np.float64)[:L] for sp in quant.species_list])
        vi = np.array([np.asarray(sp.vmr_interface, np.float64)[:I] for sp in quant.species_list])
        quant.rt.set_column_vmr(int(getattr(quant, "rt_col", 0)), vl, vi)

    def _pull_vmr(self, quant):
        """the mixing-ratio profiles of the last refresh, back onto the species (write.py prints them)"""
        rt, col = quant.rt, int(getattr(quant, "rt_col", 0))
        L, I, S = _i(quant.nlayer), _i(quant.ninterface), len(quant.species_list)
        vl, vi = rt.get("vmr_lay", col).reshape(S, I), rt.get("vmr_int", col).reshape(S, I)
        for s, sp in enumerate(quant.species_list):
            if getattr(sp, "source_for_vmr", "") == "FastChem":
                sp.vmr_layer = np.array(vl[s, :L], quant.fl_prec)
                if quant.iso == 0:
                    sp.vmr_interface = np.array(vi[s], quant.fl_prec)

    def radiation_loop(self, quant, write=None, read=None, rt_plot=None):
        """iterate to radiative equilibrium (reference computation.py:827-990)"""
        quant.iter_value = np.int32(0)
        if not self._fused_supported(quant):
            return self._radiation_loop_stagewise(quant, write, read, rt_plot)
        if quant.rt is None:
            quant.rt = self._make_rt(quant)
        rt = quant.rt
        L = _i(quant.nlayer)
        self.ctx.timer_start()
        if quant.singlewalk == 1:
            # post-processing run type (computation.py:983-984): one pass -- refresh, 1000*scat+1 sweeps inside one
            # launch of the flux kernel, quadrature -- and no temperature iteration
            if quant.opacity_mixing == "on-the-fly":
                self._push_vmr(quant)
            rt.step(0, step_temperature=False)
            self.report_diagnostics(quant)
            ms = self.ctx.timer_stop_ms()
            print("\nTime for radiative iteration [s]: {:.2f}".format(ms * 1e-3))
            print("Total number of iterative steps: " + str(quant.iter_value))
            self.sync_store_from_rt(quant)
            return
        it = 0
        condition1 = condition2 = condition3 = True
        while condition1 and condition2 and condition3:
            # work up to the next host-visible event: a refresh that needs host VMRs, a criterion
            # relaxation, the 100-iteration surface-temperature check, or the iteration limit
            if quant.opacity_mixing == "on-the-fly" and it == 0:
                self._push_vmr(quant)
            nxt = min(it + (10 - it % 10), _i(quant.max_nr_iterations) + 1)
            for r in quant.crit_relaxation_numbers:
                if it < r < nxt:
                    nxt = int(r)
            # the reference looks at the surface temperature inside every iteration whose index is a multiple of 100
            # (computation.py:946-952), i.e. after 1, 101, 201, ... completed iterations: end a chunk there
            nxt = min(nxt, it + 1 + (100 - it % 100) % 100)
            if quant.physical_tstep != 0:                        # computation.py:941-943, tested every iteration
                nxt = min(nxt, max(it + 1, self._runtime_limit_iteration(quant)))
            nxt = self._stop_for_coupling_output(quant, it, nxt)
            rt.run(it, nxt - it)
            self.report_diagnostics(quant)
            it_prev, it = it, nxt
            counts = rt.converged_layers()                       # blocks: one small D2H per <=10 iterations
            done = int(rt.get("done")[0])
            if quant.singlewalk == 0 and it_prev >= _i(quant.foreplay) and (it_prev % 100 == 0 or done):
                print("\nWe are running \"" + str(quant.name) + "\" at iteration step nr. : " + str(it_prev))
                print("Layers (& surface/BOA) converged: " + str(int(counts[0])) + " out of " + str(L + 1) + ".")
            if done:
                it = int(rt.get("iters_done")[0])                # the device froze the column exactly there
                condition1 = False
            if quant.physical_tstep != 0:
                condition3 = it * quant.physical_tstep < quant.runtime_limit
            if (it - 1) % 100 == 0:
                T_surf = rt.get("T_lay")[L]                      # computation.py:946-952
                condition2 = T_surf < quant.plancktable_dim * quant.plancktable_step - 2
                if not condition2 and quant.iso == 0:            # (isothermal layers cannot be adjusted: see convection_loop)
                    quant.convection = 1
            self._coupling_output(quant, it, write, read)
            if it in quant.crit_relaxation_numbers:
                hsfunc.relax_radiative_convergence_criterion(quant)
                rt.set_convergence_limit(0, quant.rad_convergence_limit)
            if it > quant.max_nr_iterations:
                if write is not None:
                    write.write_abort_file(quant, read)
                print("\nRun exceeds allowed maximum allowed number of iteration steps. Aborting...")
                raise SystemExit()
        quant.iter_value = np.int32(it)
        ms = self.ctx.timer_stop_ms()
        print("\nTime for radiative iteration [s]: {:.2f}".format(ms * 1e-3))
        print("Total number of iterative steps: " + str(quant.iter_value))
        self.sync_store_from_rt(quant)

    def sync_store_from_rt(self, quant, flux_state=False):
        """make the Store's dev_* arrays (reference layouts) reflect the fused state.  Large arrays go device to device;
        the persistent up-flux state (decoded from the tiles on the host) only when a per-stage loop is going to continue
        from it (`flux_state`)."""
        rt, col = quant.rt, int(getattr(quant, "rt_col", 0))
        X, Y, L, I = _i(quant.nbin), _i(quant.ny), _i(quant.nlayer), _i(quant.ninterface)
        for n in ("T_lay", "T_int", "F_up_band", "F_down_band", "F_dir_band", "F_up_tot", "F_down_tot", "F_net",
                  "F_net_diff", "planckband_lay", "planckband_int", "scat_cross_lay", "scat_cross_int",
                  "meanmolmass_lay", "meanmolmass_int", "delta_z_lay", "z_lay", "abort", "g_0_tot_lay",
                  "g_0_tot_int"):
            if quant.iso == 1 and n in ("planckband_int", "scat_cross_int", "meanmolmass_int", "g_0_tot_int"):
                continue       # the reference computes no interface values with isothermal layers: they stay zero
            v = rt.get(n, col)
            dev = getattr(quant, "dev_" + n, None)
            if dev is not None and dev.size == v.size:
                dev.set(v)
        # opacities of every spectral point: rebuilt on the device if the fused look-up skipped them, then copied d2d
        # (the Store over-allocates the layer arrays to ninterface slabs, Q2)
        for n, nlev in (("opac_wg_lay", L),) + ((("opac_wg_int", I),) if quant.iso == 0 else ()):
            getattr(quant, "dev_" + n).copy_from_device(rt.device_ptr(n, col), Y * X * nlev * 8)
        if quant.dir_beam == 1:
            for n in ("F_dir_wg",) + (("Fc_dir_wg",) if quant.iso == 0 else ()):
                getattr(quant, "dev_" + n).copy_from_device(rt.device_ptr(n, col), Y * X * I * 8)
        else:
            quant.dev_F_dir_wg.fill_zero()
            quant.dev_Fc_dir_wg.fill_zero()
        if flux_state:
            for n in ("F_up_wg",) + (("Fc_up_wg",) if quant.iso == 0 else ()):
                v = rt.get(n, col)
                dev = getattr(quant, "dev_" + n)
                buf = np.zeros(dev.size)
                buf[:v.size] = v
                dev.set(buf)
        if quant.add_heating == 1:
            for n in ("F_add_heat_lay", "F_add_heat_sum"):
                setattr(quant, n, rt.get(n, col))
                getattr(quant, "dev_" + n).set(getattr(quant, n))
        quant.dev_delta_t_prefactor.set(rt.get("delta_t_prefactor", col))
        quant.dev_T_store.set(rt.get("T_store", col))
        quant.dev_planckband_grid.copy_from_device(rt.device_ptr("planck_grid"), quant.dev_planckband_grid.nbytes)
        quant.T_lay = rt.get("T_lay", col)
        if quant.opacity_mixing == "on-the-fly":
            self._pull_vmr(quant)
        # transmission arrays for the post-loop diagnostics come from the per-stage kernel
        self.calculate_transmission(quant)

    def _radiation_loop_stagewise(self, quant, write=None, read=None, rt_plot=None):
        """the reference's loop, stage by stage (isothermal layers, post-processing run type, ...)"""
        L = _i(quant.nlayer)
        condition1 = condition2 = condition3 = True
        self.ctx.timer_start()
        while condition1 and condition2 and condition3:
            self.interpolate_temperatures(quant)
            self.interpolate_planck(quant)
            if quant.iter_value % 10 == 0:
                self._refresh_stagewise(quant)
            if quant.flux_calc_method == "iteration":
                self.populate_spectral_flux_iteratively(quant)
            elif quant.flux_calc_method == "matrix":
                self.solve_for_spectral_fluxes_via_matrix(quant)
            else:
                print("Flux calculation method unclear. Check parameter file for typos. Aborting...")
                raise SystemExit()
            self.integrate_flux(quant)
            self.report_diagnostics(quant)
            if quant.singlewalk == 0:
                abortsum = 0
                quant.marked_red = np.zeros(L + 1, np.int32)
                if quant.iter_value >= quant.foreplay:
                    self._refresh_additional_heating(quant)
                    if quant.physical_tstep != 0 and quant.iter_value % 10 == 0:
                        self.interpolate_kappa_and_cp(quant)
                    self.rad_temp_iteration(quant)
                    quant.abort = quant.dev_abort.get()
                    quant.marked_red[quant.abort == 0] = 1
                    abortsum = int(quant.abort.sum())
                    if quant.iter_value % 100 == 0:
                        print("\nWe are running \"" + str(quant.name) + "\" at iteration step nr. : " + str(quant.iter_value))
                        print("Layers (& surface/BOA) converged: " + str(abortsum) + " out of " + str(L + 1) + ".")
                condition1 = abortsum < L + 1
                if quant.physical_tstep != 0:
                    condition3 = (quant.iter_value + 1) * quant.physical_tstep < quant.runtime_limit
                if quant.iter_value % 100 == 0:
                    quant.T_lay = quant.dev_T_lay.get()
                    condition2 = quant.T_lay[L] < quant.plancktable_dim * quant.plancktable_step - 2
                    if not condition2 and quant.iso == 0:
                        quant.convection = 1
                quant.iter_value = np.int32(quant.iter_value + 1)
                self._coupling_output(quant, int(quant.iter_value), write, read)
                if quant.iter_value in quant.crit_relaxation_numbers:
                    hsfunc.relax_radiative_convergence_criterion(quant)
                if quant.iter_value > quant.max_nr_iterations:
                    if write is not None:
                        write.write_abort_file(quant, read)
                    print("\nRun exceeds allowed maximum allowed number of iteration steps. Aborting...")
                    raise SystemExit()
            else:
                condition1 = False
        ms = self.ctx.timer_stop_ms()
        print("\nTime for radiative iteration [s]: {:.2f}".format(ms * 1e-3))
        print("Total number of iterative steps: " + str(quant.iter_value))

    # ---- radiative-convective loop ---------------------------------------------------------------------------
    def _pull_for_convection(self, quant):
        for n in ("F_net", "F_up_tot", "F_down_tot", "F_net_diff", "T_lay", "meanmolmass_lay", "F_smooth_sum"):
            dev = getattr(quant, "dev_" + n)
            setattr(quant, n, dev.get())

    def _convection_loop_fused(self, quant, write=None, read=None):
        """the convection loop on the device-resident state: convective adjustment, sweeps, layer marking, equilibrium
        test and temperature step all run on the GPU (hx_rt_conv_*); the host only paces the loop in chunks that end at
        refresh boundaries, criterion relaxations and the iteration limit (reference computation.py:992-1174)"""
        rt = quant.rt
        L = _i(quant.nlayer)
        for n in ("T_lay", "F_net", "F_up_tot", "F_down_tot"):
            setattr(quant, n, rt.get(n))
        quant.p_lay, quant.p_int = np.asarray(quant.p_lay, float), np.asarray(quant.p_int, float)
        if self._kappa_from_table(quant):         # per-stage kernels on the Store's arrays (synced after the radiation loop)
            self.interpolate_kappa_and_cp(quant)
            quant.kappa_lay, quant.kappa_int = quant.dev_kappa_lay.get(), quant.dev_kappa_int.get()
            quant.c_p_lay = quant.dev_c_p_lay.get()
        hsfunc.conv_check(quant)
        hsfunc.mark_convective_layers(quant, stitching=0)
        condition = sum(quant.conv_unstable) > 0
        quant.iter_value = np.int32(0)
        if not condition:
            print("\nAll layers convectively stable. No convective adjustment necessary.\n")
            print("\nTime for rad.-conv. iteration [s]: {:.2f}".format(0.0))
            print("Total number of iterative steps: " + str(quant.iter_value))
            return
        print("\nConvectively unstable layers found. Starting convective adjustment")
        rt.set_state(0, "kappa_lay", np.asarray(quant.kappa_lay, np.float64))
        rt.set_state(0, "kappa_int", np.asarray(quant.kappa_int, np.float64))
        rt.set_state(0, "c_p_lay", np.asarray(quant.c_p_lay, np.float64))
        rt.set_state(0, "conv_layer", np.asarray(quant.conv_layer, np.int32))
        rt.set_state(0, "conv_unstable", np.asarray(quant.conv_unstable, np.int32))
        dampara = -1.0 if quant.input_dampara == "automatic" else float(quant.input_dampara)
        rt.set_state(0, "dampara", np.array([dampara], np.float64))
        rt.set_state(0, "done", np.zeros(1, np.int32))
        rt.set_convergence_limit(0, quant.rad_convergence_limit)
        self.ctx.timer_start()
        it, done = 0, 0
        while not done:
            if it % 100 == 0:
                print("\nWe are running \"" + str(quant.name) + "\" at iteration step nr. : " + str(it))
            nxt = min(it + (10 - it % 10), _i(quant.max_nr_iterations) + 1)
            for r in quant.crit_relaxation_numbers:
                if it < r < nxt:
                    nxt = int(r)
            nxt = self._stop_for_coupling_output(quant, it, nxt)
            # (the mixing ratios of FastChem species follow the profile on the device: before the adjustment for the mean
            # molecular mass, and for the adjusted profile in the refresh -- computation.py:1030-1036, :1056-1061)
            rt.conv_run(it, nxt - it)
            self.report_diagnostics(quant)
            it = nxt
            done = int(rt.get("done")[0])                        # one small D2H per <= 10 iterations
            if done:
                it = int(rt.get("iters_done")[0])
            self._coupling_output(quant, it, write, read)
            if it in quant.crit_relaxation_numbers:
                hsfunc.relax_radiative_convergence_criterion(quant)
                rt.set_convergence_limit(0, quant.rad_convergence_limit)
            if it > quant.max_nr_iterations:
                if write is not None:
                    write.write_abort_file(quant, read)
                print("\nRun exceeds allowed maximum allowed number of iteration steps. Aborting...")
                raise SystemExit()
        quant.iter_value = np.int32(it)
        ms = self.ctx.timer_stop_ms()
        quant.conv_layer = rt.get("conv_layer")
        quant.conv_unstable = rt.get("conv_unstable")
        quant.marked_red = rt.get("marked_red")
        quant.dev_F_smooth_sum.set(rt.get("F_smooth_sum"))
        for n in ("kappa_lay", "kappa_int", "c_p_lay"):
            setattr(quant, n, rt.get(n))
            getattr(quant, "dev_" + n).set(getattr(quant, n))
        print("\nTime for rad.-conv. iteration [s]: {:.2f}".format(ms * 1e-3))
        print("Total number of iterative steps: " + str(quant.iter_value))
        self.sync_store_from_rt(quant)

    def convection_loop(self, quant, write=None, read=None, rt_plot=None):
        """alternate convective adjustment and radiative steps, reference computation.py:992-1174: on the fused,
        device-resident state when the radiation loop ran there, else through the per-stage entry points with the
        adjustment on the host"""
        if not (quant.singlewalk == 0 and quant.convection == 1):
            return
        if quant.iso == 1:
            # the reference skips the stability test with isothermal layers and then sums a conv_unstable that was never
            # built (computation.py:1004-1009, quantities.py:134: `sum(None)` raises TypeError): say what is wrong instead
            raise IOError("ERROR: convective adjustment needs non-isothermal layers (the reference cannot run this "
                          "combination either); set 'isothermal layers = no' or 'convective adjustment = no'")
        if self._fused_supported(quant) and quant.rt is not None and quant.physical_tstep == 0:
            return self._convection_loop_fused(quant, write, read)
        if quant.rt is not None:      # the per-stage loop continues from the fused state
            self.sync_store_from_rt(quant, flux_state=True)
        L = _i(quant.nlayer)
        self.interpolate_kappa_and_cp(quant)
        quant.T_lay = quant.dev_T_lay.get()
        quant.p_lay = quant.dev_p_lay.get()
        quant.p_int = quant.dev_p_int.get()
        quant.kappa_lay = quant.dev_kappa_lay.get()
        if quant.iso == 0:
            quant.kappa_int = quant.dev_kappa_int.get()
            hsfunc.conv_check(quant)
            hsfunc.mark_convective_layers(quant, stitching=0)
        condition = sum(quant.conv_unstable) > 0
        quant.iter_value = np.int32(0)
        if condition:
            print("\nConvectively unstable layers found. Starting convective adjustment")
        else:
            print("\nAll layers convectively stable. No convective adjustment necessary.\n")
        self._pull_for_convection(quant)
        self.ctx.timer_start()
        while condition:
            self.interpolate_temperatures(quant)
            if quant.iter_value % 10 == 0:
                if quant.opacity_mixing == "premixed":
                    self.interpolate_meanmolmass(quant)
                else:
                    hsfunc.calculate_vmr_for_all_species(quant)
                    hsfunc.calculate_meanmolecularmass(quant)
            self.interpolate_kappa_and_cp(quant)
            quant.c_p_lay = quant.dev_c_p_lay.get()
            quant.meanmolmass_lay = quant.dev_meanmolmass_lay.get()
            quant.T_lay = quant.dev_T_lay.get()
            quant.F_smooth_sum = quant.dev_F_smooth_sum.get()
            hsfunc.convective_adjustment(quant)
            quant.dev_T_lay.set(quant.T_lay)
            self.interpolate_temperatures(quant)
            self.interpolate_planck(quant)
            if quant.iter_value % 10 == 0:
                self._refresh_stagewise(quant)
            if quant.flux_calc_method == "iteration":
                self.populate_spectral_flux_iteratively(quant)
            else:
                self.solve_for_spectral_fluxes_via_matrix(quant)
            self.integrate_flux(quant)
            self.report_diagnostics(quant)
            self._pull_for_convection(quant)
            hsfunc.mark_convective_layers(quant, stitching=1)
            if quant.physical_tstep != 0:
                break
            condition = (not hsfunc.check_for_radiative_eq(quant)) or (quant.iter_value < 400) or \
                        (sum(quant.conv_layer) == 0)
            if condition:
                self._refresh_additional_heating(quant)
                quant.dev_conv_layer.set(quant.conv_layer)
                quant.dev_marked_red.set(quant.marked_red)
                self.conv_temp_iteration(quant)
                quant.T_lay = quant.dev_T_lay.get()
                quant.iter_value = np.int32(quant.iter_value + 1)
            if quant.iter_value in quant.crit_relaxation_numbers:
                hsfunc.relax_radiative_convergence_criterion(quant)
            if quant.iter_value > quant.max_nr_iterations:
                if write is not None:
                    write.write_abort_file(quant, read)
                print("\nRun exceeds allowed maximum allowed number of iteration steps. Aborting...")
                raise SystemExit()
        ms = self.ctx.timer_stop_ms()
        print("\nTime for rad.-conv. iteration [s]: {:.2f}".format(ms * 1e-3))
        print("Total number of iterative steps: " + str(quant.iter_value))
