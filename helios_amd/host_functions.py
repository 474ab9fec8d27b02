"""Host-side physics around and inside the iteration loops.

Counterpart of the reference's source/host_functions.py for the functions the hot path calls
(SURVEY.md section 2.1): grid construction, initial temperature, numerical limits, layer altitudes,
on-the-fly mixing ratios / mean molecular mass, convective adjustment, convergence checks.  Function
names and the `quant` attributes they read/write follow the reference; the bodies are this project's
own numpy code.  Everything here runs on the host on O(nlayer) vectors.
"""
import os

import numpy as np
from numpy.polynomial.legendre import leggauss

from . import phys_const as pc


# ---- planet / grid -------------------------------------------------------------------------------
def planet_param(quant, read=None):
    """cgs conversion of the planetary, stellar and orbital parameters (host_functions.py:33-48)"""
    if quant.planet != "manual" and read is not None:
        read.read_planet_database(quant)
    if quant.g < 10:                      # values < 10 are log10(g)
        quant.g = quant.fl_prec(10 ** quant.g)
    quant.a = quant.fl_prec(quant.a * pc.AU)
    quant.R_planet = quant.fl_prec(quant.R_planet * pc.R_JUP)
    quant.R_star = quant.fl_prec(quant.R_star * pc.R_SUN)
    quant.T_star = quant.fl_prec(max(quant.T_star, 2.7))


def approx_f_from_formula(quant, read):
    """heat-redistribution factor of a rocky planet from its longwave optical depth, Koll (2021) Eq. 10; tau_lw of a
    previous run of the same name is picked up if its output exists (host_functions.py:51-77)"""
    name = quant.name[:-5] if "_post" in quant.name else quant.name
    try:
        with open(read.output_path + name + "/" + name + "_tau_lw_tau_sw_f_factor.dat", "r") as f:
            for line in f.readlines()[2:]:
                quant.tau_lw = float(line.split()[0])
        print("\ntau_lw read in from previous output file!")
        print("\ntau_lw = ", quant.tau_lw)
    except IOError:
        print("\nWarning: Unable to read in tau_lw from file. Using either commandline values or starting from 1 per default.")
    T_eq = (quant.R_star / (2 * quant.a)) ** 0.5 * quant.T_star
    term = quant.tau_lw * (quant.p_boa / 1e6) ** (2 / 3) * (T_eq / 600) ** (-4 / 3)
    quant.f_factor = 2 / 3 - 5 / 12 * term / (2 + term)


def calc_planck(lamda, temp):
    return 2 * pc.H * pc.C ** 2 / lamda ** 5 / (np.exp(pc.H * pc.C / (lamda * pc.K_B * temp)) - 1)


def calc_tau_lw_sw(quant, read):
    """Planck-weighted longwave (surface temperature) and shortwave (stellar temperature) optical depths of the whole
    column, written next to the f factor for the next run of `approx_f_from_formula` (host_functions.py:92-162)"""
    X, L = int(quant.nbin), int(quant.nlayer)
    num_lw = denom_lw = num_sw = denom_sw = 0
    tau_top = [sum(quant.delta_tau_band[x + i * X] for i in range(L)) for x in range(X)]
    for x in range(X):
        B_surf = calc_planck(quant.opac_wave[x], quant.T_lay[L])
        num_lw += B_surf * np.exp(-tau_top[x]) * quant.opac_deltawave[x]
        denom_lw += B_surf * quant.opac_deltawave[x]
        if quant.T_star > 10:
            B_star = calc_planck(quant.opac_wave[x], quant.T_star)
            num_sw += B_star * np.exp(-tau_top[x]) * quant.opac_deltawave[x]
            denom_sw += B_star * quant.opac_deltawave[x]
    with np.errstate(divide="ignore"):
        tau_lw = -np.log(num_lw / denom_lw)
        tau_sw = -np.log(num_sw / denom_sw) if quant.T_star > 10 else 0
    if np.isinf(tau_lw):
        # transmission underflowed: fall back to the Planck mean of tau itself (the reference keeps adding to the
        # running sums of the first pass, host_functions.py:134-154)
        for x in range(X):
            B_surf = calc_planck(quant.opac_wave[x], quant.T_lay[L])
            num_lw += B_surf * tau_top[x] * quant.opac_deltawave[x]
            denom_lw += B_surf * quant.opac_deltawave[x]
            if quant.T_star > 10:
                B_star = calc_planck(quant.opac_wave[x], quant.T_star)
                num_sw += B_star * tau_top[x] * quant.opac_deltawave[x]
                denom_sw += B_star * quant.opac_deltawave[x]
        tau_lw = num_lw / denom_lw
        tau_sw = num_sw / denom_sw if quant.T_star > 10 else 0
    import os
    os.makedirs(read.output_path + quant.name, exist_ok=True)
    with open(read.output_path + quant.name + "/" + quant.name + "_tau_lw_tau_sw_f_factor.dat", "w") as f:
        f.write("This file contains the total longwave and shortwave optical depths at BOA (=surface), tau_lw and tau_sw, "
                "and the f factor as used in the model")
        f.write("\n{:<15}{:<15}{:<15}".format("tau_lw", "tau_sw", "f_factor"))
        f.write("\n{:<15g}{:<15g}{:<15g}".format(tau_lw, tau_sw, quant.f_factor))


def calculate_pressure_levels(quant):
    """2*nlayer log-spaced levels; even = interfaces, odd = layer centres (host_functions.py:714-724)"""
    n = int(quant.nlayer)
    ratio = quant.p_toa / quant.p_boa
    lev = [quant.p_boa * ratio ** (i / (2 * n - 1)) for i in range(2 * n)]
    p_layer = lev[1::2]
    p_interface = lev[0::2]
    p_interface.append(quant.p_toa * ratio ** (1 / (2 * n - 1)))
    return p_layer, p_interface


def construct_grid(quant):
    """pressure grid and column masses of layers / half-layers (host_functions.py:727-735)"""
    quant.p_lay, quant.p_int = calculate_pressure_levels(quant)
    p_lay, p_int = np.asarray(quant.p_lay), np.asarray(quant.p_int)
    quant.delta_colmass = list((p_int[:-1] - p_int[1:]) / quant.g)
    quant.delta_col_upper = list((p_lay - p_int[1:]) / quant.g)
    quant.delta_col_lower = list((p_int[:-1] - p_lay) / quant.g)


def effective_temperature(quant):
    geo = (quant.R_star / quant.a) ** 0.5 * quant.T_star
    return (1.0 - quant.dir_beam) * quant.f_factor ** 0.25 * geo + quant.dir_beam * abs(quant.mu_star) ** 0.25 * geo


def initial_temp(quant, read=None):
    """isothermal start at max(T_eff, 500 K), or a profile from file (host_functions.py:164-184)"""
    from_file = quant.singlewalk == 1 or (quant.force_start_tp_from_file == 1 and quant.physical_tstep != 0)
    if not from_file:
        T0 = max(effective_temperature(quant), 500)
        quant.T_lay = np.ones(int(quant.nlayer) + 1) * T0
        print("\nStarting with an isothermal TP-profile at {:g}".format(T0) + " K.")
    else:
        read.read_temperature_file(quant)
        # restart files list the surface first; T_lay keeps it last
        quant.T_lay = np.append(quant.T_restart[1:], quant.T_restart[0])
        print("\nStarting with chosen temperature profile.")


def calc_F_intern(quant):
    quant.F_intern = quant.fl_prec(pc.SIGMA_SB * quant.T_intern ** 4.0)


def set_up_numerical_parameters(quant):
    """hard-wired numerical limits and the Gauss weights (host_functions.py:209-222)"""
    quant.w_0_limit = quant.fl_prec(1.0 - 1e-10)
    quant.w_0_scat_limit = quant.fl_prec(1e-3)
    quant.delta_tau_limit = quant.fl_prec(1e-4)
    quant.gauss_weight = leggauss(int(quant.ny))[1] if int(quant.ny) > 1 else np.array([2.0])


def relax_radiative_convergence_criterion(quant):
    quant.rad_convergence_limit *= 10.0
    quant.relaxed_criterion_trigger = 1


# ---- altitudes -----------------------------------------------------------------------------------
def calculate_height_z(quant):
    """layer-centre altitudes from the layer thicknesses (host_functions.py:673-698)"""
    L = int(quant.nlayer)
    dz = quant.delta_z_lay
    z = quant.z_lay
    if quant.planet_type == "gas":
        i0 = max(i for i in range(L) if quant.p_lay[i] >= 1e7)   # z = 0 at the 10-bar level
        z[i0] = 0
        for i in range(i0 + 1, L):
            z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i]
        for i in range(i0 - 1, -1, -1):
            z[i] = z[i + 1] - 0.5 * dz[i + 1] - 0.5 * dz[i]
    else:
        z[0] = 0.5 * dz[0]
        for i in range(1, L):
            z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i]


def calc_add_heating_flux(quant):
    quant.F_add_heat_lay = quant.add_heat_dens * quant.delta_z_lay
    quant.F_add_heat_sum = np.cumsum(quant.F_add_heat_lay)


# ---- on-the-fly mixing: VMR profiles and mean molecular mass --------------------------------------
def interpolate_grid_to_lay_or_int(log_press, temp, table_2d, log_press_profile, temp_profile):
    """bilinear interpolation of a (T, log10 P) table along a profile, clamped to the table edges --
    what scipy's RectBivariateSpline(kx=1, ky=1) returns (host_functions.py:904-910)"""
    t = np.clip(np.asarray(temp_profile, float), temp[0], temp[-1])
    p = np.clip(np.asarray(log_press_profile, float), log_press[0], log_press[-1])
    it = np.clip(np.searchsorted(temp, t, side="right") - 1, 0, len(temp) - 2)
    ip = np.clip(np.searchsorted(log_press, p, side="right") - 1, 0, len(log_press) - 2)
    ft = (t - temp[it]) / (temp[it + 1] - temp[it])
    fp = (p - log_press[ip]) / (log_press[ip + 1] - log_press[ip])
    v = (table_2d[it, ip] * (1 - ft) * (1 - fp) + table_2d[it + 1, ip] * ft * (1 - fp)
         + table_2d[it, ip + 1] * (1 - ft) * fp + table_2d[it + 1, ip + 1] * ft * fp)
    return list(v)


def interpolate_vmr_to_opacity_grid(read, quant, vmr):
    """FastChem's (T, P) grid -> the opacity tables' (T, P) grid, bilinear in T and log10 P; beyond an edge of the
    FastChem grid the nearest row/column is used (host_functions.py:783-871).  Layouts: vmr[p + np_old * t] in,
    out[p + npress * t]."""
    t_old, p_old = np.asarray(read.fastchem_temp, float), np.asarray(read.fastchem_press, float)
    t_new, p_new = np.asarray(quant.ktemp, float), np.asarray(quant.kpress, float)
    old = np.asarray(vmr, float).reshape(len(t_old), len(p_old))

    def bracket(x_old, x_new):
        """index of the last old node <= x (0 and `clamped` when there is none or it is the last node)"""
        left = np.searchsorted(x_old, x_new, side="right") - 1
        clamped = (left < 0) | (left == len(x_old) - 1)
        return np.maximum(left, 0), clamped
    tl, t_cl = bracket(t_old, t_new)
    pl, p_cl = bracket(p_old, p_new)
    tr, pr = np.minimum(tl + 1, len(t_old) - 1), np.minimum(pl + 1, len(p_old) - 1)
    lp_old, lp_new = np.log10(p_old), np.log10(p_new)
    out = np.zeros((len(t_new), len(p_new)))
    for i in range(len(t_new)):
        for j in range(len(p_new)):
            a, b, c, d = tl[i], tr[i], pl[j], pr[j]
            if p_cl[j] and t_cl[i]:
                v = old[a, c]
            elif t_cl[i]:
                v = (old[a, d] * (lp_new[j] - lp_old[c]) + old[a, c] * (lp_old[d] - lp_new[j])) / (lp_old[d] - lp_old[c])
            elif p_cl[j]:
                v = (old[b, c] * (t_new[i] - t_old[a]) + old[a, c] * (t_old[b] - t_new[i])) / (t_old[b] - t_old[a])
            else:
                v = (old[b, d] * (t_new[i] - t_old[a]) * (lp_new[j] - lp_old[c])
                     + old[b, c] * (t_new[i] - t_old[a]) * (lp_old[d] - lp_new[j])
                     + old[a, d] * (t_old[b] - t_new[i]) * (lp_new[j] - lp_old[c])
                     + old[a, c] * (t_old[b] - t_new[i]) * (lp_old[d] - lp_new[j])) \
                    / ((t_old[b] - t_old[a]) * (lp_old[d] - lp_old[c]))
            if np.isnan(v):
                print("NaN-Error at entry with indices:", "pressure:", j, "temperature:", i)
                raise SystemExit()
            out[i, j] = v
    return out.reshape(-1)


def calculate_vmr_for_all_species(quant):
    """vertical VMR profiles of FastChem-tabulated species (host_functions.py:874-901)"""
    if quant.rt is not None:
        # interface temperatures of the CURRENT profile (the device's T_int array is one iteration old -- or not yet
        # written -- whenever the host looks): same arithmetic as temp_inter (kernels.cu:496-520)
        quant.T_lay = quant.rt.get("T_lay", int(getattr(quant, "rt_col", 0)))
        T, L = quant.T_lay, int(quant.nlayer)
        T_int = np.empty(L + 1)
        T_int[1:L] = T[:L - 1] + 0.5 * (T[1:L] - T[:L - 1])
        T_int[0] = T[0] - 0.5 * (T[1] - T[0])
        T_int[L] = T[L - 1] + 0.5 * (T[L - 1] - T[L - 2])
        quant.T_int = T_int
    else:
        quant.T_lay = quant.dev_T_lay.get()
        quant.T_int = quant.dev_T_int.get()
    log_p_lay, log_p_int = np.log10(quant.p_lay), np.log10(quant.p_int)
    log_kpress = np.log10(quant.kpress)
    L = int(quant.nlayer)
    for sp in quant.species_list:
        if sp.source_for_vmr == "FastChem":
            tab = np.asarray(sp.vmr_pretab).reshape((int(quant.ntemp), int(quant.npress)))
            sp.vmr_layer = np.array(interpolate_grid_to_lay_or_int(log_kpress, quant.ktemp, tab, log_p_lay,
                                                                   quant.T_lay[:L]), quant.fl_prec)
            if quant.iso == 0:
                sp.vmr_interface = np.array(interpolate_grid_to_lay_or_int(log_kpress, quant.ktemp, tab,
                                                                           log_p_int, quant.T_int), quant.fl_prec)


def _counts_for_mu(sp):
    return ("CIA" not in sp.name) and sp.name != "H-_ff" and sp.name != "He-"


def calc_meanmolmass(quant, type="layer"):
    """mu = AMU * sum(vmr * weight) / sum(vmr) over the real gas species (host_functions.py:927-959)"""
    n = int(quant.nlayer) if type == "layer" else int(quant.ninterface)
    num, tot = np.zeros(n), np.zeros(n)
    for sp in quant.species_list:
        if _counts_for_mu(sp):
            v = np.asarray(sp.vmr_layer if type == "layer" else sp.vmr_interface)
            num += v * sp.weight
            tot += v
    return np.array(num / tot * pc.AMU, quant.fl_prec)


def calculate_meanmolecularmass(quant):
    quant.meanmolmass_lay = calc_meanmolmass(quant, type="layer")
    quant.dev_meanmolmass_lay.set(quant.meanmolmass_lay)
    if quant.iso == 0:
        quant.meanmolmass_int = calc_meanmolmass(quant, type="interface")
        quant.dev_meanmolmass_int.set(quant.meanmolmass_int)


def nullify_opac_scat_arrays(quant):
    """zero the running mix (host_functions.py:1050-1056: uploads host zeros; here a device memset)"""
    quant.dev_opac_wg_lay.fill_zero()
    quant.dev_scat_cross_lay.fill_zero()
    if quant.iso == 0:
        quant.dev_opac_wg_int.fill_zero()
        quant.dev_scat_cross_int.fill_zero()


# ---- convergence / convection ----------------------------------------------------------------------
def check_for_radiative_eq(quant):
    """local radiative-equilibrium test of the non-convective layers (host_functions.py:251-286)"""
    L = int(quant.nlayer)
    quant.converged = np.zeros(L + 1, np.int32)
    quant.marked_red = np.zeros(L + 1, np.int32)
    norm = quant.F_down_tot[L] + quant.F_intern
    for i in range(L + 1):
        if quant.conv_layer[i] == 0:
            if i < L:
                dF = abs(quant.F_intern + quant.F_add_heat_sum[i] + quant.F_smooth_sum[i] - quant.F_net[i + 1])
            else:
                dF = abs(quant.F_intern - quant.F_net[0])
            if dF < quant.rad_convergence_limit * norm:
                quant.converged[i] = 1
            else:
                quant.marked_red[i] = 1
    if quant.iter_value % 100 == 1:
        print("Number of radiative layers converged: {:d} out of {:d}.".format(
            int(sum(quant.converged)), int((L + 1) - sum(quant.conv_layer))))
    return 1 if sum(quant.converged) == (L + 1) - sum(quant.conv_layer) else 0


def _adiabat_limit(quant, i, sign):
    """temperature layer i+1 would have on the dry adiabat through layer i, kappa scaled by (1 +- 1e-6)"""
    f = 1 + sign * 1e-6
    mid = quant.T_lay[i] * (quant.p_int[i + 1] / quant.p_lay[i]) ** (quant.kappa_lay[i] * f)
    return mid * (quant.p_lay[i + 1] / quant.p_int[i + 1]) ** (quant.kappa_int[i + 1] * f)


def conv_check(quant):
    """mark super-adiabatic neighbours (host_functions.py:337-365)"""
    L = int(quant.nlayer)
    quant.conv_unstable = np.zeros(L + 1, np.int32)
    for i in range(L - 1):
        if quant.p_lay[i] <= 1e1:
            break
        if quant.T_lay[i + 1] < _adiabat_limit(quant, i, +1):
            quant.conv_unstable[i] = 1
            quant.conv_unstable[i + 1] = 1
    T_ad = quant.T_lay[L] * (quant.p_lay[0] / quant.p_int[0]) ** (quant.kappa_int[0] * (1 + 1e-6))
    if quant.T_lay[0] < T_ad:
        quant.conv_unstable[L] = 1
        quant.conv_unstable[0] = 1


def mark_convective_layers(quant, stitching):
    """layers on (or steeper than) the adiabat count as convective (host_functions.py:545-582)"""
    L = int(quant.nlayer)
    quant.conv_layer[L] = 0
    quant.conv_layer[0] = 0
    for i in range(L - 1):
        if quant.p_lay[i] <= 1e1:
            break
        if quant.T_lay[i + 1] < _adiabat_limit(quant, i, -1):
            quant.conv_layer[i] = 1
            quant.conv_layer[i + 1] = 1
        else:
            quant.conv_layer[i + 1] = 0
    for i in range(L - 1):                       # no kinks at the top edge of a zone
        if quant.T_lay[i + 1] > quant.T_lay[i]:
            quant.conv_layer[i] = 0
    T_ad = quant.T_lay[L] * (quant.p_lay[0] / quant.p_int[0]) ** (quant.kappa_int[0] * (1 - 1e-6))
    if quant.T_lay[0] < T_ad:
        quant.conv_layer[L] = 1
        quant.conv_layer[0] = 1
    if stitching == 1 and quant.iter_value > 5000:
        stitching_convective_zone_holes(quant)


def _zones(flags, L):
    """contiguous runs of flagged layers; the surface 'ghost layer' (index L) is index -1 here"""
    idx = [i for i in range(L + 1) if flags[i]]
    if L in idx:
        idx = [-1] + idx[:-1]
    starts = [i for i in idx if i - 1 not in idx]
    ends = [i for i in idx if i + 1 not in idx]
    return starts, ends


def stitching_convective_zone_holes(quant):
    """close radiative gaps narrower than a scale height between zones (host_functions.py:585-635)"""
    L = int(quant.nlayer)
    starts, ends = [], []
    for i in range(L):
        if quant.conv_layer[i] == 1:
            below = quant.conv_layer[i - 1] if i > 0 else quant.conv_layer[L]
            if below == 0:
                starts.append(i)
            if i == L - 1 or quant.conv_layer[i + 1] == 0:
                ends.append(i)
    if quant.conv_layer[L] == 1:
        starts = [-1] + starts
        if quant.conv_layer[0] == 0:
            ends = [-1] + ends
    if len(starts) != len(ends):
        raise SystemExit("Error in stitching calculation. Aborting...")
    for n in range(len(starts) - 1):
        p_top = quant.p_lay[starts[n + 1]]
        p_bot = quant.p_lay[ends[n]] if ends[n] != -1 else quant.p_int[0]
        if p_top / p_bot > 1 / np.e:
            for m in range(ends[n] + 1, starts[n + 1]):
                quant.conv_layer[m] = 1


def conv_correct(quant, fudging):
    """put every unstable/convective block on the adiabat of its enthalpy-conserving mean potential
    temperature (host_functions.py:368-506)"""
    L = int(quant.nlayer)
    flags = [(quant.conv_unstable[i] == 1 or quant.conv_layer[i] == 1) for i in range(L + 1)]
    starts, ends = _zones(flags, L)
    if len(starts) != len(ends):
        raise SystemExit("Error in convective calculation. Aborting...")
    fudge = np.ones(len(starts))
    if fudging == 1:
        nz = len(starts)
        for n in range(nz):
            test = None
            for m in range(n, nz):
                if m != nz - 1:
                    p_top = quant.p_lay[starts[m + 1]]
                    p_bot = quant.p_lay[ends[m]] if ends[m] != -1 else quant.p_int[0]
                    if p_top / p_bot < 1 / np.e:      # a radiative zone thicker than one scale height
                        test = int((ends[m] + starts[m + 1]) / 2)
                        break
                else:
                    test = int(0.8 * ends[m] + 0.2 * (quant.ninterface - 1))
            if quant.input_dampara == "automatic":
                if quant.T_star > 10:
                    quant.dampara = 0.5 if n < nz - 1 else 4.0
                else:
                    quant.dampara = 8.0
            else:
                quant.dampara = float(quant.input_dampara)
            ratio = (quant.F_intern + quant.F_add_heat_sum[test - 1] + quant.F_smooth_sum[test - 1]
                     + quant.F_down_tot[test]) / quant.F_up_tot[test]
            fudge[n] = min(1.01, max(0.99, ratio ** (1.0 / quant.dampara)))
    cp_mu = np.asarray(quant.c_p_lay) / np.asarray(quant.meanmolmass_lay)
    dp = np.asarray(quant.p_int[:-1]) - np.asarray(quant.p_int[1:])
    # adiabatic factor of one whole layer j (centre -> centre above is built from these pieces)
    up_int = (np.asarray(quant.p_lay) / np.asarray(quant.p_int[:-1])) ** np.asarray(quant.kappa_int[:-1])
    int_to_next = (np.asarray(quant.p_int[1:]) / np.asarray(quant.p_lay)) ** np.asarray(quant.kappa_lay[:L])
    for n in range(len(starts)):
        a, b = max(0, starts[n]), max(0, ends[n])
        num, den = 0.0, 0.0
        chain = 1.0
        factors = []
        for i in range(a, b + 1):
            fac = chain * up_int[i]
            factors.append(fac)
            num += cp_mu[i] * quant.T_lay[i] * dp[i]
            den += fac * cp_mu[i] * dp[i]
            chain *= up_int[i] * int_to_next[i]
        theta = num / den * fudge[n]
        for i, fac in zip(range(a, b + 1), factors):
            quant.T_lay[i] = theta * fac
        if starts[n] == -1:
            quant.T_lay[L] = theta


def convective_adjustment(quant):
    """iterate check -> mark -> correct until stable, then once more with stitching and the flux
    'fudge factor' (host_functions.py:509-542)"""
    conv_check(quant)
    while sum(quant.conv_unstable) > 0:
        mark_convective_layers(quant, stitching=0)
        conv_correct(quant, fudging=0)
        conv_check(quant)
    mark_convective_layers(quant, stitching=1)
    conv_correct(quant, fudging=1)


def sum_mean_optdepth(quant, i, opac):
    """optical depth from the TOA down to layer i with a mean opacity; layers flagged -3 ("temperature too
    low for the table") are skipped, and -3 is returned when nothing could be summed (host_functions.py:321-334)"""
    opac = np.asarray(opac[i:int(quant.nlayer)], float)
    ok = opac != -3
    tau = float(np.sum(np.asarray(quant.delta_colmass[i:int(quant.nlayer)], float)[ok][::-1] * opac[ok][::-1]))
    return tau if tau > 0 else -3


def calculate_conv_flux(quant):
    """convective net flux at the interfaces (host_functions.py:638-651)"""
    L = int(quant.nlayer)
    quant.F_net_conv = np.zeros(int(quant.ninterface), quant.fl_prec)
    for i in range(1, int(quant.ninterface)):
        if quant.conv_layer[i - 1] == 1:
            quant.F_net_conv[i] = quant.F_intern + quant.F_add_heat_sum[i - 1] + quant.F_smooth_sum[i - 1] - quant.F_net[i]
    if quant.conv_layer[L] == 1:
        quant.F_net_conv[0] = quant.F_intern - quant.F_net[0]


def calc_F_ratio(quant):
    """planet-to-star flux ratio per bin (host_functions.py:654-670; star_corr_factor stays 1, Q18)"""
    quant.F_ratio = []
    if quant.T_star > 10:
        L, X = int(quant.nlayer), int(quant.nbin)
        orbital = (quant.R_planet / quant.R_star) ** 2
        for x in range(X):
            star = np.pi * quant.planckband_lay[L + x * (L + 2)] / quant.star_corr_factor
            quant.F_ratio.append(orbital * quant.F_up_band[x + L * X] / star if star != 0 else 0)


def temp_calcs(quant):
    geo = (quant.R_star / quant.a) ** 0.5 * quant.T_star
    T_model = effective_temperature(quant)
    T_star_b = (quant.F_down_tot[int(quant.ninterface) - 1] / pc.SIGMA_SB) ** 0.25
    T_planet_b = (quant.F_up_tot[int(quant.ninterface) - 1] / pc.SIGMA_SB) ** 0.25
    return 0.25 ** 0.25 * geo, 0.667 ** 0.25 * geo, T_model, T_star_b, T_planet_b


def global_energy_imbalance(quant):
    """(F_intern + heating + smoothing - F_net[TOA]) / (F_down_tot[TOA] + F_intern), host_functions.py:1040"""
    n = int(quant.ninterface)
    return (quant.F_intern + quant.F_add_heat_sum[n - 2] + quant.F_smooth_sum[n - 2] - quant.F_net[n - 1]) / \
           (quant.F_down_tot[n - 1] + quant.F_intern)


def calculate_coupling_convergence(quant, read):
    """coupled run converged when every temperature of this coupling step is within the criterion of the previous
    step's; the verdict (0/1) goes to `<name>_coupling_convergence.dat` for the outer script (host_functions.py:962-1018)"""
    from .write import Write
    verdict = 0
    step = int(quant.coupling_iter_nr)
    if step > 0 and quant.singlewalk == 0:
        previous = Write.read_coupling_tp(Write._coupling_tp_path(quant, read, step - 1, previous=True), quant.fl_prec)
        current = Write.read_coupling_tp(Write._coupling_tp_path(quant, read, step), quant.fl_prec)
        ok = [abs(previous[t] - current[t]) / current[t] < quant.coupl_convergence_limit for t in range(len(current))]
        verdict = 1 if all(ok) else 0
        name = str(quant.name)
        with open(os.path.join(read.output_path, name, name + "_coupling_convergence.dat"), "w") as f:
            f.write(str(verdict))
    return verdict


def success_message(quant):
    Tg, Td, Tm, Ts, Tp = temp_calcs(quant)
    print("\nDone! Everything appears to have worked fine :-)\n")
    print("This has been " + ("an iterative" if quant.singlewalk == 0 else "a post-processing")
          + " run with name " + str(quant.name) + ".\n")
    if quant.physical_tstep == 0:
        print("  --> Theoretical effective temperature of planet: global {:g} K, day-side {:g} K, model {:g} K".format(Tg, Td, Tm))
        print("  --> Incident TOA brightness temperature: {:g} K, outgoing: {:g} K".format(Ts, Tp))
        print("  --> Global energy imbalance: {:.3f}ppm".format(global_energy_imbalance(quant) * 1e6))
