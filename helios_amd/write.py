"""Output writers: the fixed-width text files a HELIOS run leaves in `<output>/<name>/`.

Counterpart of the reference's `Write` (source/write.py:63-714, call order helios.py:96-126).  The file
names, title lines and column layouts ({:g} = 6 significant digits; {:<16.8e} for the fluxes; SURVEY.md
Q17) are the FORMAT downstream tools parse (source/tools.py:297, :397), so they are reproduced exactly.
They are data-driven here: three file shapes (per-bin table, per-layer table, free text) described by
small specs instead of one hand-written loop per file.  Numerical parity is checked on the in-memory
arrays, never on these text files.
"""
import os
import shutil

import numpy as np

from . import host_functions as hsfunc
from . import phys_const as pc

_BIN_HEAD = ("bin", "cent_lambda[um]", "low_int_lambda[um]", "delta_lambda[um]")

# per-bin tables: suffix -> (title, 4th-header width, column label, label width, levels, printf cell format,
#                            matrix(quant) -> values[bin, level]).  Reference layouts: band arrays [x + nbin*i]
#                            ("level-major"), Planck arrays [i + x*nlevel] ("bin-major").
def _level_major(name, levels):
    return lambda q: np.asarray(getattr(q, name), float)[:int(q.nbin) * int(getattr(q, levels))].reshape(
        int(getattr(q, levels)), int(q.nbin)).T


def _cloud_opacity(q):
    L, X = int(q.nlayer), int(q.nbin)
    return (np.asarray(q.abs_cross_all_clouds_lay, float)[:L * X].reshape(L, X)
            / np.asarray(q.meanmolmass_lay, float)[:L, None]).T


_BAND_FILES = {
    "_spec_upflux.dat": (
        "This file contains the upward spectral flux (per wavelength) at each interface. "
        "\nSpectral fluxes given in [erg s^-1 cm^-3].",
        19, "F_up[", 5, "ninterface", "%-16.8e", _level_major("F_up_band", "ninterface")),
    "_spec_downflux.dat": (
        "This file contains the downward spectral flux (per wavelength) at each interface. "
        "\nSpectral fluxes given in [erg s^-1 cm^-3].",
        19, "F_down[", 7, "ninterface", "%-16.8e", _level_major("F_down_band", "ninterface")),
    "_direct_beamflux.dat": (
        "This file contains the direct irradiation flux (per wavelength) at each interface. "
        "\nSpectral fluxes given in [erg s^-1 cm^-3].",
        18, "F_dir[", 6, "ninterface", "%-16.8e", _level_major("F_dir_band", "ninterface")),
    "_planck_int.dat": (
        "This file contains the Planck (blackbody) function at each interface. "
        "\nPlanck function given in [erg s^-1 cm^-3 sr^-1].",
        19, "B_int[", 6, "ninterface", "%-16g",
        lambda q: np.asarray(q.planckband_int, float).reshape(int(q.nbin), int(q.ninterface))),
    "_opacities.dat": (
        "This file contains the bin integrated opacities at each layer center "
        "\nOpacity given in [cm^2 g^-1].",
        19, "opac_lay[", 9, "nlayer", "%-15g", _level_major("opac_band_lay", "nlayer")),
    "_cloud_opacities.dat": (
        "This file contains the cloud opacities at each layer center "
        "\nOpacity given in [cm^2 g^-1].",
        19, "cloud_opac[", 11, "nlayer", "%-17g", _cloud_opacity),
    "_Rayleigh_cross_sect.dat": (
        "This file contains Rayleigh scattering cross sections per wavelength at each layer center. "
        "\nCross sections given in [cm^2].",
        19, "scat_cross_sect_lay[", 20, "nlayer", "%-24g", _level_major("scat_cross_lay", "nlayer")),
    "_cloud_scat_cross_sect.dat": (
        "This file contains the cloud scattering cross sections per wavelength at each layer center. "
        "\nCross sections given in [cm^2].",
        19, "cloud_cross_sect_lay[", 21, "nlayer", "%-25g", _level_major("scat_cross_all_clouds_lay", "nlayer")),
    "_g_0.dat": (
        "This file contains the scattering asymmetry parameter values per wavelength at each layer center."
        "\nValues are between -1 and 1.",
        19, "g_0_lay[", 8, "nlayer", "%-16g", _level_major("g_0_tot_lay", "nlayer")),
    "_transmission.dat": (
        "This file contains the transmission function for each layer and waveband.",
        19, "transm_lay[", 11, "nlayer", "%-18g", _level_major("trans_band", "nlayer")),
    "_optdepth.dat": (
        "This file contains the optical depth for each layer and waveband.",
        19, "delta_tau_lay[", 14, "nlayer", "%-20g", _level_major("delta_tau_band", "nlayer")),
    "_cloud_optdepth.dat": (
        "This file contains the cloud optical depth for each layer and waveband.",
        19, "cloud_delta_tau[", 16, "nlayer", "%-22g", _level_major("delta_tau_all_clouds", "nlayer")),
    "_contribution.dat": (
        "This file contains the contribution function for each layer and waveband.",
        19, "contr_func_lay[", 15, "nlayer", "%-22g", _level_major("contr_func_band", "nlayer")),
    "_transweight.dat": (
        "This file contains the transmission weighting function for each layer and waveband. "
        "The units are [erg s^-1 cm^-3 sr^-1]",
        19, "transm_weight_lay[", 18, "nlayer", "%-25g", _level_major("trans_weight_band", "nlayer")),
}


def _or_not_calculated(value, width):
    return ("{:<%ds}" % width).format("not_calculated") if value == 0 else ("{:<%dg}" % width).format(value)


def _mean_or_flag(value):
    """-3 marks a mean opacity that could not be evaluated (write.py:54-60)"""
    return "{:<20}".format("temp_too_low") if value == -3 else "{:<20g}".format(value)


def _format_rows(quant, values, cell):
    """the rows of a per-bin table -- "\n" + bin index, centre, lower edge and width in micron, then one cell per level --
    formatted by the library's host utility on several threads (hx_host_format_rows: printf's conversions are the ones of
    the `%` operator; byte-identical to `_format_rows_python`, which the tests hold it to)"""
    import ctypes
    from . import _lib
    l = _lib.lib()
    X = int(quant.nbin)
    prefix = np.empty((X, 4))
    prefix[:, 0] = np.arange(X)
    prefix[:, 1] = np.asarray(quant.opac_wave, float)[:X] * 1e4
    prefix[:, 2] = np.asarray(quant.opac_interwave, float)[:X] * 1e4
    prefix[:, 3] = np.asarray(quant.opac_deltawave, float)[:X] * 1e4
    v = np.ascontiguousarray(values, np.float64)
    text, n = ctypes.c_void_p(), ctypes.c_size_t()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = l.hx_host_format_rows(prefix.ctypes.data_as(dp), v.ctypes.data_as(dp), X, int(v.shape[1]), cell.encode(),
                               min(64, os.cpu_count() or 1), ctypes.byref(text), ctypes.byref(n))
    if rc != 0:
        raise _lib.HeliosHipError("hx_host_format_rows failed with status %d (cell format %r)" % (rc, cell))
    try:
        return ctypes.string_at(text, n.value)        # ASCII bytes
    finally:
        l.hx_host_free(text)


def _format_rows_python(quant, values, cell):
    """the same rows through Python's own formatting (the statement of what the rows are; used by the tests)"""
    row_fmt = cell * int(values.shape[1])
    out = []
    for x in range(int(quant.nbin)):
        out.append(Write._bin_prefix(quant, x))
        out.append(row_fmt % tuple(values[x].tolist()))
    return "".join(out).encode("ascii")


class Write(object):

    # ---- plumbing -----------------------------------------------------------------------------------
    @staticmethod
    def _dir(quant, read):
        d = os.path.join(read.output_path, str(quant.name))
        os.makedirs(d, exist_ok=True)
        return d

    @staticmethod
    def _path(quant, read, suffix):
        return os.path.join(Write._dir(quant, read), str(quant.name) + suffix)

    @staticmethod
    def _bin_prefix(quant, x):
        return "\n{:<8g}{:<18.9g}{:<21.9g}{:<19.9g}".format(
            x, quant.opac_wave[x] * 1e4, quant.opac_interwave[x] * 1e4, quant.opac_deltawave[x] * 1e4)

    @staticmethod
    def _band_file(quant, read, suffix):
        """one row per wavelength bin: bin geometry, then one cell per level (a whole row is formatted by one
        printf-style call: the files hold up to millions of cells)"""
        title, w4, label, lw, levels, cell, matrix = _BAND_FILES[suffix]
        nlev = int(getattr(quant, levels))
        X = int(quant.nbin)
        values = matrix(quant)
        head = title + ("\n{:<8}{:<18}{:21}{:%d}" % w4).format(*_BIN_HEAD)
        head += "".join(("{:<%d}{:g}{:<4}" % lw).format(label, i, "]") for i in range(nlev))
        with open(Write._path(quant, read, suffix), "wb") as f:
            f.write(head.encode("utf-8"))
            f.write(_format_rows(quant, values, cell))

    @staticmethod
    def convert_1_0_to_yes_no(variable):
        return "yes" if variable == 1 else "no"

    @staticmethod
    def write_physical_timestep(variable):
        return "no" if variable == 0 else "{:g}".format(variable)

    # ---- markers ------------------------------------------------------------------------------------
    @staticmethod
    def write_abort_file(quant, read):
        with open(Write._path(quant, read, "_ABORT.dat"), "w") as f:
            f.write("Run exceeded the maximum number of iteration steps (%d)." % int(quant.max_nr_iterations))

    @staticmethod
    def write_criterion_warning_file(quant, read):
        """only when the driver had to relax the convergence criterion (computation.py:950-975)"""
        if getattr(quant, "relaxed_criterion_trigger", 0) != 1:
            return
        with open(Write._path(quant, read, "_convergence_warning.dat"), "w") as f:
            f.write("WARNING: Due to exceeding runtime the convergence criterion has been made more loose over time.\n")
            f.write("The final relative criterion used is: {:.1e} \n".format(quant.rad_convergence_limit))
            f.write("Even with a looser (not loser) criterion, the model results may still be accurate enough. "
                    "Use at your own discretion!")

    @staticmethod
    def create_output_dir_and_copy_param_file(read, quant):
        d = Write._dir(quant, read)
        src = getattr(read, "param_file", None)
        if src and os.path.isfile(src):
            shutil.copyfile(src, os.path.join(d, str(quant.name) + "_" + os.path.basename(src)))

    # ---- per-layer tables ---------------------------------------------------------------------------
    @staticmethod
    def write_tp(quant, read):
        T_bright = hsfunc.temp_calcs(quant)[4]
        L = int(quant.nlayer)
        conv_cols = quant.iso == 0 and quant.convection == 1 and quant.conv_unstable is not None
        with open(Write._path(quant, read, "_tp.dat"), "w") as f:
            f.write("This file contains the corresponding layer temperatures and pressures, and the altitude "
                    "and the height of each layer.")
            f.write("\n{:<8}{:<18}{:<24}{:<21}{:<23}{:<30}{:<32}{:<18}".format(
                "layer", "temp.[K]", "press.[10^-6bar]", "altitude[cm]", "height.of.layer[cm]",
                "conv.unstable?[1:yes,0:no]", "conv.lapse-rate?[1:yes,0:no]", "pl.eff.temp.[K]"))
            f.write("\n{:<8}{:<18g}{:<24g}{:<21g}{:<23}".format(
                "BOA", quant.T_lay[L], quant.p_int[0], quant.z_lay[0] - 0.5 * quant.delta_z_lay[0], "not_avail."))
            if conv_cols:
                f.write("{:<30g}{:<32g}".format(quant.conv_unstable[L], quant.conv_layer[L]))
            else:
                f.write("{:<30}{:<32}".format("not_calculated", "not_calculated"))
            f.write("{:<18g}".format(T_bright))
            for i in range(L):
                f.write("\n{:<8g}{:<18g}{:<24g}{:<21g}{:<23g}".format(
                    i, quant.T_lay[i], quant.p_lay[i], quant.z_lay[i], quant.delta_z_lay[i]))
                if conv_cols:
                    f.write("{:<30g}{:<32g}".format(quant.conv_unstable[i], quant.conv_layer[i]))
                else:
                    f.write("{:<30}{:<32}".format("not_calculated", "not_calculated"))

    @staticmethod
    def write_tp_cut(quant, read):
        """the profile below 1e-6 bar (p_lay > 0.099 dyn cm^-2 as written in the reference)"""
        L = int(quant.nlayer)
        with open(Write._path(quant, read, "_tp_cut.dat"), "w") as f:
            f.write("This file contains the corresponding layer temperatures and pressures.")
            f.write("\n{:<8}{:<18}{:<24}".format("layer", "temp.[K]", "press.[10^-6bar]"))
            f.write("\n{:<8}{:<18g}{:<24g}".format("BOA", quant.T_lay[L], quant.p_int[0]))
            for i in range(L):
                if quant.p_lay[i] > 0.099:
                    f.write("\n{:<8g}{:<18g}{:<24g}".format(i, quant.T_lay[i], quant.p_lay[i]))

    @staticmethod
    def write_colmass_mu_cp_entropy(quant, read):
        with open(Write._path(quant, read, "_colmass_mu_cp_kappa_entropy.dat"), "w") as f:
            f.write("This file contains the total pressure and the column mass difference, mean molecular weight "
                    "and specific heat capacity of each layer.")
            f.write("\n{:<8}{:<24}{:<26}{:<21}{:<32}{:<23}{:<30}".format(
                "layer", "cent.press.[10^-6bar]", "delta_col.mass[g cm^-2]", "mean mol. weight",
                "spec.heat cap.[erg mol^-1 K^-1]", "adiabatic coefficient", "entropy [erg g^-1 K^-1]"))
            for i in range(int(quant.nlayer)):
                f.write("\n{:<8g}{:<24g}{:<26g}{:<21g}".format(
                    i, quant.p_lay[i], quant.delta_colmass[i], quant.meanmolmass_lay[i] / pc.AMU))
                f.write(_or_not_calculated(quant.c_p_lay[i], 32))
                f.write(_or_not_calculated(quant.kappa_lay[i], 23))
                f.write(_or_not_calculated(quant.entropy_lay[i], 30))

    @staticmethod
    def write_phase_state(quant, read):
        """only for the `water_atmo` kappa format (write.py:210-233)"""
        if str(quant.input_kappa_value) != "water_atmo":
            return
        with open(Write._path(quant, read, "_state.dat"), "w") as f:
            f.write("Checks the phase state of the water atmosphere. If '1' the water in the atmosphere is vaporous "
                    "or supercritical. If '<1' atmosphere might be unstable, i.e., water in liquid or solid form.")
            f.write("\n{:<8}{:<18}{:<24}{:<24}".format(
                "layer", "temp.[K]", "press.[10^-6bar]",
                "state_of_water (0: liquid or solid, 1: vapor or supercritical)"))
            for i in range(int(quant.nlayer)):
                if quant.p_lay[i] > 0.99:
                    f.write("\n{:<8g}{:<18g}{:<24g}{:<24g}".format(
                        i, quant.T_lay[i], quant.p_lay[i], quant.phase_number_lay[i]))

    @staticmethod
    def write_cloud_mixing_ratio(quant, read):
        with open(Write._path(quant, read, "_cloud_mixing_ratio.dat"), "w") as f:
            f.write("This file contains the cloud volume mixing ratio (= n_cloud/n_gas) at each vertical layer.")
            f.write("\n{:<8}{:<24}{:<18}".format("layer", "press.[10^-6bar]", "cloud_vmr"))
            for i in range(int(quant.nlayer)):
                f.write("\n{:<8g}{:<24g}{:<18g}".format(i, quant.p_lay[i], quant.f_all_clouds_lay[i]))

    @staticmethod
    def write_integrated_flux(quant, read):
        L = int(quant.nlayer)
        with open(Write._path(quant, read, "_integrated_flux.dat"), "w") as f:
            f.write("This file contains the integrated total and net fluxes at each interface resp. "
                    "layer. \nFluxes given in [erg s^-1 cm^-2].")
            f.write("\n{:<20}{:<24}{:<25}{:<25}{:<23}{:<25}{:<34}{:<24}{:<24}{:<12}".format(
                "interface", "press.[10^-6bar]", "F_down", "F_up", "F_net", "F_dir",
                "delta_F_net (layer quantity)", "F_net_conv", "F_add_heat", "F_intern"))
            for i in range(int(quant.ninterface)):
                f.write("\n{:<20g}{:<24g}{:<25g}{:<25g}{:<23g}{:<25g}".format(
                    i, quant.p_int[i], quant.F_down_tot[i], quant.F_up_tot[i], quant.F_net[i], quant.F_dir_tot[i]))
                if quant.singlewalk == 0 and i < L:
                    f.write("{:<34g}".format(quant.F_net_diff[i]))
                else:
                    f.write("{:<34}".format("not_avail."))
                f.write("{:<24g}".format(quant.F_net_conv[i]))
                if i < L:
                    f.write("{:<24g}".format(quant.F_add_heat_lay[i]))
                else:
                    f.write("{:<24}".format("not_avail."))
                if i == 0:
                    f.write("{:<12g}".format(quant.F_intern))

    def write_mean_extinction(self, quant, read):
        with open(Write._path(quant, read, "_mean_extinct.dat"), "w") as f:
            f.write("This file contains the Rosseland and Planck mean opacities of layers & optical depths "
                    "summed up to a certain layer, weighted either by the blackbody function "
                    "with the stellar or the planetary atmospheric temperature."
                    "\nMean opacity given in [cm^2 g^-1].")
            f.write("\n{:<10}{:<20}{:<20}{:<20}{:<20}{:<20}{:<20}{:<20}{:<20}{:<20}".format(
                "layer", "press.[10^-6bar]", "Planck_opac_T_lay", "Ross_opac_T_lay", "Planck_opac_T_star",
                "Ross_opac_T_star", "Planck_tau_T_lay", "Ross_tau_T_lay", "Planck_tau_T_star", "Ross_tau_T_star"))
            means = (quant.planck_opac_T_pl, quant.ross_opac_T_pl, quant.planck_opac_T_star, quant.ross_opac_T_star)
            for i in range(int(quant.nlayer)):
                f.write("\n{:<8g}{:<20g}".format(i, quant.p_lay[i]))
                f.write("".join(_mean_or_flag(m[i]) for m in means))
                f.write("".join(_mean_or_flag(hsfunc.sum_mean_optdepth(quant, i, m)) for m in means))

    # ---- per-bin tables -----------------------------------------------------------------------------
    @staticmethod
    def write_upward_spectral_flux(quant, read):
        Write._band_file(quant, read, "_spec_upflux.dat")

    @staticmethod
    def write_downward_spectral_flux(quant, read):
        Write._band_file(quant, read, "_spec_downflux.dat")

    @staticmethod
    def write_direct_spectral_beam_flux(quant, read):
        Write._band_file(quant, read, "_direct_beamflux.dat")

    @staticmethod
    def write_planck_interface(quant, read):
        if quant.iso == 0:
            Write._band_file(quant, read, "_planck_int.dat")

    @staticmethod
    def write_planck_center(quant, read):
        """layer centres plus the stellar and the internal-temperature Planck columns"""
        X, L = int(quant.nbin), int(quant.nlayer)
        head = ("This file contains the Planck (blackbody) function at each layer center and "
                "from the stellar (2nd last column) and internal (last column) temperatures. "
                "\nPlanck function given in [erg s^-1 cm^-3 sr^-1].")
        head += "\n{:<8}{:<18}{:21}{:19}".format(*_BIN_HEAD)
        head += "".join("{:<6}{:g}{:<4}".format("B_lay[", i, "]") for i in range(L))
        head += "{:<16}{:<16}".format("Planck_T_star", "Planck_T_intern")
        rows = np.asarray(quant.planckband_lay, float).reshape(X, L + 2)
        with open(Write._path(quant, read, "_planck_cent.dat"), "wb") as f:
            f.write(head.encode("utf-8"))
            f.write(_format_rows(quant, rows, "%-16g"))

    @staticmethod
    def write_opacities(quant, read):
        Write._band_file(quant, read, "_opacities.dat")

    @staticmethod
    def write_cloud_opacities(quant, read):
        Write._band_file(quant, read, "_cloud_opacities.dat")

    @staticmethod
    def write_Rayleigh_cross_sections(quant, read):
        Write._band_file(quant, read, "_Rayleigh_cross_sect.dat")

    @staticmethod
    def write_cloud_scat_cross_sections(quant, read):
        Write._band_file(quant, read, "_cloud_scat_cross_sect.dat")

    @staticmethod
    def write_g_0(quant, read):
        Write._band_file(quant, read, "_g_0.dat")

    @staticmethod
    def write_transmission(quant, read):
        Write._band_file(quant, read, "_transmission.dat")

    @staticmethod
    def write_opt_depth(quant, read):
        Write._band_file(quant, read, "_optdepth.dat")

    @staticmethod
    def write_cloud_opt_depth(quant, read):
        Write._band_file(quant, read, "_cloud_optdepth.dat")

    @staticmethod
    def write_contribution_function(quant, read):
        Write._band_file(quant, read, "_contribution.dat")

    @staticmethod
    def write_trans_weight_function(quant, read):
        Write._band_file(quant, read, "_transweight.dat")

    @staticmethod
    def write_surface_albedo(quant, read):
        with open(Write._path(quant, read, "_surf_albedo.dat"), "w") as f:
            f.write("This file contains the surface albedo per wavelength.")
            if str(read.input_surf_albedo) == "file":
                f.write("\nThe surface material used is: " + str(read.albedo_file_surface_name))
            else:
                f.write("\nA value was chosen manually, hence all the values below are constant.")
            f.write("\n{:<8}{:<18}{:<21}{:<19}{:<16}".format(*(_BIN_HEAD + ("surface_albedo",))))
            for x in range(int(quant.nbin)):
                f.write(Write._bin_prefix(quant, x) + "{:<16g}".format(quant.surf_albedo[x]))

    @staticmethod
    def write_TOA_flux_eclipse_depth(quant, read):
        X, L = int(quant.nbin), int(quant.nlayer)
        with open(Write._path(quant, read, "_TOA_flux_eclipse.dat"), "w") as f:
            f.write("This file contains the downward and upward spectral flux (per wavelength) at TOA "
                    "and the secondary eclipse depth (= planet to star flux ratio)."
                    "\nSpectral fluxes given in [erg s^-1 cm^-3].")
            f.write("\n{:<8}{:<18}{:<21}{:<19}{:<16}{:<16}{:<24}".format(
                *(_BIN_HEAD + ("F_down_at_TOA", "F_up_at_TOA", "planet/star flux ratio"))))
            for x in range(X):
                f.write(Write._bin_prefix(quant, x))
                f.write("{:<16g}{:<16g}".format(quant.F_down_band[x + L * X], quant.F_up_band[x + L * X]))
                if quant.T_star > 10:
                    f.write("{:<24g}".format(quant.F_ratio[x]))
                else:
                    f.write("{:<24}".format("not_avail."))

    @staticmethod
    def write_flux_ratio_only(quant, read):
        """wavelength [um] and planet/star flux ratio only (e.g. for PandExo)"""
        with open(Write._path(quant, read, "_flux_ratio.dat"), "w") as f:
            for x in range(int(quant.nbin)):
                f.write("{:<18.9g}".format(quant.opac_wave[x] * 1e4))
                if quant.T_star > 10:
                    f.write("{:<12g}\n".format(quant.F_ratio[x]))
                else:
                    f.write("{:<12}\n".format("not_avail."))

    # ---- photochemical-kinetics coupling: the T-P profile handed to the chemistry code (write.py:717-771) ---------
    @staticmethod
    def _coupling_tp_path(quant, read, step, previous=False):
        """`<name>_tp_coupling_<step>.dat`; with one output directory per coupling step the previous step lives in
        `<base>_<step>/`"""
        name = str(quant.name)
        if previous and quant.coupling_full_output == 1:
            name = name[:name.rfind("_") + 1] + str(step)
        return os.path.join(read.output_path, name, name + "_tp_coupling_" + str(step) + ".dat")

    @staticmethod
    def read_coupling_tp(path, prec=float):
        with open(path, "r") as f:
            return [prec(ln.split()[1]) for ln in f.readlines()[1:] if len(ln.split()) > 1]

    @staticmethod
    def write_tp_for_coupling(quant, read):
        """surface first, then the layers; from the second coupling step on the mean of this and the previous profile
        when the speed-up is on"""
        L, step = int(quant.nlayer), int(quant.coupling_iter_nr)
        T_new = [quant.T_lay[L]] + [quant.T_lay[i] for i in range(L)]
        if quant.coupling_speed_up == 1 and step > 0:
            T_prev = Write.read_coupling_tp(Write._coupling_tp_path(quant, read, step - 1, previous=True), quant.fl_prec)
            T_new = [0.5 * T_new[i] + 0.5 * T_prev[i] for i in range(len(T_prev))]
        Write._dir(quant, read)
        with open(Write._coupling_tp_path(quant, read, step), "w") as f:
            f.write("{:<24}{:<18}".format("press.[10^-6bar]", "temp.[K]"))
            f.write("\n{:<24g}{:<18g}".format(quant.p_int[0], T_new[0]))
            for i in range(L):
                f.write("\n{:<24g}{:<18g}".format(quant.p_lay[i], T_new[i + 1]))

    # ---- everything, in the reference's order (helios.py:100-126) --------------------------------------
    def write_all(self, quant, read):
        Write.create_output_dir_and_copy_param_file(read, quant)
        for w in (Write.write_colmass_mu_cp_entropy, Write.write_integrated_flux,
                  Write.write_downward_spectral_flux, Write.write_upward_spectral_flux,
                  Write.write_TOA_flux_eclipse_depth, Write.write_direct_spectral_beam_flux,
                  Write.write_planck_interface, Write.write_planck_center, Write.write_tp, Write.write_tp_cut,
                  Write.write_opacities, Write.write_cloud_mixing_ratio, Write.write_cloud_opacities,
                  Write.write_Rayleigh_cross_sections, Write.write_cloud_scat_cross_sections, Write.write_g_0,
                  Write.write_transmission, Write.write_opt_depth, Write.write_cloud_opt_depth,
                  Write.write_trans_weight_function, Write.write_contribution_function,
                  self.write_mean_extinction, Write.write_flux_ratio_only, Write.write_phase_state,
                  Write.write_surface_albedo, Write.write_criterion_warning_file):
            w(quant, read)
