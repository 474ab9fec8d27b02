"""Output writers for the files parity is judged on: `_tp.dat`, `_integrated_flux.dat`,
`_spec_upflux.dat`, `_spec_downflux.dat`, `_TOA_flux_eclipse.dat`, plus the abort marker.

Counterpart of the reference's `Write` (source/write.py:114-340).  The column layouts ({:g} = 6
significant digits, {:<16.8e} for the spectra) are the file FORMAT that downstream tools
parse (source/tools.py:297, :397), so they are reproduced exactly; numerical parity is checked on the
in-memory arrays, never on these text files.
"""
import os

from . import host_functions as hsfunc


class Write(object):

    @staticmethod
    def _path(quant, read, suffix):
        d = os.path.join(read.output_path, str(quant.name))
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, str(quant.name) + suffix)

    @staticmethod
    def write_abort_file(quant, read):
        with open(Write._path(quant, read, "_ABORT.dat"), "w") as f:
            f.write("Run exceeded the maximum number of iteration steps (%d)." % int(quant.max_nr_iterations))

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

    @staticmethod
    def _spectral(quant, read, suffix, title, label, width, band):
        X, I = int(quant.nbin), int(quant.ninterface)
        with open(Write._path(quant, read, suffix), "w", encoding="utf-8") as f:
            f.write("This file contains the " + title + " spectral flux (per wavelength) at each interface. "
                    "\nSpectral fluxes given in [erg s^-1 cm^-3].")
            f.write("\n{:<8}{:<18}{:21}{:19}".format("bin", "cent_lambda[um]", "low_int_lambda[um]", "delta_lambda[um]"))
            for i in range(I):
                f.write(("{:<%d}{:g}{:<4}" % width).format(label, i, "]"))
            for x in range(X):
                f.write("\n{:<8g}{:<18.9g}{:<21.9g}{:<19.9g}".format(
                    x, quant.opac_wave[x] * 1e4, quant.opac_interwave[x] * 1e4, quant.opac_deltawave[x] * 1e4))
                for i in range(I):
                    f.write("{:<16.8e}".format(band[x + i * X]))

    @staticmethod
    def write_upward_spectral_flux(quant, read):
        Write._spectral(quant, read, "_spec_upflux.dat", "upward", "F_up[", 5, quant.F_up_band)

    @staticmethod
    def write_downward_spectral_flux(quant, read):
        Write._spectral(quant, read, "_spec_downflux.dat", "downward", "F_down[", 7, quant.F_down_band)

    @staticmethod
    def write_TOA_flux_eclipse_depth(quant, read):
        X, L = int(quant.nbin), int(quant.nlayer)
        with open(Write._path(quant, read, "_TOA_flux_eclipse.dat"), "w") as f:
            f.write("This file contains the downward and upward spectral flux (per wavelength) at TOA "
                    "and the secondary eclipse depth (= planet to star flux ratio)."
                    "\nSpectral fluxes given in [erg s^-1 cm^-3].")
            f.write("\n{:<8}{:<18}{:<21}{:<19}{:<16}{:<16}{:<24}".format(
                "bin", "cent_lambda[um]", "low_int_lambda[um]", "delta_lambda[um]", "F_down_at_TOA", "F_up_at_TOA",
                "planet/star flux ratio"))
            for x in range(X):
                f.write("\n{:<8g}{:<18.9g}{:<21.9g}{:<19.9g}".format(
                    x, quant.opac_wave[x] * 1e4, quant.opac_interwave[x] * 1e4, quant.opac_deltawave[x] * 1e4))
                f.write("{:<16g}{:<16g}".format(quant.F_down_band[x + L * X], quant.F_up_band[x + L * X]))
                if quant.T_star > 10:
                    f.write("{:<24g}".format(quant.F_ratio[x]))
                else:
                    f.write("{:<24}".format("not_avail."))

    @staticmethod
    def write_all(quant, read):
        Write.write_tp(quant, read)
        Write.write_integrated_flux(quant, read)
        Write.write_upward_spectral_flux(quant, read)
        Write.write_downward_spectral_flux(quant, read)
        Write.write_TOA_flux_eclipse_depth(quant, read)
