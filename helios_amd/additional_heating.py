"""Additional heating term (e.g. UV heating of the upper atmosphere): a heating density profile from a text file on
the model's layers (reference source/additional_heating.py:29-74).  `calc_add_heating_flux` in host_functions turns it
into the per-layer flux that `rad_temp_iter` adds."""
import numpy as np


def read_heating_file(quant):
    tab = np.genfromtxt(quant.add_heating_path, names=True, dtype=None, skip_header=quant.add_heating_file_header_lines)
    press = np.array(tab[quant.add_heating_file_press_name], float)
    unit = quant.add_heating_file_press_unit
    if unit not in ("cgs", "bar", "Pa"):
        raise IOError("Unknown pressure unit in additional heating file. Please double-check your input.")
    press *= {"cgs": 1.0, "bar": 1e6, "Pa": 1e1}[unit]
    heat = np.array(tab[quant.add_heating_file_data_name], float) * quant.add_heating_file_data_conv_factor
    return press, heat


def load_heating_terms_or_not(quant):
    if quant.add_heating == 1:
        from .read import Read
        press, heat = read_heating_file(quant)
        quant.add_heat_dens = Read._profile_on(np.log10(press), heat, np.log10(np.asarray(quant.p_lay, float)))
    else:
        quant.add_heat_dens = np.zeros(int(quant.nlayer))
