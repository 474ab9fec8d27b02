#!/usr/bin/env python3
"""Workflow script: read -> grid -> device -> Planck table -> radiation loop -> convection loop ->
post-loop diagnostics -> write.  Same call order as the reference's helios.py:35-137 (`run_helios`), with
the iteration body running on the MI355X through libhelios_hip.so.

    python helios.py -opacity_mixing synthetic -number_of_layers 50 -name test
"""
import sys

from helios_amd import additional_heating as add_heat
from helios_amd import computation as comp
from helios_amd import host_functions as hsfunc
from helios_amd import quantities as quant_mod
from helios_amd import read as read_mod
from helios_amd import write as write_mod


def run_helios(argv=None):
    reader = read_mod.Read()
    keeper = quant_mod.Store()
    computer = comp.Compute()
    keeper._ctx = computer.ctx
    writer = write_mod.Write()

    reader.read_param_file_and_command_line(keeper, reader.cloud, argv)
    reader.check_run_configuration(keeper)
    if keeper.opacity_mixing in ("premixed", "synthetic"):
        reader.load_premixed_opacity_table(keeper)
    elif keeper.opacity_mixing == "on-the-fly":
        reader.read_species_file(keeper)
        reader.read_species_opacities(keeper)
        reader.read_species_scat_cross_sections(keeper)
        reader.read_species_mixing_ratios(keeper)
    else:
        raise IOError("ERROR: opacity mixing must be 'premixed' or 'on-the-fly' (or 'synthetic' for generated tables)")
    reader.read_kappa_table_or_use_constant_kappa(keeper)
    reader.read_or_fill_surf_albedo_array(keeper)
    keeper.dimensions()
    reader.read_star(keeper)
    hsfunc.planet_param(keeper, reader)
    hsfunc.set_up_numerical_parameters(keeper)
    hsfunc.construct_grid(keeper)
    hsfunc.initial_temp(keeper, reader)
    if keeper.approx_f == 1 and keeper.planet_type == "rocky":
        hsfunc.approx_f_from_formula(keeper, reader)
    hsfunc.calc_F_intern(keeper)
    add_heat.load_heating_terms_or_not(keeper)
    reader.cloud.cloud_pre_processing(keeper)

    keeper.create_zero_arrays()
    keeper.convert_input_list_to_array()
    keeper.copy_host_to_device()
    keeper.allocate_on_device()

    computer.construct_planck_table(keeper)
    computer.correct_incident_energy(keeper)
    computer.radiation_loop(keeper, writer, reader, None)
    computer.convection_loop(keeper, writer, reader, None)

    computer.integrate_optdepth_transmission(keeper)
    computer.calculate_contribution_function(keeper)
    if keeper.convection == 1:
        computer.interpolate_entropy(keeper)
        computer.interpolate_phase_state(keeper)
    computer.calculate_mean_opacities(keeper)
    computer.integrate_beamflux(keeper)

    keeper.copy_device_to_host()
    if keeper.conv_unstable is None:
        import numpy as np
        keeper.conv_unstable = np.zeros(int(keeper.nlayer) + 1, np.int32)
    hsfunc.calculate_conv_flux(keeper)
    hsfunc.calc_F_ratio(keeper)
    writer.write_all(keeper, reader)
    if keeper.coupling == 1:
        writer.write_tp_for_coupling(keeper, reader)
        hsfunc.calculate_coupling_convergence(keeper, reader)
    if keeper.approx_f == 1:
        hsfunc.calc_tau_lw_sw(keeper, reader)
    hsfunc.success_message(keeper)
    return keeper


if __name__ == "__main__":
    run_helios(sys.argv[1:])
