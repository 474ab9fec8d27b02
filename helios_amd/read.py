"""Input side of the driver surface: parameter file + command line, opacity tables, star, albedo, kappa.

Counterpart of the reference's `Read` (source/read.py).  Option names, defaults and derived settings
follow param.dat / read.py:210-988; every option can be overridden on the command line
with the reference's single-dash flags (`-number_of_layers 50`).  File formats:

  * opacity tables / scattering cross-sections / stellar spectra: the reference reads HDF5 through h5py
    (read.py:1041-1103, :1195-1236, :1639-1640).  Here: through h5py where it is installed, otherwise through the
    HDF5 C library itself (helios_amd/hdf5_lite.py, ctypes on libhdf5 >= 1.10); pinned on real h5py-written files
    against the reference's own reader (tests/golden/reader/hdf5/, tests/test_read_hdf5.py).  The same dataset
    names are also accepted from a `.npz` archive (numpy), which is what the synthetic-table generator writes.
  * `opacity mixing = synthetic` (extension): build the seeded synthetic tables of SURVEY.md 8(d)
    in memory -- used by bench.py and the tests, since the real tables cannot be downloaded here.
"""
import argparse
import os

import numpy as np

from . import host_functions as hsfunc
from . import phys_const as pc
from . import synthetic as syn
from .clouds import Cloud

# (param.dat key, attribute, command-line flag, default) for the options that reach the hot path
_OPTIONS = [
    ("name", "name", "name", "0"),
    ("output directory", "output_path", "output_directory", "./output/"),
    ("planet type", "planet_type", "planet_type", "gas"),
    ("TOA pressure [10^-6 bar]", "p_toa", "toa_pressure", "1e-1"),
    ("BOA pressure [10^-6 bar]", "p_boa", "boa_pressure", "1e9"),
    ("run type", "run_type", "run_type", "iterative"),
    ("post-proc. --> path to temperature file", "temp_path", "path_to_temperature_file", "./output/0/0_tp.dat"),
    ("post-proc. --> temperature file format", "temp_file_format", None, "helios"),
    ("scattering", "scat", "scattering", "yes"),
    ("direct irradiation beam", "dir_beam", "direct_irradiation_beam", "no"),
    ("no  --> f factor", "f_factor", "f_factor", "0.5"),
    ("yes --> stellar zenith angle [deg]", "zenith_angle", "stellar_zenith_angle", "60"),
    ("internal temperature [K]", "T_intern", "internal_temperature", "30"),
    ("surface albedo", "input_surf_albedo", "surface_albedo", "0.0"),
    ("file --> path to albedo file", "albedo_file", "path_to_albedo_file", "./input/albedo.dat"),
    ("file --> albedo file format", "albedo_file_format", None, "2 Wavelength micron"),
    ("file --> surface name", "albedo_file_surface_name", "surface_name", "Feldspathic"),
    ("rocky planet --> use f approximation formula", "approx_f", "use_f_approximation_formula", "no"),
    ("opacity mixing", "opacity_mixing", "opacity_mixing", "premixed"),
    ("premixed   --> path to opacity file", "ktable_path", "path_to_opacity_file", "./input/r50_kdistr_solar_eq.h5"),
    ("on-the-fly --> path to species file", "species_file", "path_to_species_file", "./input/species.dat"),
    ("on-the-fly --> file with vertical mixing ratios", "vertical_vmr_file", "file_with_vertical_mixing_ratios", "./input/vmr_mix.txt"),
    ("on-the-fly --> vertical VMR file format", "vertical_vmr_file_format", None, "1 Pressure cgs"),
    ("on-the-fly --> directory with FastChem files", "fastchem_path", "directory_with_fastchem_files", "./input/chemistry/lodders_m0/"),
    ("on-the-fly --> directory with opacity files", "opacity_path", "directory_with_opacity_files", "./input/opacity/r50_kdistr/"),
    ("convective adjustment", "convection", "convective_adjustment", "yes"),
    ("kappa value", "input_kappa_value", "kappa_value", "0.285714"),
    ("file --> kappa file path", "entr_kappa_path", "kappa_file_path", "./input/delad_example.dat"),
    ("stellar spectral model", "stellar_model", "stellar_spectral_model", "blackbody"),
    ("file --> path to stellar spectrum file", "stellar_path", "path_to_stellar_spectrum_file", "./input/star_2022.h5"),
    ("file --> dataset in stellar spectrum file", "stellar_data_set", "dataset_in_stellar_spectrum_file", "/r50_kdistr/phoenix/gj1214"),
    ("planet", "planet", "planet", "manual"),
    ("manual --> surface gravity [cm s^-2]", "g", "surface_gravity", "1000"),
    ("manual --> orbital distance [AU]", "a", "orbital_distance", "0.05"),
    ("manual --> radius planet [R_Jup]", "R_planet", "radius_planet", "1"),
    ("manual --> radius star [R_Sun]", "R_star", "radius_star", "1"),
    ("manual --> temperature star [K]", "T_star", "temperature_star", "5000"),
    ("number of cloud decks", "nr_cloud_decks", "number_of_cloud_decks", "0"),
    ("path to Mie files", "mie_path", "path_to_mie_files", "./input/cloud1/ ./input/cloud2/"),
    ("aerosol radius mode [micron]", "cloud_r_mode", "aerosol_radius_mode", "10 20"),
    ("aerosol radius geometric std dev", "cloud_r_std_dev", "aerosol_radius_geometric_std_dev", "2 1.5"),
    ("cloud mixing ratio", "cloud_mixing_ratio_setting", "cloud_mixing_ratio", "manual"),
    ("file --> path to file with cloud data", "cloud_vmr_file", "path_to_file_with_cloud_data", "./input/cloud_file.txt"),
    ("file --> cloud file format", "cloud_file_format", None, "1 Pressure cgs"),
    ("file --> aerosol name", "cloud_file_species_name", "aerosol_name", "Aerosol1 Aerosol2"),
    ("manual --> cloud bottom pressure [10^-6 bar]", "p_cloud_bot", "cloud_bottom_pressure", "1e5 1e3"),
    ("manual --> cloud bottom mixing ratio", "f_cloud_bot", "cloud_bottom_mixing_ratio", "1e-19 1e-19"),
    ("manual --> cloud to gas scale height ratio", "cloud_to_gas_scale_height", "cloud_to_gas_scale_height_ratio", "0.5 0.5"),
    ("coupling mode", "coupling", "coupling_mode", "no"),
    ("yes --> full output each iteration step", "coupling_full_output", "full_output_each_iteration_step", "no"),
    ("yes --> force eq chem for first iteration", "force_eq_chem", "force_eq_chem_for_first_iteration", "yes"),
    ("yes --> coupling speed up", "coupling_speed_up", "coupling_speed_up", "yes"),
    ("yes --> coupling iteration step", "coupling_iter_nr", "coupling_iteration_step", "0"),
    ("coupling --> write TP profile during run", "write_tp_during_run", "write_tp_profile_during_run", "no"),
    ("coupling --> convergence criterion", "coupl_convergence_limit", "convergence_criterion", "1e-4"),
    ("debugging feedback", "debug", "debugging_feedback", "no"),
    ("precision", "prec", "precision", "double"),
    ("number of layers", "nlayer", "number_of_layers", "automatic"),
    ("isothermal layers", "iso_input", "isothermal_layers", "automatic"),
    ("adaptive interval", "adapt_interval", "adaptive_interval", "20"),
    ("TP profile smoothing", "smooth", "tp_profile_smoothing", "no"),
    ("improved two stream correction", "scat_corr", "improved_two_stream_correction", "no"),
    ("yes --> I2S transition point", "i2s_transition", "i2s_transition_point", "0.1"),
    ("asymmetry factor g_0", "g_0", "asymmetry_factor_g_0", "0"),
    ("diffusivity factor", "diffusivity", "diffusivity_factor", "2"),
    ("second Eddington coefficient", "epsi2", "second_eddington_coefficient", "0.5"),
    ("geometric zenith angle correction", "zenith_correction", "geometric_zenith_angle_correction", "automatic"),
    ("flux calculation method", "flux_calc_method", "flux_calculation_method", "iteration"),
    ("on-the-fly --> k coefficients mixing method", "kcoeff_mixing", "k_coefficients_mixing_method", "RO"),
    ("energy budget correction", "energy_corr", "energy_budget_correction", "automatic"),
    ("convective damping parameter", "input_dampara", "convective_damping_parameter", "automatic"),
    ("plancktable dimension and stepsize", "plancktable", None, "8000 2"),
    ("maximum number of iterations", "max_nr_iterations", "maximum_number_of_iterations", "100000"),
    ("radiative equilibrium criterion", "rad_convergence_limit", "radiative_equilibrium_criterion", "1e-8"),
    ("relax radiative criterion at", "crit_relaxation_numbers", None, "1e4 2e4"),
    ("number of prerun timesteps", "foreplay", "number_of_prerun_timesteps", "0"),
    ("physical timestep [s]", "physical_tstep", "physical_timestep", "no"),
    ("number --> runtime limit [s]", "runtime_limit", "runtime_limit", "86400"),
    ("number --> start from provided TP profile", "force_start_tp_from_file", "start_from_provided_tp_profile", "no"),
    ("include additional heating", "add_heating", "include_additional_heating", "no"),
    ("yes --> path to heating file", "add_heating_path", "path_to_heating_file", "./input/heating_file.txt"),
    ("yes --> heating file format", "add_heating_file_format", None, "1 Pressure cgs Heating 1e7"),
    # extensions for synthetic inputs (not in the reference)
    ("synthetic --> bins layers-are-set-above ntemp npress seed", "synthetic_spec", "synthetic", "300 30 20 20241"),
]


def _yes_no(v):
    if v in ("yes", "no"):
        return np.int32(1 if v == "yes" else 0)
    raise IOError("ERROR: expected 'yes' or 'no', got %r" % (v,))


class Species(object):
    """one entry of the on-the-fly species list (reference source/read.py `Species`, :1324-1443)"""

    def __init__(self, name="", absorbing="no", scattering="no", weight=None, source_for_vmr="constant",
                 mixing_ratio=None):
        self.name = name
        self.absorbing = absorbing            # "yes" / "no"
        self.scattering = scattering          # "yes" / "no"
        self.weight = weight                  # molar weight [g/mol]
        self.source_for_vmr = source_for_vmr  # "constant", "file" or "FastChem"
        self.mixing_ratio = mixing_ratio
        self.fc_name = None
        self.vmr_pretab = None                # [ntemp * npress] for FastChem-tabulated species
        self.vmr_layer = []
        self.vmr_interface = []
        self.opacity_pretab = None            # flat k-table [y + ny*x + ny*nbin*p + ny*nbin*npress*t]
        self.scat_cross_sect_pretab = None    # [nbin]
        self.scat_cross_sect_layer = []       # tiled to [nbin * nlayer] (read.py:1642-1645)
        self.scat_cross_sect_interface = []


class _Table(object):
    """`name in table`, `table[name]` -> numpy array, for .npz archives and HDF5 files (the reference: `h5py.File(name,
    "r")` + `file[name][:]`, source/read.py:1044-1101)"""

    def __init__(self, path):
        self.path = str(path)
        self._npz = self._h5 = None
        if self.path.endswith(".npz"):
            self._npz = dict(np.load(self.path))
            return
        if not os.path.exists(self.path):
            raise IOError("Unable to open file (no such file: %s)" % self.path)
        try:
            import h5py
            self._h5 = h5py.File(self.path, "r")
        except ImportError:
            from . import hdf5_lite
            if not hdf5_lite.available():
                raise IOError("neither h5py nor an HDF5 library is installed: convert %s to .npz (same dataset names), "
                              "install h5py or point HELIOS_HDF5_LIB at a libhdf5.so" % self.path)
            self._h5 = hdf5_lite.File(self.path, "r")

    def __contains__(self, name):
        if self._npz is not None:
            return name in self._npz
        try:
            obj = self._h5[name]
        except (KeyError, ValueError):
            return False
        return not hasattr(obj, "keys")      # a group is not a dataset

    def __getitem__(self, name):
        if self._npz is not None:
            return self._npz[name]
        a = np.asarray(self._h5[name][()])
        # one dtype whichever backend read the file: h5py hands over the storage type (float32, int16, big-endian ...),
        # hdf5_lite fp64 / int64
        if a.dtype.kind == "f":
            return a.astype(np.float64, copy=False)
        if a.dtype.kind in "iu" and (a.dtype.kind == "i" or a.size == 0 or int(a.max()) <= np.iinfo(np.int64).max):
            return a.astype(np.int64, copy=False)
        return a

    def keys(self):
        return list(self._npz.keys()) if self._npz is not None else list(self._h5.keys())


class Read(object):
    """reads the parameter file, the command line and the input tables"""

    def __init__(self):
        self.param_file = "param.dat"
        self.output_path = "./output/"
        self.ktable_path = None
        self.temp_path = None
        self.stellar_model = "blackbody"
        self.entr_kappa_path = None
        self.input_surf_albedo = "0.0"
        self.synthetic_spec = None
        self.cloud = Cloud()

    @staticmethod
    def check_run_configuration(quant):
        """set-up time checks of combinations the reference only trips over deep into a run.  Called by run_helios and by
        the sweep driver right after the input has been read, before any GPU work (the reader itself parses what the
        reference's reader parses, tests/golden/reader)"""
        if quant.iso == 1 and quant.convection == 1 and quant.singlewalk == 0:
            # the reference skips the stability test with isothermal layers and then sums a conv_unstable that was never
            # built (computation.py:1004-1009, quantities.py:134: `sum(None)` raises TypeError after the whole radiative
            # loop has run)
            raise IOError("ERROR: convective adjustment needs non-isothermal layers (the reference cannot run this "
                          "combination either); set 'isothermal layers = no' or 'convective adjustment = no'")

    # ---------------------------------------------------------------------------------------------
    @staticmethod
    def _parse_param_file(path):
        """'key = value   [allowed values] (CL: Y)' lines; everything right of the value is comment"""
        out = {}
        if not os.path.exists(path):
            return out
        with open(path) as f:
            for line in f:
                if "=" not in line or line.lstrip().startswith(("#", "===")):
                    continue
                key, rest = line.split("=", 1)
                rest = rest.split("[", 1)[0].split("(CL:", 1)[0].strip()
                if rest:
                    out[" ".join(key.split())] = rest
        return out

    def read_param_file_and_command_line(self, quant, cloud=None, argv=None):
        parser = argparse.ArgumentParser(description="command-line parameters (same flags as HELIOS)")
        parser.add_argument("-parameter_file", required=False)
        for _key, _attr, flag, _default in _OPTIONS:
            if flag:
                parser.add_argument("-" + flag, required=False)
        args, _unknown = parser.parse_known_args(argv)
        if args.parameter_file:
            self.param_file = args.parameter_file
        file_vals = self._parse_param_file(self.param_file)
        val = {}
        for key, attr, flag, default in _OPTIONS:
            v = file_vals.get(" ".join(key.split()), default)
            if flag and getattr(args, flag) is not None:
                v = getattr(args, flag)
            val[attr] = v

        f64, i32 = np.float64, np.int32
        quant.name = val["name"]
        self.output_path = val["output_path"]
        quant.planet_type = val["planet_type"]
        quant.p_toa, quant.p_boa = f64(val["p_toa"]), f64(val["p_boa"])
        quant.run_type = val["run_type"]
        self.temp_path = val["temp_path"]
        tf = str(val["temp_file_format"]).split()
        self.temp_format = tf[0]
        self.temp_pressure_unit = tf[1] if len(tf) > 1 else "[helios,"      # the reference reads the next token, whatever it is
        quant.scat = _yes_no(val["scat"])
        quant.dir_beam = _yes_no(val["dir_beam"])
        quant.f_factor = f64(val["f_factor"])
        zenith_angle = f64(val["zenith_angle"])
        quant.T_intern = f64(val["T_intern"])
        self.input_surf_albedo = val["input_surf_albedo"]
        self.albedo_file = val["albedo_file"]
        af = str(val["albedo_file_format"]).split()
        self.albedo_file_header_lines = int(af[0])
        self.albedo_file_wavelength_name, self.albedo_file_wavelength_unit = af[1], af[2]
        self.albedo_file_surface_name = val["albedo_file_surface_name"]
        quant.approx_f = _yes_no(val["approx_f"])
        quant.opacity_mixing = val["opacity_mixing"]
        self.ktable_path = val["ktable_path"]
        self.species_file = val["species_file"]
        self.vertical_vmr_file = val["vertical_vmr_file"]
        fmt = str(val["vertical_vmr_file_format"]).split()
        self.vertical_vmr_file_header_lines = int(fmt[0])
        self.vertical_vmr_file_press_name, self.vertical_vmr_file_press_units = fmt[1], fmt[2]
        self.fastchem_path = val["fastchem_path"]
        self.opacity_path = val["opacity_path"]
        quant.convection = _yes_no(val["convection"])
        quant.input_kappa_value = val["input_kappa_value"]
        self.entr_kappa_path = val["entr_kappa_path"]
        self.stellar_model = val["stellar_model"]
        self.stellar_path, self.stellar_data_set = val["stellar_path"], val["stellar_data_set"]
        quant.planet = val["planet"]
        quant.g, quant.a = f64(val["g"]), f64(val["a"])
        quant.R_planet, quant.R_star, quant.T_star = f64(val["R_planet"]), f64(val["R_star"]), f64(val["T_star"])
        cloud = cloud if cloud is not None else self.cloud
        self.cloud = cloud
        cloud.nr_cloud_decks = np.int32(val["nr_cloud_decks"])
        n_decks = max(int(cloud.nr_cloud_decks), 0)

        def per_deck(key, conv, flag):
            """one value per deck from the parameter file; a command-line flag sets a single deck (read.py:759-787)"""
            if flag and getattr(args, flag) is not None:
                return [conv(getattr(args, flag))]
            return [conv(v) for v in str(val[key]).split()[:n_decks]]
        cloud.mie_path = per_deck("mie_path", str, "path_to_mie_files")
        cloud.cloud_r_mode = per_deck("cloud_r_mode", f64, "aerosol_radius_mode")
        cloud.cloud_r_std_dev = per_deck("cloud_r_std_dev", f64, "aerosol_radius_geometric_std_dev")
        cloud.cloud_mixing_ratio_setting = val["cloud_mixing_ratio_setting"]
        cloud.cloud_vmr_file = val["cloud_vmr_file"]
        fmt = str(val["cloud_file_format"]).split()
        cloud.cloud_vmr_file_header_lines = int(fmt[0])
        cloud.cloud_file_press_name, cloud.cloud_file_press_units = fmt[1], fmt[2]
        manual = cloud.cloud_mixing_ratio_setting == "manual"
        cloud.cloud_file_species_name = [] if manual else per_deck("cloud_file_species_name", str, "aerosol_name")
        cloud.p_cloud_bot = per_deck("p_cloud_bot", f64, "cloud_bottom_pressure") if manual else []
        cloud.f_cloud_bot = per_deck("f_cloud_bot", f64, "cloud_bottom_mixing_ratio") if manual else []
        cloud.cloud_to_gas_scale_height = per_deck("cloud_to_gas_scale_height", f64,
                                                   "cloud_to_gas_scale_height_ratio") if manual else []
        quant.debug = _yes_no(val["debug"])
        quant.prec = val["prec"]
        quant.nlayer = val["nlayer"]
        quant.adapt_interval = i32(val["adapt_interval"])
        quant.smooth = _yes_no(val["smooth"])
        quant.scat_corr = _yes_no(val["scat_corr"])
        quant.i2s_transition = f64(val["i2s_transition"])
        quant.g_0 = f64(val["g_0"])
        quant.diffusivity = f64(val["diffusivity"])
        quant.epsi2 = f64(val["epsi2"])
        quant.flux_calc_method = val["flux_calc_method"]
        quant.kcoeff_mixing = val["kcoeff_mixing"]
        quant.input_dampara = val["input_dampara"]
        dim, step = str(val["plancktable"]).split()[:2]
        quant.plancktable_dim, quant.plancktable_step = i32(dim), i32(step)
        quant.max_nr_iterations = i32(float(val["max_nr_iterations"]))
        quant.rad_convergence_limit = f64(val["rad_convergence_limit"])
        quant.crit_relaxation_numbers = [int(float(v)) for v in str(val["crit_relaxation_numbers"]).split()]
        quant.foreplay = i32(val["foreplay"])
        quant.physical_tstep = f64(0 if val["physical_tstep"] == "no" else val["physical_tstep"])
        quant.runtime_limit = f64(val["runtime_limit"])
        quant.add_heating = _yes_no(val["add_heating"])
        quant.add_heating_path = val["add_heating_path"]
        hf = str(val["add_heating_file_format"]).split()
        quant.add_heating_file_header_lines = int(hf[0])
        quant.add_heating_file_press_name, quant.add_heating_file_press_unit = hf[1], hf[2]
        quant.add_heating_file_data_name, quant.add_heating_file_data_conv_factor = hf[3], f64(hf[4])
        quant.force_start_tp_from_file = _yes_no(val["force_start_tp_from_file"])
        # photochemical-kinetics coupling: a file protocol around the run (read.py:521-535, :631-635, :789-803)
        quant.coupling = _yes_no(val["coupling"])
        quant.coupling_full_output = _yes_no(val["coupling_full_output"])
        self.force_eq_chem = val["force_eq_chem"]
        quant.coupling_speed_up = _yes_no(val["coupling_speed_up"])
        quant.coupling_iter_nr = i32(val["coupling_iter_nr"])
        quant.coupl_convergence_limit = f64(val["coupl_convergence_limit"])
        quant.coupl_tp_write_interval = 0 if val["write_tp_during_run"] == "no" else int(val["write_tp_during_run"])
        quant.realtime_plot = i32(0)
        self.synthetic_spec = val["synthetic_spec"]

        # ---- derived settings (read.py:884-985) ----
        if quant.prec != "double":
            raise IOError("ERROR: this build computes in double precision only (SURVEY.md Q16)")
        quant.fl_prec, quant.nr_bytes = np.float64, 8
        if quant.run_type == "iterative":
            quant.singlewalk, quant.iso, quant.energy_correction = i32(0), i32(0), i32(1)
        elif quant.run_type == "post-processing":
            quant.singlewalk, quant.iso, quant.energy_correction = i32(1), i32(1), i32(0)
        else:
            raise IOError("ERROR: unknown run type %r" % quant.run_type)
        quant.dir_angle = f64((180 - zenith_angle) * np.pi / 180.0)
        quant.mu_star = f64(np.cos(quant.dir_angle))
        if self.cloud.nr_cloud_decks < 0:
            raise IOError("\nParameter Error: Number of cloud decks must be >=0. Please correct input value.")
        quant.clouds = i32(1 if self.cloud.nr_cloud_decks > 0 else 0)
        if quant.coupling == 1 and quant.opacity_mixing == "premixed":
            raise IOError("ERROR: Coupling mode cannot be set when a premixed opacity table is used.")
        if quant.coupling == 1 and quant.coupling_full_output == 1:      # one output directory per coupling step
            quant.name = str(quant.name) + "_" + str(quant.coupling_iter_nr)
        if quant.nlayer == "automatic":
            quant.nlayer = i32(np.ceil(10.5 * np.log10(quant.p_boa / quant.p_toa)))
        else:
            quant.nlayer = i32(quant.nlayer)
        if quant.g < 10:
            quant.g = f64(10 ** quant.g)
        if val["iso_input"] != "automatic":
            quant.iso = _yes_no(val["iso_input"])
        quant.epsi = f64(1.0 / quant.diffusivity)
        if val["zenith_correction"] != "automatic":
            quant.geom_zenith_corr = _yes_no(val["zenith_correction"])
        else:
            quant.geom_zenith_corr = i32(1 if zenith_angle > 70 else 0)
        if quant.flux_calc_method == "iterative":
            quant.flux_calc_method = "iteration"
        if val["energy_corr"] != "automatic":
            quant.energy_correction = _yes_no(val["energy_corr"])
        if quant.physical_tstep > 0 and quant.convection == 0:
            raise IOError("ERROR: Physical timesteppings needs convective adjustment switched on.")
        if quant.planet_type == "no_atmosphere":
            quant.no_atmo_mode = i32(1)
            quant.p_toa, quant.p_boa = 1e-3, 2e-3
            quant.scat, quant.convection, quant.nlayer = i32(0), i32(0), i32(2)
        quant.ninterface = i32(quant.nlayer + 1)
        print("\n### Welcome! This run has the name: " + str(quant.name) + ". ###")

    # ---------------------------------------------------------------------------------------------
    @staticmethod
    def _open_table(path):
        """dataset-name -> array mapping of an HDF5 or .npz opacity / star file.  HDF5 through h5py where it is installed,
        otherwise through the HDF5 C library (helios_amd/hdf5_lite.py).  Datasets are read when asked for, nested paths
        ("/r50_kdistr/phoenix/gj1214", the reference's `dataset in stellar spectrum file`) are taken as h5py takes
        them."""
        return _Table(path)

    def read_opac_file(self, quant, path, type="premixed", read_grid_parameters=False):
        """one opacity container as written by the reference's k-table tool (read.py:1041-1103): `kpoints` (or
        `opacities`) flat in [y + ny*x + ny*nbin*p + ny*nbin*npress*t]; for the premixed table also the Rayleigh
        cross-sections and the mean molecular weight (stored in amu, used in g); the grids are taken from the
        premixed table or from the first species table."""
        d = self._open_table(path)
        print("\nReading opacity file:", path)
        opac_k = np.asarray(d["kpoints"] if "kpoints" in d else d["opacities"], np.float64).reshape(-1)
        if type == "premixed":
            quant.opac_scat_cross = np.asarray(d["weighted Rayleigh cross-sections"], np.float64).reshape(-1)
            quant.opac_meanmass = np.asarray(d["meanmolmass"], np.float64).reshape(-1) * pc.AMU
        if type == "premixed" or read_grid_parameters:
            wave = d["center wavelengths"] if "center wavelengths" in d else d["wavelengths"]
            quant.opac_wave = np.asarray(wave, np.float64)
            quant.nbin = np.int32(len(quant.opac_wave))
            quant.gauss_y = np.asarray(d["ypoints"], np.float64) if "ypoints" in d else np.array([0.0])
            quant.ny = np.int32(len(quant.gauss_y))
            if "interface wavelengths" in d:
                quant.opac_interwave = np.asarray(d["interface wavelengths"], np.float64)
            else:       # mid-points, end bins mirrored
                w = quant.opac_wave
                quant.opac_interwave = np.concatenate(([w[0] - (w[1] - w[0]) / 2], (w[1:] + w[:-1]) / 2,
                                                       [w[-1] + (w[-1] - w[-2]) / 2]))
            if "wavelength width of bins" in d:
                quant.opac_deltawave = np.asarray(d["wavelength width of bins"], np.float64)
            else:
                quant.opac_deltawave = np.diff(quant.opac_interwave)
            quant.ktemp = np.asarray(d["temperatures"], np.float64)
            quant.ntemp = np.int32(len(quant.ktemp))
            quant.kpress = np.asarray(d["pressures"], np.float64)
            quant.npress = np.int32(len(quant.kpress))
        return opac_k

    def load_premixed_opacity_table(self, quant):
        if quant.opacity_mixing == "synthetic" or str(self.ktable_path) == "synthetic":
            self.load_synthetic_premixed_table(quant)
            quant.opacity_mixing = "premixed"
        else:
            quant.opac_k = self.read_opac_file(quant, self.ktable_path, type="premixed")
            if getattr(quant, "no_atmo_mode", 0) == 1:      # "no atmosphere": all opacities discarded (read.py:1014-1024)
                quant.opac_k = np.full(len(quant.opac_k), 1e-30)

    def load_synthetic_premixed_table(self, quant, nbin=None, ny=20, ntemp=None, npress=None, seed=None):
        spec = str(self.synthetic_spec).split()
        nbin = int(nbin or spec[0])
        ntemp = int(ntemp or spec[1])
        npress = int(npress or spec[2])
        seed = int(seed or spec[3])
        rng = np.random.default_rng(seed)
        quant.nbin, quant.ny = np.int32(nbin), np.int32(ny)
        quant.opac_interwave, quant.opac_wave, quant.opac_deltawave = syn.wavelength_grid(nbin)
        quant.gauss_y, _w = syn.gauss_points(ny)
        quant.ktemp, quant.kpress = syn.tp_grid(ntemp, npress)
        quant.ntemp, quant.npress = np.int32(ntemp), np.int32(npress)
        quant.opac_k = syn.ktable(rng, nbin, ny, quant.ktemp, quant.kpress, quant.gauss_y)
        quant.opac_scat_cross = syn.rayleigh_table(quant.opac_wave, ntemp, npress)
        quant.opac_meanmass = syn.meanmass_table(ntemp, npress)

    def read_kappa_table_or_use_constant_kappa(self, quant):
        """constant kappa -> kappa_lay/int and c_p = R/kappa; `file` / `water_atmo` -> the (T, P) table of
        kappa (= delad), c_p, entropy [and water phase number] that the interpolation kernels use
        (read.py:1105-1193).  Table layout as read: value[p + npress * t]; rows sorted by T then P in the file."""
        L, I = int(quant.nlayer), int(quant.ninterface)
        for name in ("entr_temp", "entr_press", "entr_kappa", "entr_c_p", "entr_entropy", "entr_phase_number"):
            setattr(quant, name, [])
        quant.entr_ntemp = quant.entr_npress = np.int32(0)
        if quant.convection != 1:
            quant.c_p_lay, quant.kappa_lay, quant.kappa_int = np.zeros(L), np.zeros(L), np.zeros(I)
            return
        try:
            quant.input_kappa_value = np.float64(quant.input_kappa_value)
        except ValueError:
            pass
        if not isinstance(quant.input_kappa_value, str):
            kap = float(quant.input_kappa_value)
            quant.kappa_lay = np.ones(L) * kap
            quant.c_p_lay = np.ones(L) * (pc.R_UNIV / kap)
            quant.kappa_int = np.ones(I) * kap
            return
        water = quant.input_kappa_value == "water_atmo"
        if not water and quant.input_kappa_value != "file":
            raise IOError("ERROR: kappa value must be a number, 'file' or 'water_atmo'")
        print("\nReading kappa/delad values from file (%s format)." % ("water atmospheres" if water else "standard"))
        with open(self.entr_kappa_path, "r") as f:
            rows = [ln.split() for ln in f.readlines()[5 if water else 2:]]
        for col in (r for r in rows if r):
            quant.entr_temp.append(quant.fl_prec(col[0]))
            quant.entr_press.append(quant.fl_prec(col[1]))
            quant.entr_kappa.append(quant.fl_prec(col[2]))
            quant.entr_c_p.append(quant.fl_prec(col[3]))
            if water:
                quant.entr_entropy.append(10 ** quant.fl_prec(col[4]))
                quant.entr_phase_number.append(quant.fl_prec(col[7]))
            else:
                quant.entr_entropy.append(10 ** quant.fl_prec(col[4]) if len(col) > 4 else 0)
        quant.entr_press = np.sort(list(set(quant.entr_press)))
        quant.entr_temp = np.sort(list(set(quant.entr_temp)))
        quant.entr_npress = np.int32(len(quant.entr_press))
        quant.entr_ntemp = np.int32(len(quant.entr_temp))
        quant.kappa_lay, quant.c_p_lay, quant.kappa_int = np.zeros(L), np.zeros(L), np.zeros(I)

    # ---- on-the-fly mixing: species list, mixing ratios, per-species tables (read.py:1324-1645) ---------------
    def read_species_file(self, quant):
        """`species  absorbing  scattering  mixing_ratio` rows; H- is split into its bound-free and free-free parts;
        the first entry must absorb (it starts the mix by the correlated-k rule); weights and FastChem names come from
        the species table"""
        from .species_data import species_lib
        quant.species_list = []
        with open(self.species_file) as f:
            rows = [ln.split() for ln in f.readlines()[1:]]
        for col in (r for r in rows if r):
            names = ["H-_bf", "H-_ff"] if col[0] == "H-" else [col[0]]
            for name in names:
                sp = Species(name=name, absorbing=col[1], scattering=col[2], source_for_vmr=col[3])
                quant.species_list.append(sp)
        if quant.coupling == 1 and getattr(self, "force_eq_chem", "no") == "yes" and getattr(quant, "coupling_iter_nr", 0) == 0:
            for sp in quant.species_list:
                if sp.source_for_vmr == "file":
                    sp.source_for_vmr = "FastChem"
        first = next((k for k, sp in enumerate(quant.species_list) if sp.absorbing == "yes"), None)
        if first is None:
            raise IOError("Oops! At least one species needs to be absorbing. Please double-check your included species "
                          "file. \nAborting ... ")
        quant.species_list.insert(0, quant.species_list.pop(first))
        for sp in quant.species_list:
            entry = species_lib.get(sp.name)
            if entry is None:
                raise IOError("Oops! Species '" + sp.name + "' was not found in the species data base. Please check "
                              "that the name is spelled correctly. If so, add it to helios_amd/species_data.py. Aborting ...")
            sp.weight, sp.fc_name = entry.weight, entry.fc_name
            if sp.fc_name is None and sp.source_for_vmr == "FastChem":
                raise IOError("Oops! FastChem name for species " + sp.name + " unknown. Aborting ...")

    def load_fastchem_data(self):
        """FastChem output: one `chem.dat`, or `chem_low.dat` + `chem_high.dat`; columns `Pbar`, `Tk`, one per species"""
        strip = " !#$%&'()*,./:;<=>?@[\\]^{|}~"
        self.fastchem_data = self.fastchem_data_low = self.fastchem_data_high = None

        def table(name):
            return np.genfromtxt(self.fastchem_path + name, names=True, dtype=None, skip_header=0, deletechars=strip)
        if os.path.exists(self.fastchem_path + "chem.dat"):
            self.fastchem_data = table("chem.dat")
            read_press, read_temp = self.fastchem_data["Pbar"], self.fastchem_data["Tk"]
        else:
            self.fastchem_data_low, self.fastchem_data_high = table("chem_low.dat"), table("chem_high.dat")
            read_press = np.concatenate((self.fastchem_data_low["Pbar"], self.fastchem_data_high["Pbar"]))
            read_temp = np.concatenate((self.fastchem_data_low["Tk"], self.fastchem_data_high["Tk"]))
        self.fastchem_temp = sorted(set(read_temp))
        self.fastchem_press = [p * 1e6 for p in sorted(set(read_press))]
        self.fastchem_n_t, self.fastchem_n_p = len(self.fastchem_temp), len(self.fastchem_press)

    def _fastchem_column(self, name):
        if self.fastchem_data is not None:
            return np.asarray(self.fastchem_data[name], float)
        return np.concatenate((self.fastchem_data_low[name], self.fastchem_data_high[name])).astype(float)

    def read_fastchem_vmr_and_interpolate_to_opacity_PT_grid(self, quant, species):
        """abundance of one species (product of two for pair processes) on the opacity (T, P) grid"""
        pair = ("CIA" in species.name) or species.name in ("H-_ff", "He-")
        if pair:
            n1, n2 = species.fc_name.split("&")
            chem = self._fastchem_column(n1) * self._fastchem_column(n2)
        else:
            chem = self._fastchem_column(species.fc_name)
        return hsfunc.interpolate_vmr_to_opacity_grid(self, quant, chem)

    @staticmethod
    def read_vertical_vmr_and_interpolate_to_helios_press_grid(vmr_file, species, file_press, helios_press):
        """a species' column of the vertical-profile file (product of two for pair processes), linear in log10 P; beyond
        the file the reference's interp1d fill (last value below, first value above) applies"""
        from .species_data import species_lib
        if "CIA" in species.name:
            n1, n2 = species.fc_name.split("&")
            key = {e.fc_name: k for k, e in species_lib.items()}
            prof = np.asarray(vmr_file[key[n1]], float) * np.asarray(vmr_file[key[n2]], float)
        elif species.name == "H-_bf":
            prof = np.asarray(vmr_file["H-"], float)
        elif species.name == "H-_ff":
            prof = np.asarray(vmr_file["H"], float) * np.asarray(vmr_file["e-"], float)
        elif species.name == "He-":
            prof = np.asarray(vmr_file["He"], float) * np.asarray(vmr_file["e-"], float)
        else:
            prof = np.asarray(vmr_file[species.name], float)
        x, xn = np.log10(np.asarray(file_press, float)), np.log10(np.asarray(helios_press, float))
        order = np.argsort(x)
        v = np.interp(xn, x[order], prof[order])
        v = np.where(xn < x.min(), prof[-1], v)
        return np.where(xn > x.max(), prof[0], v)

    def read_species_mixing_ratios(self, quant):
        """`mixing_ratio` column of the species file: a number (constant; `a&b` for CIA pairs), `file` (vertical profile
        file) or `FastChem` (tabulated on the opacity grid here, interpolated along the T-P profile during the run)"""
        L, I = int(quant.nlayer), int(quant.ninterface)
        sources = [sp.source_for_vmr for sp in quant.species_list]
        if "file" in sources:
            vmr_file = np.genfromtxt(self.vertical_vmr_file, names=True, dtype=None,
                                     skip_header=self.vertical_vmr_file_header_lines)
            file_press = np.array(vmr_file[self.vertical_vmr_file_press_name], float)
            file_press *= {"Pa": 10.0, "bar": 1e6}.get(self.vertical_vmr_file_press_units, 1.0)
            p_layer, p_interface = hsfunc.calculate_pressure_levels(quant)
        if "FastChem" in sources:
            self.load_fastchem_data()
        for sp in quant.species_list:
            if sp.source_for_vmr == "file":
                sp.vmr_layer = np.array(self.read_vertical_vmr_and_interpolate_to_helios_press_grid(
                    vmr_file, sp, file_press, p_layer), quant.fl_prec)
                sp.vmr_interface = np.array(self.read_vertical_vmr_and_interpolate_to_helios_press_grid(
                    vmr_file, sp, file_press, p_interface) if quant.iso == 0 else [], quant.fl_prec)
            elif sp.source_for_vmr == "FastChem":
                sp.vmr_pretab = self.read_fastchem_vmr_and_interpolate_to_opacity_PT_grid(quant, sp)
            else:
                value = float(np.prod([float(v) for v in sp.source_for_vmr.split("&")])) if "CIA" in sp.name \
                    else float(sp.source_for_vmr)
                sp.vmr_layer = np.array(np.ones(L) * value, quant.fl_prec)
                if quant.iso == 0:
                    sp.vmr_interface = np.array(np.ones(I) * value, quant.fl_prec)

    def read_species_opacities(self, quant):
        """`<name>_opac_ip_kdistr`, `_opac_ip` or `_opac_ip_sampling` container of every absorber (.h5, or .npz with the
        same dataset names); the first one also provides the wavelength, Gauss-point and (T, P) grids"""
        for s_, sp in enumerate(quant.species_list):
            if sp.absorbing != "yes":
                continue
            for stem in ("_opac_ip_kdistr", "_opac_ip", "_opac_ip_sampling"):
                hits = [self.opacity_path + sp.name + stem + ext for ext in (".h5", ".npz")
                        if os.path.exists(self.opacity_path + sp.name + stem + ext)]
                if hits:
                    sp.opacity_pretab = np.array(self.read_opac_file(quant, hits[0], type="species",
                                                                     read_grid_parameters=(s_ == 0)), quant.fl_prec)
                    break
            else:
                raise IOError("no opacity file for species " + sp.name + " under " + str(self.opacity_path))

    def read_species_scat_cross_sections(self, quant):
        """`rayleigh_<name>` per bin from `scat_cross_sections.{h5,npz}`, tiled over the levels; H2O is computed on the
        device instead (calc_h2o_scat)"""
        table = None
        for sp in quant.species_list:
            if sp.scattering == "yes" and sp.name != "H2O":
                if table is None:
                    path = self.opacity_path + "scat_cross_sections"
                    table = self._open_table(path + (".h5" if os.path.exists(path + ".h5") else ".npz"))
                sp.scat_cross_sect_pretab = [r for r in np.asarray(table["rayleigh_" + sp.name], float)]
                sp.scat_cross_sect_layer = np.array(sp.scat_cross_sect_pretab * int(quant.nlayer), quant.fl_prec)
                if quant.iso == 0:
                    sp.scat_cross_sect_interface = np.array(sp.scat_cross_sect_pretab * int(quant.ninterface), quant.fl_prec)

    def read_star(self, quant):
        """stellar spectrum on the opacity wavelength grid from an HDF5 / .npz container, or the black-body flag"""
        if self.stellar_model == "blackbody":
            quant.starflux = np.zeros(int(quant.nbin), quant.fl_prec)
            quant.real_star = np.int32(0)
            print("\nUsing blackbody flux for the stellar irradiation.")
        elif self.stellar_model == "file":
            d = self._open_table(self.stellar_path)
            key = self.stellar_data_set if self.stellar_data_set in d else str(self.stellar_data_set).strip("/")
            if key not in d:
                raise IOError("There is no such stellar spectrum found. Please check file path and data set.")
            quant.starflux = np.asarray(d[key], np.float64)
            quant.real_star = np.int32(1)
            print("\nReading", str(self.stellar_path) + str(self.stellar_data_set), "as spectral model of the host star.")
            if len(quant.starflux) != quant.nbin:
                raise OverflowError("Stellar spectrum and opacity files have different lengths. Please double-check your "
                                    "input files.")
        else:
            raise IOError("Unknown Stellar model. Please check your input.")

    @staticmethod
    def _profile_on(x_file, y_file, x_new):
        """linear interpolation as the reference's `interp1d(..., fill_value=(y[-1], y[0]))` does it: beyond the low
        end of x the LAST tabulated value, beyond the high end the FIRST (files are usually ordered top-down)"""
        x_file, y_file, x_new = np.asarray(x_file, float), np.asarray(y_file, float), np.asarray(x_new, float)
        order = np.argsort(x_file)
        v = np.interp(x_new, x_file[order], y_file[order])
        v = np.where(x_new < x_file.min(), y_file[-1], v)
        return np.where(x_new > x_file.max(), y_file[0], v)

    def read_or_fill_surf_albedo_array(self, quant):
        """scalar albedo clamped to [1e-8, 0.999], or a surface's column of an albedo file on the model's bins
        (read.py:1238-1264; beyond the file's range its first / last value)"""
        if str(self.input_surf_albedo) == "file":
            tab = np.genfromtxt(self.albedo_file, names=True, dtype=None, skip_header=self.albedo_file_header_lines)
            lam = np.array(tab[self.albedo_file_wavelength_name], float)
            lam *= {"micron": 1e-4, "m": 1e2}.get(self.albedo_file_wavelength_unit, 1.0)
            alb = np.array(tab[self.albedo_file_surface_name], float)
            order = np.argsort(lam)
            quant.surf_albedo = np.interp(np.asarray(quant.opac_wave, float), lam[order], alb[order],
                                          left=alb[0], right=alb[-1])
        else:
            self.input_surf_albedo = max(1e-8, min(0.999, quant.fl_prec(self.input_surf_albedo)))
            quant.surf_albedo = np.ones(int(quant.nbin)) * self.input_surf_albedo

    @staticmethod
    def interpolate_to_own_press(old_press, old_array, new_press):
        return Read._profile_on(np.log10(np.asarray(old_press, float)), old_array, np.log10(np.asarray(new_press, float)))

    def read_temperature_file(self, quant):
        """a T-P profile for post-processing or restarts: `helios` (a `_tp.dat` output: two header lines; columns
        layer, T, P), `TP` or `PT` (two numeric columns, pressure in cgs or bar); interpolated in log10 P onto
        [BOA interface, layer centres] (read.py:1274-1322)"""
        T, P = [], []
        try:
            with open(self.temp_path, "r") as f:
                lines = f.readlines()
        except IOError:
            print("ABORT - TP file not found!")
            raise SystemExit()
        if self.temp_format == "helios":
            for line in lines[2:]:
                col = line.split()
                T.append(quant.fl_prec(col[1]))
                P.append(quant.fl_prec(col[2]))
        else:       # the reference's `elif temp_format == 'TP' or 'PT'` accepts anything else as two-column
            for line in lines:
                col = line.split()
                try:
                    float(col[0])
                except (ValueError, IndexError):
                    continue
                if self.temp_format == "TP":
                    T.append(quant.fl_prec(col[0]))
                    P.append(quant.fl_prec(col[1]))
                elif self.temp_format == "PT":
                    P.append(quant.fl_prec(col[0]))
                    T.append(quant.fl_prec(col[1]))
            if self.temp_pressure_unit == "bar":
                P = [p * 1e6 for p in P]
        quant.T_restart = self.interpolate_to_own_press(P, T, [quant.p_int[0]] + list(quant.p_lay))

    def read_planet_database(self, quant):
        raise IOError("planet database look-ups are outside this build's scope; use planet = manual")
