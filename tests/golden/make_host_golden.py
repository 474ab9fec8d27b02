"""Generates golden vectors for the HOST side (pure-Python grid / convective-adjustment / bookkeeping
functions and the text-file writers) by importing the reference's own modules in the build container.

    python tests/golden/make_host_golden.py          # needs /root/reference; never runs on the GPU box

The reference's `source/host_functions.py` and `source/write.py` import pycuda and astropy at module
top; neither is installed, and none of the functions exercised here touches them.  The modules are
therefore imported with empty stand-ins for `pycuda.*` and an `astropy.constants` object carrying the
cgs constants of helios_amd/phys_const.py ("pure-numpy host functions can be exercised
in this container only by pre-seeding sys.modules").  Run under the image's conda interpreter
(`/opt/conda/bin/python3.9`, which has astropy 4.3.1 and h5py 3.3.0) the script takes the REAL astropy and
h5py instead (only pycuda stays empty): the 51 writer files and parsed.json come out byte for byte the same,
the numeric fixture within 1e-13 (numpy 1.26's `10.0 ** x` makes the seeded inputs one ulp different from
numpy 2.2's) -- checked in round 4; the committed fixtures are the system interpreter's.  Outputs are data only:

    tests/golden/host_functions.npz      inputs + results of the grid / convection / bookkeeping functions
    tests/golden/writer/*.dat            every output file the reference writes for one small seeded state
    tests/golden/reader/parsed.json      what the reference's reader parses from reader/param_sample.dat
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from helios_amd import phys_const as pc  # noqa: E402


def import_reference():
    for name in ("pycuda", "pycuda.driver", "pycuda.autoinit", "pycuda.gpuarray", "pycuda.compiler"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pycuda.compiler"].SourceModule = object
    sys.modules["pycuda"].driver = sys.modules["pycuda.driver"]
    sys.modules["pycuda"].gpuarray = sys.modules["pycuda.gpuarray"]

    class _Q(object):
        def __init__(self, v):
            self.value = v
            self.cgs = self
            self.esu = self

    try:        # the build image's conda interpreter (/opt/conda/bin/python3.9) has the real astropy 4.3.1
        for gone, fn in (("asscalar", lambda a: a.item()), ("alen", len)):     # names astropy 4.3 lists, numpy >= 1.23 lacks
            if not hasattr(np, gone):
                setattr(np, gone, fn)
        import astropy.constants  # noqa: F401
        real_astropy = True
    except ImportError:
        real_astropy = False
    if real_astropy:
        sys.path.insert(0, REF)
        from source import host_functions as ref_hs
        from source import write as ref_write
        print("host goldens with the real astropy", sys.modules["astropy"].__version__)
        return ref_hs, ref_write

    const = types.ModuleType("astropy.constants")
    # CODATA 2018 / IAU 2015 values for the constants helios_amd/phys_const.py does not need itself
    extra = dict(m_e=9.1093837015e-28, e=4.803204712570263e-10, M_sun=1.988409870698051e33,
                 M_jup=1.8981245973360505e30, M_earth=5.972167867791379e27, sigma_T=6.6524587321e-25)
    for attr, val in dict(c=pc.C, k_B=pc.K_B, h=pc.H, R=pc.R_UNIV, N_A=pc.N_A, sigma_sb=pc.SIGMA_SB, au=pc.AU,
                          u=pc.AMU, R_sun=pc.R_SUN, R_jup=pc.R_JUP, R_earth=pc.R_EARTH, G=pc.G, **extra).items():
        setattr(const, attr, _Q(val))
    astropy = types.ModuleType("astropy")
    astropy.constants = const
    sys.modules["astropy"] = astropy
    sys.modules["astropy.constants"] = const
    sys.path.insert(0, REF)
    from source import host_functions as ref_hs
    from source import write as ref_write
    return ref_hs, ref_write


# ---- shared seeded states (also imported by tests/test_host_golden.py) ----------------------------------
def grid_state(nlayer=13, p_boa=1e8, p_toa=1.0, g=2500.0):
    q = types.SimpleNamespace()
    q.nlayer, q.ninterface = np.int32(nlayer), np.int32(nlayer + 1)
    q.p_boa, q.p_toa, q.g = p_boa, p_toa, g
    q.p_lay, q.p_int, q.delta_colmass, q.delta_col_upper, q.delta_col_lower = [], [], [], [], []
    return q


def convection_state(seed, nlayer=24, kind="deep"):
    """a column with a super-adiabatic region: `deep` (bottom + surface), `detached` (two zones with a hole)"""
    rng = np.random.default_rng(seed)
    q = grid_state(nlayer)
    lev = [q.p_boa * (q.p_toa / q.p_boa) ** (i / (2 * nlayer - 1)) for i in range(2 * nlayer)]
    q.p_lay = np.array(lev[1::2])
    q.p_int = np.array(lev[0::2] + [q.p_toa * (q.p_toa / q.p_boa) ** (1 / (2 * nlayer - 1))])
    kappa = 2.0 / 7.0
    q.kappa_lay = np.full(nlayer, kappa) * (1 + 0.05 * rng.uniform(-1, 1, nlayer))
    q.kappa_int = np.full(nlayer + 1, kappa) * (1 + 0.05 * rng.uniform(-1, 1, nlayer + 1))
    q.c_p_lay = np.full(nlayer, 3.5 * pc.R_UNIV) * (1 + 0.02 * rng.uniform(-1, 1, nlayer))
    q.meanmolmass_lay = np.full(nlayer, 2.3 * pc.AMU)
    T = 1800.0 * (q.p_lay / q.p_lay[0]) ** 0.12                  # stable background
    if kind == "deep":
        T[:8] = 2600.0 * (q.p_lay[:8] / q.p_lay[0]) ** 0.45      # steep (unstable) bottom
        T_surf = 3000.0
    else:
        T[2:7] = 2200.0 * (q.p_lay[2:7] / q.p_lay[2]) ** 0.5
        T[9:13] = T[8] * (q.p_lay[9:13] / q.p_lay[8]) ** 0.42
        T_surf = T[0] * 1.0001
    q.T_lay = np.append(T * (1 + 0.002 * rng.uniform(-1, 1, nlayer)), T_surf)
    q.conv_unstable = np.zeros(nlayer + 1, np.int32)
    q.conv_layer = np.zeros(nlayer + 1, np.int32)
    q.marked_red = np.zeros(nlayer + 1, np.int32)
    q.F_intern = 1.0e5
    q.input_dampara = "automatic"
    q.T_star = 5000.0 if kind == "deep" else 0.0
    q.F_add_heat_sum = np.zeros(nlayer)
    q.F_smooth_sum = np.zeros(nlayer)
    q.F_down_tot = 10.0 ** rng.uniform(5, 8, nlayer + 1)
    q.F_up_tot = (q.F_down_tot + q.F_intern) * (1 + 0.05 * rng.uniform(-1, 1, nlayer + 1))
    q.iter_value = 40
    q.input_kappa_value = "0.2857"
    q.T_surf = T_surf
    return q


def writer_state(seed=7, nbin=4, nlayer=5, iso=0, T_star=5000.0, convection=1):
    """every array the writers touch, filled with seeded values of mixed magnitude"""
    rng = np.random.default_rng(seed)
    q = types.SimpleNamespace()
    X, L, I = nbin, nlayer, nlayer + 1
    q.name = "gold"
    q.fl_prec = np.float64
    q.nbin, q.nlayer, q.ninterface = np.int32(X), np.int32(L), np.int32(I)
    q.iso, q.convection, q.singlewalk = np.int32(iso), np.int32(convection), np.int32(0)
    q.T_star, q.R_star, q.a, q.R_planet = T_star, 6.96e10, 7.5e11, 7.1e9
    q.f_factor, q.F_intern, q.T_intern = 0.25, 12345.678, 30.0
    q.dir_beam, q.mu_star = np.int32(0), -0.5
    q.max_nr_iterations = 1000
    q.rad_convergence_limit, q.relaxed_criterion_trigger = 1e-7, 1
    q.input_kappa_value = "water_atmo"
    q.planet_type = "gas"
    edges = 0.3e-4 * (500.0 / 0.3) ** (np.arange(X + 1) / X)
    q.opac_interwave = edges
    q.opac_wave = 0.5 * (edges[1:] + edges[:-1])
    q.opac_deltawave = np.diff(edges)
    q.p_int = 10.0 ** np.linspace(8, -0.2, I)
    q.p_lay = np.sqrt(q.p_int[1:] * q.p_int[:-1])
    q.T_lay = rng.uniform(300, 2500, L + 1)
    q.z_lay = np.cumsum(rng.uniform(1e5, 5e6, L))
    q.delta_z_lay = rng.uniform(1e5, 5e6, L)
    q.conv_unstable = rng.integers(0, 2, L + 1).astype(np.int32)
    q.conv_layer = rng.integers(0, 2, L + 1).astype(np.int32)
    q.delta_colmass = rng.uniform(1e-3, 1e3, L)
    q.meanmolmass_lay = rng.uniform(2, 30, L) * pc.AMU
    q.c_p_lay = rng.uniform(1e8, 4e8, L)
    q.c_p_lay[1] = 0
    q.kappa_lay = rng.uniform(0.1, 0.4, L)
    q.kappa_lay[2] = 0
    q.entropy_lay = np.zeros(L)
    q.entropy_lay[0] = 3.21e8
    q.phase_number_lay = rng.integers(0, 2, L).astype(float)
    q.f_all_clouds_lay = 10.0 ** rng.uniform(-12, -4, L)

    def band(n, lo, hi):
        return 10.0 ** rng.uniform(lo, hi, X * n)
    for name, n, lo, hi in (("F_up_band", I, -3, 9), ("F_down_band", I, -120, 9), ("F_dir_band", I, -300, 8),
                            ("opac_band_lay", L, -8, 4), ("abs_cross_all_clouds_lay", L, -30, -20),
                            ("scat_cross_lay", L, -30, -22), ("scat_cross_all_clouds_lay", L, -30, -20),
                            ("trans_band", L, -8, 0), ("delta_tau_band", L, -9, 3), ("delta_tau_all_clouds", L, -9, 1),
                            ("contr_func_band", L, -3, 7), ("trans_weight_band", L, -3, 7)):
        setattr(q, name, band(n, lo, hi))
    q.g_0_tot_lay = rng.uniform(-1, 1, X * L)
    q.planckband_int = 10.0 ** rng.uniform(-20, 8, X * I)
    q.planckband_lay = 10.0 ** rng.uniform(-20, 8, X * (L + 2))
    q.surf_albedo = rng.uniform(0, 1, X)
    for name in ("F_down_tot", "F_up_tot", "F_dir_tot"):
        setattr(q, name, 10.0 ** rng.uniform(3, 9, I))
    q.F_net = q.F_up_tot - q.F_down_tot
    q.F_net_diff = rng.normal(0, 1e3, L)
    q.F_add_heat_lay = rng.uniform(0, 10, L)
    q.F_add_heat_sum = np.cumsum(q.F_add_heat_lay)
    q.F_smooth_sum = np.zeros(L)
    for name in ("planck_opac_T_pl", "ross_opac_T_pl", "planck_opac_T_star", "ross_opac_T_star"):
        v = 10.0 ** rng.uniform(-6, 2, L)
        v[rng.integers(0, L)] = -3
        setattr(q, name, v)
    q.star_corr_factor = 1.0
    q.F_ratio = []
    return q


def mixing_state(seed=21, nlayer=9, ntemp=6, npress=5):
    """VMR tables on the opacity (T, log10 P) grid, a T-P profile that leaves the grid on both sides, and a
    species list with one CIA pair and H- continuum entries (excluded from the mean molecular mass)"""
    rng = np.random.default_rng(seed)
    q = types.SimpleNamespace()
    q.fl_prec = np.float64
    q.nlayer, q.ninterface = np.int32(nlayer), np.int32(nlayer + 1)
    q.ktemp = np.linspace(200.0, 2200.0, ntemp)
    q.log_kpress = np.linspace(0.0, 8.0, npress)
    q.log_p_lay = np.linspace(9.0, -1.0, nlayer)              # beyond the table at both ends
    q.log_p_int = np.linspace(9.5, -1.5, nlayer + 1)
    q.T_prof_lay = np.concatenate(([100.0, 200.0, 2200.0, 3000.0], rng.uniform(200, 2200, nlayer - 4)))
    q.T_prof_int = rng.uniform(150, 2500, nlayer + 1)
    q.species_list = []
    for name, weight in (("H2O", 18.0153), ("CO2", 44.01), ("CIA_H2H2", 4.03), ("H-_ff", 1.0), ("He-", 4.0),
                         ("H2", 2.016)):
        sp = types.SimpleNamespace()
        sp.name, sp.weight = name, weight
        sp.vmr_pretab = 10.0 ** rng.uniform(-8, -0.3, (ntemp, npress))
        sp.vmr_layer = 10.0 ** rng.uniform(-8, -0.3, nlayer)
        sp.vmr_interface = 10.0 ** rng.uniform(-8, -0.3, nlayer + 1)
        q.species_list.append(sp)
    return q


def radeq_state(seed, limit):
    rng = np.random.default_rng(seed)
    q = types.SimpleNamespace()
    L = 12
    q.nlayer = np.int32(L)
    q.T_lay = rng.uniform(500, 2000, L + 1)
    q.conv_layer = (rng.uniform(size=L + 1) < 0.3).astype(np.int32)
    q.F_intern = 5.0e4
    q.F_add_heat_sum = np.cumsum(rng.uniform(0, 50, L))
    q.F_smooth_sum = np.zeros(L)
    q.F_down_tot = 10.0 ** rng.uniform(6, 8, L + 1)
    q.F_net = q.F_intern * (1 + 10.0 ** rng.uniform(-9, -1, L + 1) * rng.choice([-1, 1], L + 1))
    q.rad_convergence_limit = limit
    q.iter_value = 7
    return q


def start_state(dir_beam, T_star):
    q = types.SimpleNamespace()
    q.fl_prec = np.float64
    q.planet = "manual"
    q.g, q.a, q.R_planet, q.R_star, q.T_star = 3.4, 0.05, 1.2, 0.9, T_star
    q.dir_beam, q.f_factor, q.mu_star = dir_beam, 0.25, -0.6
    q.singlewalk, q.force_start_tp_from_file, q.physical_tstep = 0, 0, 0
    q.nlayer = np.int32(7)
    q.T_intern = 150.0
    q.ny = 20
    return q


def write_mie_files(directory, seed):
    """synthetic LX-MIE output: one file per radius of the reference's hard-wired grid, 7 columns of which the
    reader uses wavelength [micron], scattering and absorption cross-section and g_0 (columns 0, 3, 4, 6)"""
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)
    lam = 0.2 * (600.0 / 0.2) ** (np.arange(37) / 36.0)          # micron
    slope = rng.uniform(0.8, 1.6)
    for r in 10 ** np.arange(-2, 3.1, 0.1):
        x = 2 * np.pi * r / lam
        q_sca = x ** 4 / (1 + x ** 4) * (1.5 + 0.3 * np.sin(3 * np.log(x)))
        q_abs = 0.2 * x ** slope / (1 + x ** slope)
        geo = np.pi * (r * 1e-4) ** 2
        g0 = 0.85 * x ** 2 / (1 + x ** 2)
        with open(os.path.join(directory, "r{:.6f}.dat".format(r)), "w") as f:
            f.write("# wavelength[micron] size_par ext scat abs albedo g_0\n")
            for k in range(len(lam)):
                f.write("%.8e %.6e %.8e %.8e %.8e %.6e %.8e\n" % (lam[k], x[k], geo * (q_sca[k] + q_abs[k]),
                                                                  geo * q_sca[k], geo * q_abs[k],
                                                                  q_sca[k] / (q_sca[k] + q_abs[k]), g0[k]))


def cloud_case(tag, workdir):
    """(cloud settings, quant) of the two golden cloud cases; Mie files and the VMR file are written to `workdir`"""
    q = types.SimpleNamespace()
    nbin, nlayer = 14, 11
    q.nbin, q.nlayer, q.ninterface = np.int32(nbin), np.int32(nlayer), np.int32(nlayer + 1)
    edges = 0.1e-4 * (1000.0 / 0.1) ** (np.arange(nbin + 1) / nbin)      # 0.1 - 1000 micron: wider than the tables
    q.opac_interwave = edges
    q.opac_wave = 0.5 * (edges[1:] + edges[:-1])
    lev = [1e8 * (1e0 / 1e8) ** (i / (2 * nlayer - 1)) for i in range(2 * nlayer)]
    q.p_lay = lev[1::2]
    q.p_int = lev[0::2] + [1e0 * (1e0 / 1e8) ** (1 / (2 * nlayer - 1))]
    q.clouds = np.int32(1)
    s = dict(nr_cloud_decks=2, mie_path=[os.path.join(workdir, "mie1") + "/", os.path.join(workdir, "mie2") + "/"],
             cloud_r_mode=[0.5, 8.0], cloud_r_std_dev=[1.8, 1.3])
    write_mie_files(s["mie_path"][0], 41)
    write_mie_files(s["mie_path"][1], 42)
    if tag == "manual":
        q.iso = np.int32(0)
        s.update(cloud_mixing_ratio_setting="manual", p_cloud_bot=[2e6, 3e3], f_cloud_bot=[1e-12, 4e-14],
                 cloud_to_gas_scale_height=[0.4, 1.0])
    else:
        q.iso = np.int32(1)
        path = os.path.join(workdir, "cloud_vmr.txt")
        with open(path, "w") as f:
            f.write("cloud mixing ratios\nPressure Dust Ice\n")
            for p, a, b in ((5e-1, 1e-16, 3e-15), (5e0, 1e-15, 2e-14), (5e1, 1e-14, 1e-13), (5e2, 3e-14, 5e-15)):
                f.write("%g %g %g\n" % (p, a, b))
        s.update(cloud_mixing_ratio_setting="file", cloud_vmr_file=path, cloud_vmr_file_header_lines=1,
                 cloud_file_press_name="Pressure", cloud_file_press_units="Pa", cloud_file_species_name=["Dust", "Ice"])
    return s, q


CLOUD_KEYS = ["f_all_clouds_lay", "f_all_clouds_int", "abs_cross_all_clouds_lay", "abs_cross_all_clouds_int",
              "scat_cross_all_clouds_lay", "scat_cross_all_clouds_int", "g_0_all_clouds_lay", "g_0_all_clouds_int"]


def run_clouds(cloud_obj, tag, workdir):
    s, q = cloud_case(tag, workdir)
    for k, v in s.items():
        setattr(cloud_obj, k, v)
    cloud_obj.cloud_pre_processing(q)
    return q


def write_kappa_file(path, water, seed=3, const_kappa=None):
    """a kappa (= delad) / c_p / entropy table in the reference's two ASCII layouts (read.py:1121-1162)"""
    rng = np.random.default_rng(seed)
    temps = np.linspace(100.0, 3100.0, 7)
    press = 10.0 ** np.linspace(0.0, 9.0, 6)
    with open(path, "w") as f:
        for _ in range(5 if water else 2):
            f.write("# header line\n")
        for t in temps:
            for p in press:
                kap = const_kappa if const_kappa is not None else rng.uniform(0.15, 0.4)
                cp = 8.31446261815324e7 / kap if const_kappa is not None else rng.uniform(2.5e8, 4e8)
                row = [t, p, kap, cp, rng.uniform(7.5, 9.5)]
                if water:
                    row += [rng.uniform(0, 1), rng.uniform(0, 1), float(rng.integers(0, 2))]
                f.write(" ".join("%.10e" % v for v in row) + "\n")
            f.write("\n")


def kappa_state(mode, path):
    q = types.SimpleNamespace()
    q.fl_prec = np.float64
    q.convection, q.iso = np.int32(1), np.int32(0)
    q.nlayer, q.ninterface = np.int32(6), np.int32(7)
    q.input_kappa_value = mode
    for name in ("entr_temp", "entr_press", "entr_kappa", "entr_c_p", "entr_entropy", "entr_phase_number"):
        setattr(q, name, [])
    r = types.SimpleNamespace()
    r.entr_kappa_path = path
    return q, r


KAPPA_KEYS = ["entr_temp", "entr_press", "entr_kappa", "entr_c_p", "entr_entropy", "entr_phase_number", "kappa_lay",
              "c_p_lay", "kappa_int"]


def write_species_inputs(workdir, seed=9, nbin=6, ny=4, sorted_k=False):
    """everything the on-the-fly readers open: species file, FastChem table, vertical-profile file, per-species opacity
    containers and Rayleigh cross-sections (.npz with the reference's HDF5 dataset names)"""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(workdir, "opac"), exist_ok=True)
    os.makedirs(os.path.join(workdir, "chem"), exist_ok=True)
    with open(os.path.join(workdir, "species.dat"), "w") as f:
        f.write("species      absorbing       scattering         mixing_ratio\n\n"
                "H2   no  yes  FastChem\n\nH2O  yes yes FastChem\n\nCO2  yes no  file\nCH4 yes no 1e-4\n"
                "H-   yes no  FastChem\nHe  no yes 0.15\nCIA_H2H2 yes no FastChem\nCIA_H2He yes no 0.85&0.15\n"
                "CIA_CO2CO2 yes no file\n")
    temps, press_bar = np.linspace(300.0, 2700.0, 5), 10.0 ** np.linspace(-5.0, 2.0, 4)
    cols = ["H2O1", "C1O2", "H2", "He", "H1-", "H", "e-"]
    with open(os.path.join(workdir, "chem", "chem.dat"), "w") as f:
        f.write("#P(bar) T(k) n_<tot>(cm-3) " + " ".join(cols) + "\n")
        for t in temps:
            for p in press_bar:
                f.write("%.6e %.6e %.6e " % (p, t, p * 1e6 / (1.380649e-16 * t)) +
                        " ".join("%.8e" % v for v in 10.0 ** rng.uniform(-12, -0.5, len(cols))) + "\n")
    with open(os.path.join(workdir, "vmr.txt"), "w") as f:
        f.write("vertical mixing ratios\nPressure CO2 H2O\n")
        for p in (1e-2, 1.0, 1e2, 1e4, 1e6):                       # Pa
            f.write("%g %.6e %.6e\n" % (p, 10.0 ** rng.uniform(-6, -3), 10.0 ** rng.uniform(-6, -3)))
    ntemp, npress = 3, 3
    grids = {"center wavelengths": 1e-4 * 2.0 ** np.arange(nbin), "ypoints": (np.arange(ny) + 0.5) / ny,
             "temperatures": np.array([200.0, 1000.0, 1800.0]), "pressures": np.array([1e2, 1e5, 1e8])}
    for k, name in enumerate(("H2O", "CO2", "CH4", "H-_bf", "H-_ff", "CIA_H2H2", "CIA_H2He", "CIA_CO2CO2")):
        d = dict(grids) if name == "H2O" else {}
        tab = 10.0 ** rng.uniform(-8, 2, (ntemp, npress, nbin, ny))
        if sorted_k:                # k-distributions increase with the Gauss point, as real tables do
            tab = np.sort(tab, axis=-1)
        d["kpoints" if k % 2 == 0 else "opacities"] = tab.reshape(-1)
        stem = ("_opac_ip_kdistr", "_opac_ip", "_opac_ip_sampling")[k % 3]
        np.savez(os.path.join(workdir, "opac", name + stem + ".npz"), **d)
    np.savez(os.path.join(workdir, "opac", "scat_cross_sections.npz"),
             rayleigh_H2=10.0 ** rng.uniform(-28, -24, nbin), rayleigh_He=10.0 ** rng.uniform(-29, -25, nbin))


def species_reader_setup(reader, quant, workdir):
    reader.species_file = os.path.join(workdir, "species.dat")
    reader.fastchem_path = os.path.join(workdir, "chem") + "/"
    reader.opacity_path = os.path.join(workdir, "opac") + "/"
    reader.vertical_vmr_file = os.path.join(workdir, "vmr.txt")
    reader.vertical_vmr_file_header_lines = 1
    reader.vertical_vmr_file_press_name, reader.vertical_vmr_file_press_units = "Pressure", "Pa"
    reader.force_eq_chem = "no"
    quant.fl_prec = np.float64
    quant.coupling, quant.coupling_iter_nr, quant.iso = 0, 0, np.int32(0)
    quant.nlayer, quant.ninterface = np.int32(7), np.int32(8)
    quant.p_boa, quant.p_toa = 1e8, 1e0
    quant.species_list = []


def run_species_readers(reader, quant):
    reader.read_species_file(quant)
    reader.read_species_opacities(quant)
    reader.read_species_scat_cross_sections(quant)
    reader.read_species_mixing_ratios(quant)


def species_record(quant):
    rec = {"names": np.array([sp.name for sp in quant.species_list]),
           "weights": np.array([sp.weight for sp in quant.species_list], float),
           "fc_names": np.array([str(sp.fc_name) for sp in quant.species_list])}
    for k in ("opac_wave", "opac_interwave", "opac_deltawave", "gauss_y", "ktemp", "kpress"):
        rec["grid." + k] = np.array(getattr(quant, k), float)
    for n, sp in enumerate(quant.species_list):
        for attr in ("vmr_layer", "vmr_interface", "vmr_pretab", "opacity_pretab", "scat_cross_sect_layer",
                     "scat_cross_sect_interface"):
            v = getattr(sp, attr, None)
            if v is not None and len(np.atleast_1d(v)) > 0:
                rec["%d.%s" % (n, attr)] = np.array(v, float)
    return rec


def write_misc_inputs(workdir, seed=13):
    """albedo file, T-P files in the three accepted formats, heating file, stellar spectrum container"""
    rng = np.random.default_rng(seed)
    with open(os.path.join(workdir, "albedo.dat"), "w") as f:
        f.write("# surface albedos\n# source: synthetic\nWavelength Basaltic Granitoid\n")
        for lam in (0.3, 0.5, 1.0, 2.0, 5.0, 12.0, 25.0):
            f.write("%g %.5f %.5f\n" % (lam, rng.uniform(0.02, 0.4), rng.uniform(0.1, 0.7)))
    press = 10.0 ** np.linspace(8.2, -0.5, 12)                      # cgs, bottom-up like a HELIOS output
    temp = rng.uniform(400, 2200, 12)
    with open(os.path.join(workdir, "tp_helios.dat"), "w") as f:
        f.write("header one\nlayer temp press altitude\n")
        f.write("BOA %.6f %.6e 0\n" % (temp[0], press[0]))
        for i in range(1, 12):
            f.write("%d %.6f %.6e %g\n" % (i - 1, temp[i], press[i], 1e5 * i))
    with open(os.path.join(workdir, "tp_TP.dat"), "w") as f:
        f.write("# T[K] P[bar]\n")
        for t, p in zip(temp[::-1], press[::-1]):
            f.write("%.6f %.6e\n" % (t, p / 1e6))
    with open(os.path.join(workdir, "tp_PT.dat"), "w") as f:
        f.write("pressure temperature\n")
        for t, p in zip(temp, press):
            f.write("%.6e %.6f\n" % (p, t))
    with open(os.path.join(workdir, "heating.txt"), "w") as f:
        f.write("extra heating\nPressure Heating Other\n")
        for p in (1e-3, 1e-1, 1e1, 1e3):                          # bar
            f.write("%g %.5e %g\n" % (p, 10.0 ** rng.uniform(-9, -5), 1.0))
    np.savez(os.path.join(workdir, "star.npz"), **{"/grid/some_star": 10.0 ** rng.uniform(3, 7, 9)})


def misc_state():
    q = types.SimpleNamespace()
    q.fl_prec = np.float64
    q.nbin, q.nlayer = np.int32(9), np.int32(6)
    q.opac_wave = 1e-4 * 0.2 * 2.2 ** np.arange(9)                 # 0.2 ... 110 micron: wider than the albedo file
    lev = [3e7 * (2.0 / 3e7) ** (i / 11) for i in range(12)]
    q.p_lay, q.p_int = lev[1::2], lev[0::2] + [1.5]
    q.add_heating = np.int32(1)
    q.add_heating_file_header_lines, q.add_heating_file_press_name = 1, "Pressure"
    q.add_heating_file_press_unit, q.add_heating_file_data_name = "bar", "Heating"
    q.add_heating_file_data_conv_factor = np.float64(1e7)
    q.delta_z_lay = np.linspace(2e5, 9e6, 6)
    q.F_add_heat_lay = np.zeros(6)
    q.F_add_heat_sum = np.zeros(6)
    # rocky-planet f approximation
    q.name, q.R_star, q.a, q.T_star, q.p_boa, q.tau_lw, q.f_factor = "rock", 3.2e10, 3.5e11, 3300.0, 3e7, 1, 0.6667
    q.delta_tau_band = 10.0 ** np.random.default_rng(5).uniform(-4, 1.5, 9 * 6)
    q.opac_deltawave = q.opac_wave * 0.7
    q.T_lay = np.linspace(900, 500, 7)
    return q


def run_coupling(hs, Write, workdir, full_output, speed_up=1):
    """three coupling steps of a fake run whose profile settles; returns the text of every file the protocol leaves"""
    rng = np.random.default_rng(17)
    L = 6
    base = rng.uniform(500, 2000, L + 1)
    texts = []
    for step, wobble in enumerate((0.05, 1e-6, 1e-7)):
        q = types.SimpleNamespace()
        q.fl_prec = np.float64
        q.nlayer = np.int32(L)
        q.name = "cpl_%d" % step if full_output else "cpl"
        q.coupling_iter_nr, q.coupling_speed_up = np.int32(step), np.int32(speed_up)
        q.coupling_full_output = np.int32(full_output)
        q.coupl_convergence_limit, q.singlewalk = 1e-4, np.int32(0)
        q.T_lay = base * (1 + wobble * rng.uniform(-1, 1, L + 1))
        q.p_int = 10.0 ** np.linspace(8, 0, L + 1)
        q.p_lay = np.sqrt(q.p_int[1:] * q.p_int[:-1])
        r = types.SimpleNamespace(output_path=workdir + "/")
        os.makedirs(os.path.join(workdir, q.name), exist_ok=True)
        Write.write_tp_for_coupling(q, r)
        hs.calculate_coupling_convergence(q, r)
        d = os.path.join(workdir, q.name)
        for fn in sorted(os.listdir(d)):
            if ("_tp_coupling_%d" % step) in fn or "convergence" in fn:
                texts.append(fn + "\n" + open(os.path.join(d, fn)).read())
    return texts


def reader_stub(out_dir):
    r = types.SimpleNamespace()
    r.output_path = out_dir if out_dir.endswith("/") else out_dir + "/"
    r.input_surf_albedo = 0.1
    r.albedo_file_surface_name = None
    r.param_file = "param.dat"
    return r


WRITERS = ["write_colmass_mu_cp_entropy", "write_integrated_flux", "write_downward_spectral_flux",
           "write_upward_spectral_flux", "write_TOA_flux_eclipse_depth", "write_direct_spectral_beam_flux",
           "write_planck_interface", "write_planck_center", "write_tp", "write_tp_cut", "write_opacities",
           "write_cloud_mixing_ratio", "write_cloud_opacities", "write_Rayleigh_cross_sections",
           "write_cloud_scat_cross_sections", "write_g_0", "write_transmission", "write_opt_depth",
           "write_cloud_opt_depth", "write_trans_weight_function", "write_contribution_function",
           "write_mean_extinction", "write_flux_ratio_only", "write_phase_state", "write_surface_albedo",
           "write_criterion_warning_file"]


def run_writers(hs, Write, q, out_dir):
    """the reference's output sequence (helios.py:96-126) without the parameter-file copy"""
    r = reader_stub(out_dir)
    os.makedirs(os.path.join(out_dir, q.name), exist_ok=True)
    hs.calculate_conv_flux(q)
    hs.calc_F_ratio(q)
    w = Write()
    for name in WRITERS:
        getattr(w, name)(q, r)


READER_CASES = {
    "file_only": [],
    "with_command_line": ["-name", "cl_run", "-number_of_layers", "42", "-scattering", "no",
                          "-direct_irradiation_beam", "no", "-f_factor", "0.3", "-isothermal_layers", "yes",
                          "-surface_gravity", "981", "-temperature_star", "0", "-convective_adjustment", "yes",
                          "-flux_calculation_method", "iteration", "-planet_type", "gas", "-toa_pressure", "1e-2",
                          "-geometric_zenith_angle_correction", "no", "-energy_budget_correction", "yes",
                          "-radiative_equilibrium_criterion", "1e-6", "-physical_timestep", "120",
                          "-stellar_zenith_angle", "30", "-kappa_value", "0.25"],
    "post_processing": ["-run_type", "post-processing", "-planet_type", "no_atmosphere"],
    "one_deck_command_line": ["-number_of_cloud_decks", "1", "-path_to_mie_files", "./mie/enstatite/",
                              "-aerosol_radius_mode", "3.5", "-aerosol_radius_geometric_std_dev", "1.7",
                              "-cloud_bottom_pressure", "2e4", "-cloud_bottom_mixing_ratio", "3e-13",
                              "-cloud_to_gas_scale_height_ratio", "0.7"],
    "two_decks_manual": [],
    "two_decks_file": [],
}
# per case: textual substitutions applied to reader/param_sample.dat before it is parsed
READER_EDITS = {
    "two_decks_manual": [("number of cloud decks =                               0",
                          "number of cloud decks =                               2")],
    "two_decks_file": [("number of cloud decks =                               0",
                        "number of cloud decks =                               2"),
                       ("cloud mixing ratio =                                  manual",
                        "cloud mixing ratio =                                  file")],
}


def reader_param_file(case, workdir):
    """the sample parameter file, edited for `case` if needed; returns its path"""
    sample = os.path.join(HERE, "reader", "param_sample.dat")
    edits = READER_EDITS.get(case)
    if not edits:
        return sample
    text = open(sample).read()
    for a, b in edits:
        assert a in text
        text = text.replace(a, b)
    path = os.path.join(workdir, "param_%s.dat" % case)
    with open(path, "w") as f:
        f.write(text)
    return path


def reader_fixture():
    """what the reference's reader makes of tests/golden/reader/param_sample.dat (+ command-line flags)"""
    import json
    try:
        import h5py  # noqa: F401  (the real one under the conda interpreter)
    except ImportError:
        sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    from source import clouds as ref_clouds
    from source import quantities as ref_quant
    from source import read as ref_read
    # the reference rewrites ./source/kernels.cu for the chosen precision at this point; the reference tree is
    # read-only here and nothing else of that step is needed
    ref_read.Read.set_prec_in_cudafile = lambda self, quant: None
    out = {}
    import tempfile
    for case, flags in READER_CASES.items():
        argv0 = sys.argv
        with tempfile.TemporaryDirectory() as wd:
            sys.argv = ["helios.py", "-parameter_file", reader_param_file(case, wd)] + flags
            try:
                k, r, c = ref_quant.Store(), ref_read.Read(), ref_clouds.Cloud()
                r.read_param_file_and_command_line(k, c)
            finally:
                sys.argv = argv0
        rec = {}
        for prefix, obj in (("quant.", k), ("read.", r), ("cloud.", c)):
            for a, v in vars(obj).items():
                if a == "param_file" or v is None:
                    continue
                if isinstance(v, (str, int, float, np.integer, np.floating)):
                    rec[prefix + a] = v if isinstance(v, str) else float(v)
                elif isinstance(v, list) and v and all(isinstance(e, (int, float, np.integer, np.floating)) for e in v):
                    rec[prefix + a] = [float(e) for e in v]
                elif isinstance(v, list) and v and all(isinstance(e, str) for e in v):
                    rec[prefix + a] = list(v)
                elif isinstance(v, list) and not v and prefix == "cloud.":
                    rec[prefix + a] = []
        out[case] = dict(flags=flags, parsed=rec)
    with open(os.path.join(HERE, "reader", "parsed.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    hs, ref_write = import_reference()
    reader_fixture()
    data = {}
    # grid construction
    for tag, kw in (("g13", dict(nlayer=13)), ("g50", dict(nlayer=50, p_boa=1e9, p_toa=1e-1, g=980.0))):
        q = grid_state(**kw)
        hs.construct_grid(q)
        for k in ("p_lay", "p_int", "delta_colmass", "delta_col_upper", "delta_col_lower"):
            data["%s.%s" % (tag, k)] = np.array(getattr(q, k))
        data[tag + ".args"] = np.array([kw.get("nlayer"), kw.get("p_boa", 1e8), kw.get("p_toa", 1.0), kw.get("g", 2500.0)])
    # convective adjustment: check / mark / correct chain
    for tag, seed, kind in (("deep", 11, "deep"), ("detached", 12, "detached")):
        q = convection_state(seed, kind=kind)
        data["conv.%s.T_in" % tag] = q.T_lay.copy()
        hs.conv_check(q)
        data["conv.%s.unstable0" % tag] = q.conv_unstable.copy()
        hs.mark_convective_layers(q, stitching=0)
        data["conv.%s.layer0" % tag] = q.conv_layer.copy()
        q = convection_state(seed, kind=kind)
        hs.convective_adjustment(q)
        data["conv.%s.T_out" % tag] = np.array(q.T_lay, float)
        data["conv.%s.layer" % tag] = np.array(q.conv_layer)
        data["conv.%s.unstable" % tag] = np.array(q.conv_unstable)
    # altitude grid
    for ptype in ("gas", "rocky"):
        q = convection_state(3)
        q.planet_type = ptype
        q.delta_z_lay = np.random.default_rng(5).uniform(1e5, 4e6, int(q.nlayer))
        q.z_lay = np.zeros(int(q.nlayer))
        hs.calculate_height_z(q)
        data["z.%s" % ptype] = q.z_lay.copy()
        data["z.%s.dz" % ptype] = q.delta_z_lay.copy()
    # bookkeeping used by the writers
    q = writer_state()
    hs.calculate_conv_flux(q)
    hs.calc_F_ratio(q)
    data["book.F_net_conv"] = np.array(q.F_net_conv)
    data["book.F_ratio"] = np.array(q.F_ratio)
    data["book.tau"] = np.array([[hs.sum_mean_optdepth(q, i, getattr(q, m)) for i in range(int(q.nlayer))]
                                 for m in ("planck_opac_T_pl", "ross_opac_T_pl")], float)
    data["book.temp_calcs"] = np.array(hs.temp_calcs(q), float)
    # on-the-fly mixing: VMR profile interpolation (scipy bilinear spline in the reference), mean molecular mass
    q = mixing_state()
    for n, sp in enumerate(q.species_list):
        data["mix.vmr_lay.%d" % n] = np.array(hs.interpolate_grid_to_lay_or_int(
            q.log_kpress, q.ktemp, sp.vmr_pretab, q.log_p_lay, q.T_prof_lay), float)
        data["mix.vmr_int.%d" % n] = np.array(hs.interpolate_grid_to_lay_or_int(
            q.log_kpress, q.ktemp, sp.vmr_pretab, q.log_p_int, q.T_prof_int), float)
    data["mix.mu_lay"] = hs.calc_meanmolmass(q, type="layer")
    data["mix.mu_int"] = hs.calc_meanmolmass(q, type="interface")
    # local radiative-equilibrium check of the convection loop
    for tag, seed, limit in (("tight", 31, 1e-7), ("loose", 32, 1e-2)):
        q = radeq_state(seed, limit)
        data["radeq.%s.criterion" % tag] = np.array(hs.check_for_radiative_eq(q))
        data["radeq.%s.converged" % tag] = q.converged.copy()
        data["radeq.%s.marked_red" % tag] = q.marked_red.copy()
    # start-up: unit conversion, numerical limits, isothermal start, internal flux
    for tag, db, ts in (("hemi", 0, 5200.0), ("beam", 1, 3000.0), ("nostar", 0, 0.0)):
        q = start_state(db, ts)
        hs.planet_param(q, None)
        hs.set_up_numerical_parameters(q)
        hs.initial_temp(q, None)
        hs.calc_F_intern(q)
        data["start.%s" % tag] = np.array([q.g, q.a, q.R_planet, q.R_star, q.T_star, q.w_0_limit, q.w_0_scat_limit,
                                           q.delta_tau_limit, q.F_intern, q.T_lay[0]], float)
        data["start.%s.gauss_weight" % tag] = np.array(q.gauss_weight)
    # cloud pre-processing: Mie tables -> size distribution -> wavelength bins -> decks (source/clouds.py)
    import contextlib
    import io
    import tempfile
    from source import clouds as ref_clouds
    from source import tools as ref_tools
    for tag in ("manual", "file"):
        with tempfile.TemporaryDirectory() as wd, contextlib.redirect_stdout(io.StringIO()):
            q = run_clouds(ref_clouds.Cloud(), tag, wd)
        for k in CLOUD_KEYS:
            data["cloud.%s.%s" % (tag, k)] = np.array(getattr(q, k), float)
    from source import read as ref_read2
    # on-the-fly species readers.  The reference opens its containers with h5py (not installed); for this generator
    # only, `h5py.File` is a thin read-only adapter over .npz files holding the same dataset names.
    class _NpzFile(object):
        def __init__(self, name, mode="r"):
            path = name if os.path.exists(name) else name[:-3] + ".npz"
            if not os.path.exists(path):
                raise IOError(name)
            self.d = dict(np.load(path))

        def __enter__(self):
            return self.d

        def __exit__(self, *a):
            return False
    sys.modules["h5py"].File = _NpzFile
    from source import quantities as ref_quant2
    with tempfile.TemporaryDirectory() as wd, contextlib.redirect_stdout(io.StringIO()):
        write_species_inputs(wd)
        rq, rr = ref_quant2.Store(), ref_read2.Read()
        species_reader_setup(rr, rq, wd)
        run_species_readers(rr, rq)
        for k, v in species_record(rq).items():
            data["species." + k] = v
    from source import species_database as ref_sdb
    data["speciesdb.names"] = np.array(list(ref_sdb.species_lib.keys()))
    data["speciesdb.fc"] = np.array([v.fc_name for v in ref_sdb.species_lib.values()])
    data["speciesdb.weight"] = np.array([v.weight for v in ref_sdb.species_lib.values()], float)
    # albedo file, T-P files, heating file, stellar spectrum, rocky-planet f approximation
    from source import additional_heating as ref_heat
    with tempfile.TemporaryDirectory() as wd, contextlib.redirect_stdout(io.StringIO()):
        write_misc_inputs(wd)
        for surface in ("Basaltic", "Granitoid"):
            q, r = misc_state(), ref_read2.Read()
            r.input_surf_albedo, r.albedo_file = "file", os.path.join(wd, "albedo.dat")
            r.albedo_file_header_lines, r.albedo_file_wavelength_name = 2, "Wavelength"
            r.albedo_file_wavelength_unit, r.albedo_file_surface_name = "micron", surface
            r.read_or_fill_surf_albedo_array(q)
            data["misc.albedo." + surface] = np.array(q.surf_albedo, float)
        for fmt, unit in (("helios", "[helios,"), ("TP", "bar"), ("PT", "cgs")):
            q, r = misc_state(), ref_read2.Read()
            r.temp_path, r.temp_format, r.temp_pressure_unit = os.path.join(wd, "tp_%s.dat" % fmt), fmt, unit
            r.read_temperature_file(q)
            data["misc.T_restart." + fmt] = np.array(q.T_restart, float)
        q = misc_state()
        q.add_heating_path = os.path.join(wd, "heating.txt")
        ref_heat.load_heating_terms_or_not(q)
        hs.calc_add_heating_flux(q)
        data["misc.add_heat_dens"] = np.array(q.add_heat_dens, float)
        data["misc.F_add_heat_lay"] = np.array(q.F_add_heat_lay, float)
        data["misc.F_add_heat_sum"] = np.array(q.F_add_heat_sum, float)
        q, r = misc_state(), ref_read2.Read()
        r.stellar_model, r.stellar_path, r.stellar_data_set = "file", os.path.join(wd, "star.h5"), "/grid/some_star"
        r.read_star(q)
        data["misc.starflux"] = np.array(q.starflux, float)
        q, r = misc_state(), types.SimpleNamespace(output_path=wd + "/")
        os.makedirs(os.path.join(wd, "rock"))
        hs.calc_tau_lw_sw(q, r)
        data["misc.tau_file"] = np.array(open(os.path.join(wd, "rock", "rock_tau_lw_tau_sw_f_factor.dat")).read())
        hs.approx_f_from_formula(q, r)
        data["misc.f_factor"] = np.array([q.tau_lw, q.f_factor], float)
    # photochemistry coupling: T-P hand-over files of three consecutive coupling steps and the convergence verdicts
    for full, speed in ((0, 1), (1, 0)):
        with tempfile.TemporaryDirectory() as wd, contextlib.redirect_stdout(io.StringIO()):
            texts = run_coupling(hs, ref_write.Write, wd, full, speed)
        data["coupling.full%d" % full] = np.array(texts)
    # kappa / c_p / entropy tables and the constant-kappa shortcut
    for mode in ("file", "water_atmo", "0.2857"):
        with tempfile.TemporaryDirectory() as wd, contextlib.redirect_stdout(io.StringIO()):
            path = os.path.join(wd, "delad.dat")
            write_kappa_file(path, mode == "water_atmo")
            q, r = kappa_state(mode, path)
            ref_read2.Read.read_kappa_table_or_use_constant_kappa(r, q)
        for k in KAPPA_KEYS:
            data["kappa.%s.%s" % (mode, k)] = np.array(getattr(q, k), float)
        data["kappa.%s.dims" % mode] = np.array([getattr(q, "entr_ntemp", 0) or 0, getattr(q, "entr_npress", 0) or 0], float)
    # the spectrum re-binning on its own (both interpolation types, table narrower / wider than the bins)
    rng = np.random.default_rng(77)
    old_l = np.sort(10.0 ** rng.uniform(-0.3, 1.7, 60))
    old_f = 10.0 ** rng.uniform(-3, 2, 60)
    new_l = 10.0 ** np.linspace(-0.6, 2.0, 23)
    data["rebin.old_lambda"], data["rebin.old_flux"], data["rebin.new_lambda"] = old_l, old_f, new_l
    for kind in ("linear", "log"):
        with contextlib.redirect_stdout(io.StringIO()):
            data["rebin." + kind] = np.array(ref_tools.convert_spectrum(old_l, old_f, new_l, type=kind), float)
            data["rebin.fine." + kind] = np.array(ref_tools.convert_spectrum(
                old_l, old_f, 10.0 ** np.linspace(0.0, 1.5, 400), type=kind), float)
    np.savez_compressed(os.path.join(HERE, "host_functions.npz"), **data)

    # writers: two states (non-isothermal with convection columns; isothermal, no star, no convection)
    out = os.path.join(HERE, "writer")
    for tag, kw in (("a", dict()), ("b", dict(seed=8, iso=1, T_star=0.0, convection=0, nbin=3, nlayer=4))):
        q = writer_state(**kw)
        q.name = "gold_" + tag
        run_writers(hs, ref_write.Write, q, out)
    n = sum(len(f) for _, _, f in os.walk(out))
    print("wrote host_functions.npz and %d writer files" % n)


if __name__ == "__main__":
    main()
