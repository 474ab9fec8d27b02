"""Cloud pre-processing: Mie tables of one aerosol per particle radius -> size-distribution-weighted
cross-sections on the model's wavelength bins -> vertical decks -> the six arrays the hot path consumes
(`abs/scat_cross_all_clouds_{lay,int}`, `g_0_all_clouds_{lay,int}`) plus the total mixing ratio.

Counterpart of the reference's `Cloud` (source/clouds.py:27-253), same attribute names (they are filled by the
parameter reader).  Host-only, once per run.  Kept as in the reference on purpose: the asymmetry parameter of a deck
is weighted with the SCATTERING cross-section of each radius, not with g_0 itself (clouds.py:108),
so that cloudy runs reproduce the reference's fluxes.
"""
import numpy as np

from . import tools as tls

# the LX-MIE radius grid the reference hard-wires: 10^-2 ... 10^3 micron in steps of 0.1 dex (clouds.py:86-88)
R_VALUES = 10 ** np.arange(-2, 3.1, 0.1)
DELTA_R = R_VALUES * (10 ** 0.05 - 10 ** -0.05)


class Cloud(object):

    def __init__(self):
        self.nr_cloud_decks = 0
        self.mie_path = None
        self.cloud_r_mode = None
        self.cloud_r_std_dev = None
        self.cloud_mixing_ratio_setting = None
        self.cloud_vmr_file = None
        self.cloud_vmr_file_header_lines = None
        self.cloud_file_press_name = None
        self.cloud_file_press_units = None
        self.cloud_file_species_name = None
        self.p_cloud_bot = None
        self.f_cloud_bot = None
        self.cloud_to_gas_scale_height = None
        self.lamda_mie = None
        self.abs_cross_one_cloud = None
        self.scat_cross_one_cloud = None
        self.g_0_one_cloud = None
        self.f_one_cloud_lay = None
        self.f_one_cloud_int = None

    # ---- Mie tables --------------------------------------------------------------------------------------
    @staticmethod
    def read_mie_file(mie_file):
        """one LX-MIE output file: wavelength [micron -> cm], scattering and absorption cross-sections, g_0"""
        tab = np.loadtxt(mie_file, skiprows=1, usecols=(0, 3, 4, 6), ndmin=2)
        return list(tab[:, 0] * 1e-4), list(tab[:, 1]), list(tab[:, 2]), list(tab[:, 3])

    @staticmethod
    def lognorm_pdf(r, r_mode, sigma):
        r_median = r_mode / np.exp(-np.log(sigma) ** 2)
        return np.exp(-0.5 * (np.log(r / r_median) / np.log(sigma)) ** 2) / (r * np.log(sigma) * (2 * np.pi) ** 0.5)

    def calc_weighted_cross_sections_with_pdf_and_interpolate_wavelengths(self, nr, quant):
        pdf = self.lognorm_pdf(R_VALUES, self.cloud_r_mode[nr], self.cloud_r_std_dev[nr])
        per_r = [self.read_mie_file(self.mie_path[nr] + "r{:.6f}.dat".format(r)) for r in R_VALUES]
        self.lamda_mie = per_r[0][0]
        scat = np.array([t[1] for t in per_r])          # [radius][wavelength]
        absorb = np.array([t[2] for t in per_r])
        w = (pdf * DELTA_R)[:, None]
        weighted_abs = list((absorb * w).sum(axis=0))
        weighted_scat = list((scat * w).sum(axis=0))
        weighted_g_0 = list((scat * w).sum(axis=0))     # sic: see the module docstring
        self.abs_cross_one_cloud = tls.convert_spectrum(self.lamda_mie, weighted_abs, quant.opac_wave,
                                                        int_lambda=quant.opac_interwave, type="log")
        self.scat_cross_one_cloud = tls.convert_spectrum(self.lamda_mie, weighted_scat, quant.opac_wave,
                                                         int_lambda=quant.opac_interwave, type="log")
        self.g_0_one_cloud = tls.convert_spectrum(self.lamda_mie, weighted_g_0, quant.opac_wave,
                                                  int_lambda=quant.opac_interwave, type="linear")

    # ---- vertical distribution -----------------------------------------------------------------------------
    def create_cloud_deck(self, nr, quant):
        L, I = int(quant.nlayer), int(quant.ninterface)
        p_lay, p_int = np.asarray(quant.p_lay, float), np.asarray(quant.p_int, float)
        self.f_one_cloud_lay = np.zeros(L)
        self.f_one_cloud_int = np.zeros(I)
        if self.cloud_mixing_ratio_setting == "manual":
            # deck base in the layer that contains p_cloud_bot; above it a power law in pressure whose exponent is
            # set by the cloud-to-gas scale-height ratio (clouds.py:131-151)
            inside = np.nonzero((p_int[:-1] >= self.p_cloud_bot[nr]) & (self.p_cloud_bot[nr] > p_int[1:]))[0]
            i_bot = 0
            if len(inside):
                i_bot = int(inside[0])
                self.f_one_cloud_lay[i_bot] = self.f_cloud_bot[nr]
            expo = 1 / self.cloud_to_gas_scale_height[nr] - 1
            self.f_one_cloud_lay[i_bot + 1:] = self.f_cloud_bot[nr] * (p_lay[i_bot + 1:] / p_lay[i_bot]) ** expo
            if quant.iso == 0:
                self.f_one_cloud_int[i_bot + 1:] = self.f_cloud_bot[nr] * (p_int[i_bot + 1:] / p_lay[i_bot]) ** expo
        elif self.cloud_mixing_ratio_setting == "file":
            tab = np.genfromtxt(self.cloud_vmr_file, names=True, dtype=None, skip_header=self.cloud_vmr_file_header_lines)
            press = np.array(tab[self.cloud_file_press_name], float)
            press *= {"Pa": 10.0, "bar": 1e6}.get(self.cloud_file_press_units, 1.0)
            f_orig = np.array(tab[self.cloud_file_species_name[nr]], float)
            # linear in log10 p; beyond the file: its last value below the first pressure, its first above the last
            # (scipy interp1d fill_value=(f[-1], f[0]) in the reference, clouds.py:170)
            logp = np.log10(press)

            def on_profile(p):
                x = np.log10(np.asarray(p, float))
                order = np.argsort(logp)
                v = np.interp(x, logp[order], f_orig[order])
                v = np.where(x < logp.min(), f_orig[-1], v)
                return np.where(x > logp.max(), f_orig[0], v)
            self.f_one_cloud_lay = on_profile(p_lay)
            if quant.iso == 0:
                self.f_one_cloud_int = on_profile(p_int)

    def add_individual_cloud_decks_to_total(self, quant):
        a, s, g = (np.asarray(v, float) for v in (self.abs_cross_one_cloud, self.scat_cross_one_cloud, self.g_0_one_cloud))
        levels = [("lay", self.f_one_cloud_lay)] + ([("int", self.f_one_cloud_int)] if quant.iso == 0 else [])
        for name, f in levels:
            f = np.asarray(f, float)
            getattr(quant, "f_all_clouds_" + name)[:] += f
            getattr(quant, "abs_cross_all_clouds_" + name)[:] += np.outer(f, a).reshape(-1)
            getattr(quant, "scat_cross_all_clouds_" + name)[:] += np.outer(f, s).reshape(-1)
            getattr(quant, "g_0_all_clouds_" + name)[:] += np.outer(f, g * s).reshape(-1)   # scattering-weighted

    @staticmethod
    def normalize_g_0(quant):
        for name in ("lay",) + (("int",) if quant.iso == 0 else ()):
            sc = getattr(quant, "scat_cross_all_clouds_" + name)
            g0 = getattr(quant, "g_0_all_clouds_" + name)
            pos = sc > 0
            g0[pos] /= sc[pos]

    def cloud_pre_processing(self, quant):
        X, L, I = int(quant.nbin), int(quant.nlayer), int(quant.ninterface)
        quant.f_all_clouds_lay, quant.f_all_clouds_int = np.zeros(L), np.zeros(I)
        for stem in ("abs_cross_all_clouds", "scat_cross_all_clouds", "g_0_all_clouds"):
            setattr(quant, stem + "_lay", np.zeros(L * X))
            setattr(quant, stem + "_int", np.zeros(I * X))
        if quant.clouds == 1:
            for nr in range(int(self.nr_cloud_decks)):
                self.calc_weighted_cross_sections_with_pdf_and_interpolate_wavelengths(nr, quant)
                self.create_cloud_deck(nr, quant)
                self.add_individual_cloud_decks_to_total(quant)
            self.normalize_g_0(quant)
