"""Physical constants in cgs units.

Host-side counterpart of the reference's source/phys_const.py:27-44, which takes its values from
whatever `astropy.constants` version is installed (unpinned by the reference).  The values below
are the CODATA-2018 / IAU-2015 figures astropy >= 4.0 returns -- bit for bit what the reference's phys_const.py
holds under astropy 4.3.1 (tests/golden/reader/hdf5/expected_constants.json, made by tests/golden/make_hdf5_golden.py;
tests/test_read_hdf5.py::test_constants_are_astropys).  Device-side constants live in
helios_amd/csrc/hx_common.h and follow source/kernels.cu:36-41.
"""

C = 2.99792458e10                # speed of light, cm / s
K_B = 1.380649e-16               # Boltzmann constant, erg / K
H = 6.62607015e-27               # Planck constant, erg s
R_UNIV = 8.31446261815324e7      # universal gas constant, erg / mol / K
N_A = 6.02214076e23              # Avogadro's number, 1 / mol
SIGMA_SB = 5.6703744191844314e-05  # Stefan-Boltzmann constant, erg / cm2 / s / K4
AU = 1.495978707e13              # astronomical unit, cm
AMU = 1.6605390666e-24           # atomic mass unit, g
R_SUN = 6.957e10                 # nominal solar radius, cm
R_JUP = 7.1492e9                 # nominal equatorial Jupiter radius, cm
R_EARTH = 6.3781e8               # nominal equatorial Earth radius, cm
G = 6.674299999999999e-08        # gravitational constant, cgs (astropy's `.cgs` of 6.6743e-11 SI, one ulp below 6.6743e-8)
