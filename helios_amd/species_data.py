"""Species table of the on-the-fly mixing mode: for every opacity source its name in the species file, its name in
FastChem's output columns (two names joined by '&' for pair processes: the product of both abundances is used) and
its molar weight [g/mol].  Same content as the reference's source/species_database.py (checked entry by entry in
tests/test_host_golden.py); `None` = FastChem does not provide the species.
"""


class SpeciesEntry(object):
    __slots__ = ("name", "fc_name", "weight")

    def __init__(self, name, fc_name, weight):
        self.name, self.fc_name, self.weight = name, fc_name, weight


_TABLE = """
CO2 C1O2 44.01 | H2O H2O1 18.0153 | CO C1O1 28.01 | O2 O2 31.9988 | CH4 C1H4 16.04 | HCN C1H1N1 27.0253
NH3 H3N1 17.031 | H2S H2S1 34.081 | PH3 H3P1 33.99758 | O3 O3 47.9982 | O3_IR O3 47.9982 | O3_UV O3 47.9982
NO N1O1 30.01 | SO2 O2S1 64.066 | SH H1S1 33.073 | H2 H2 2.01588 | N2 N2 28.0134 | SO O1S1 48.0644
OH H1O1 17.007 | COS C1O1S1 60.0751 | CS C1S1 44.0757 | HCHO H2C1O1 30.02598 | C2H4 C2H4 28.05316
C2H2 C2H2 26.04 | CH3 C1H3 37.04004 | C3H C3H1 37.04004 | C2H C2H1 25.02934 | C2N2 C2N2 52.0348
C3O2 C3O2 68.0309 | C4N2 C4N2 76.0562 | C3 C3 36.0321 | S2 S2 64.13 | S3 S3 96.195 | S2O O1S2 80.1294
CS2 C1S2 76.1407 | NO2 N1O2 46.0055 | N2O N2O1 44.013 | HNO3 H1N1O3 63.01 | SO3 O3S1 80.066
H2SO4 H2O4S1 98.0785 | TiO O1Ti1 63.866 | TiH - 48.87 | VO O1V1 66.9409 | SiO O1Si1 44.08 | AlO Al1O1 42.98
CaO Ca1O1 56.0774 | PO O1P1 46.97316 | SiH H1Si1 29.09344 | CaH Ca1H1 41.085899 | AlH Al1H1 27.9889
MgH H1Mg1 25.3129 | CrH Cr1H1 53.004 | NaH H1Na1 23.99771
H H 1.007825 | He He 4.0026 | C C 12.0096 | N N 14.007 | O O 15.999 | F F 18.9984 | Na Na 22.989769
Ne Ne 20.1797 | Ni Ni 58.6934 | Mg Mg 24.305 | Mn Mn 54.938044 | Al Al 26.9815385 | Ar Ar 39.948
Si Si 28.085 | P P 30.973761998 | S S 32.06 | Cl Cl 35.45 | K K 39.0983 | Ca Ca 40.078 | Ti Ti 47.867
V V 50.9415 | Co Co 58.933194 | Cr Cr 51.9961 | Cu Cu 63.546 | Fe Fe 55.845 | Zn Zn 65.38
H-_bf H1- 1.007825 | H-_ff H&e- 1.007825 | He- He&e- 4.0026 | H3+ - 3.02382 | HeH+ - 5.01054
Fe+ Fe1+ 55.845 | Ti+ Ti1+ 47.867 | e- e- 0.00054858
CIA_H2H2 H2&H2 2.01588 | CIA_H2He H2&He 4.0026 | CIA_CO2CO2 C1O2&C1O2 44.01 | CIA_O2CO2 O2&C1O2 44.01
CIA_O2O2 O2&O2 31.9988 | CIA_O2N2 O2&N2 28.0134 | CIA_N2N2 N2&N2 28.0134 | CIA_N2H2 N2&H2 2.01588
"""

species_lib = {}
for _item in _TABLE.replace("\n", " | ").split("|"):
    _f = _item.split()
    if _f:
        species_lib[_f[0]] = SpeciesEntry(_f[0], None if _f[1] == "-" else _f[1], float(_f[2]))
