"""Whole radiation loops and tiling-sized columns: the CPU oracle against the vectors the reference's
own kernels produced on the MI355X (tests/golden/loop_*.npz, big_*.npz; SURVEY.md 8(c): "after N = 1,
10, 11, 50 iterations and at convergence")."""
import numpy as np
import pytest

import cases
import golden_checks as gc
import loop_driver as ld


@pytest.mark.parametrize("name", ld.LOOP_NAMES)
def test_oracle_loop_golden(port, name):
    gc.check_loop(lambda c, s, relax: ld.radiation_loop(port, c, s, ld.SNAP_AT, crit_relaxation_numbers=relax,
                                                        refresh=ld.loop_refresh(c))[:2], name)


@pytest.mark.parametrize("name", gc.BIG_NAMES)
def test_oracle_big_golden(port, name):
    def run(c, planck_grid, n_iter, state):
        if state is None:
            s = cases.alloc_state(c)
            s.planck_grid[:] = planck_grid
            cases.radiation_iterations(port, c, s, n_iter)
        else:
            s = state
            cases.radiation_iterations(port, c, s, n_iter - 1, start=1)
        out = dict(s)
        out.update(T_lay=c.T_lay, T_int=c.T_int, z_lay=c.z_lay)
        return out, s
    gc.check_big(run, name)


def test_loop_fixtures_cover_the_survey_list():
    z = np.load(gc.os.path.join(gc.GOLDEN, "loop_default.npz"))
    for at in ("it1", "it10", "it11", "it50", "end"):
        assert at + ".T_lay" in z.files
    assert "gfx950" in str(z["meta"])


@pytest.mark.parametrize("name", ld.CONV_NAMES + ("deep_hostref",))
def test_oracle_convection_loop_golden(port, name, capsys):
    """radiation loop + convection loop (source/computation.py:992-1174 restated in tests/loop_driver.py) through the C
    oracle and helios_amd/host_functions.py against the loops of the reference's kernels on the MI355X
    (loopconv_<name>.npz) and against the loop the reference's own Python host functions drove in the build container
    (loopconv_deep_hostref.npz): same iteration counts, same layer flags, T / fluxes / spectrum within 1e-6"""
    from helios_amd import host_functions as hs

    def run(c, s, kappa, radiative_first):
        n_rad, n, snaps, _q = ld.rad_conv_run(port, hs, c, s, kappa, radiative_first)
        return n_rad, n, snaps
    z = gc.check_loopconv(run, name)
    assert ("imported" in str(z["meta"])) == name.endswith("hostref")


def test_convection_loop_fixtures_cover_the_verdict_list():
    for name in ld.CONV_NAMES:
        z = np.load(gc.os.path.join(gc.GOLDEN, "loopconv_%s.npz" % name), allow_pickle=False)
        for at in ("start", "it1", "it10", "it11", "it50", "it400", "end"):
            for k in ("T_lay", "F_net", "F_up_band_TOA", "conv_layer", "marked_red"):
                assert "%s.%s" % (at, k) in z.files
        assert "gfx950" in str(z["meta"]) and int(z["iter_count"]) >= 400
        assert name != "c5physics_onthefly" or "species.0.pretab" in z.files
    z = np.load(gc.os.path.join(gc.GOLDEN, "loopconv_detached.npz"))
    zones = "".join(str(int(v)) for v in z["it10.conv_layer"])
    assert "10" in zones.strip("0")                       # two zones with a radiative hole between them
