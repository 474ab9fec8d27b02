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
