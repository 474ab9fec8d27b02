"""The extended-precision referee of the matrix method (tests/matrix_referee.py) against the oracle, on the CPU: the
referee restates the reference's tridiagonal system and its Thomas elimination (source/kernels.cu:1864-1967, :2109-2284) in x87
long double, so the oracle -- pinned to the reference's golden vectors (tests/test_golden.py::test_oracle_matrix_golden) --
must sit on it in its well-conditioned half (up-fluxes, 1e-9) and near it in the other (down-fluxes: the elimination's own
rounding noise, see the module's header)."""
import numpy as np
import pytest

import cases
import fused_helpers as fh
import matrix_referee

CONFIGS = {
    "default": dict(albedo=0.1),
    "noscat": dict(scat=0, albedo=0.1),
    "dirbeam": dict(dir_beam=1, albedo=0.3),
    "clouds_g0": dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2),
    "L100": dict(nbin=24, nlayer=100, albedo=0.1),
    "iso": dict(iso=1, nbin=11, nlayer=20, albedo=0.1),
    "iso_clouds_beam": dict(iso=1, nbin=9, nlayer=37, clouds=1, scat_corr=1, g_0=0.2, dir_beam=1, albedo=0.15),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_sits_on_the_extended_precision_solution(port, name):
    c0 = cases.make_case(**CONFIGS[name])
    c0.flux_calc_method = "matrix"
    o = fh.run_oracle(port, c0, 1)
    # (the oracle stands in for the library: its up-fluxes at the tolerance of the GPU tests, the referee's own values for the
    # down-fluxes; compare_first_solve then asserts the oracle's distance from the referee)
    f = dict(o, **{k: v for k, v in matrix_referee.exact_fluxes(c0, o).items() if "down" in k})
    matrix_referee.compare_first_solve(fh, f, o, c0, 1e-9)


def test_the_reference_elimination_loses_digits_where_little_is_reflected(port):
    """what the referee is for, as a number: with the albedo at the reader's floor (1e-8, source/read.py:1261) the first
    row of the elimination divides by it, and the reference's down-flux at the surface is eight digits worse than its
    up-flux -- against the same extended-precision solution"""
    c0 = cases.make_case(albedo=1e-8, nlayer=40)
    c0.flux_calc_method = "matrix"
    o = fh.run_oracle(port, c0, 1)
    ex = matrix_referee.exact_fluxes(c0, o)
    nc = c0.nbin * c0.ny
    trig = np.asarray(o["scat_trigger"]) == 1
    up = np.abs(o["F_up_wg"][:nc] / ex["F_up_wg"][:nc] - 1.0)[trig].max()
    down = np.abs(o["F_down_wg"][:nc] / ex["F_down_wg"][:nc] - 1.0)[trig].max()
    assert up < 1e-12 and 1e-10 < down < 1e-5, (up, down)
