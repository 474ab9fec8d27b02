"""TEST INFRASTRUCTURE: the reference's radiation_loop control flow (source/computation.py:827-990)
restated around the stage functions of tests/cases.py, so that the SAME loop can be run through the
reference's own kernels (oracle.refgpu on the MI355X, oracle.ref on the host), through the C
restatement (oracle.port) and through the per-stage HIP entry points.  SURVEY.md section 8(c): "oracle
loop driver ... goldens {T_lay, F_net, F_up_band[TOA], F_down_band[BOA], iter_count} after N = 1, 10,
11, 50 iterations and at convergence".

Control flow followed line by line:
    :851-853  while condition1 and condition2 (condition3 belongs to physical time-stepping)
    :856-857  temperatures + Planck every iteration
    :860-879  opacities / transmission / heights / beam when iter % 10 == 0
    :880-883  3*scat+1 sweeps,  :888 quadrature
    :906      temperature step only once iter >= foreplay
    :927-932  abortsum = sum(abort);  :938 condition1 = abortsum < nlayer + 1
    :941-943  time-stepped runs: condition3 = (iter + 1) * physical_tstep < runtime_limit
    :946-952  every 100th iteration: condition2 = T_lay[nlayer] < dim*step - 2
    :954      iter += 1
    :974-975  rad_convergence_limit *= 10 when the NEW iter value is in crit_relaxation_numbers
    :978-981  abort when iter > max_nr_iterations
"""
import numpy as np

import cases

SNAP_KEYS = ("F_net", "F_up_tot", "F_down_tot")


def snapshot(c, s):
    X, L, I = c.nbin, c.nlayer, c.ninterface
    out = {k: s[k].copy() for k in SNAP_KEYS}
    out["T_lay"] = c.T_lay.copy()
    out["F_up_band_TOA"] = s.F_up_band[X * (I - 1):X * I].copy()
    out["F_down_band_BOA"] = s.F_down_band[:X].copy()
    out["F_dir_band_BOA"] = s.F_dir_band[:X].copy()
    out["abort"] = s.abort.copy()
    out["deltat_prefactor"] = s.deltat_prefactor.copy()
    return out


def radiation_loop(impl, c, s, snap_at=(), max_nr_iterations=20000, crit_relaxation_numbers=(),
                   refresh=cases.refresh_premixed, on_iteration=None, runtime_limit=None):
    """runs the loop until the reference's loop would leave it.  Returns (iter_count, snaps, reason)
    with snaps[n] = state after n completed iterations for n in snap_at, snaps['end'] = final state."""
    snaps = {}
    it = 0
    L = c.nlayer
    condition1 = condition2 = condition3 = True
    reason = "converged"
    while condition1 and condition2 and condition3:
        cases.interpolate_temperatures_and_planck(impl, c, s)
        if it % 10 == 0:
            refresh(impl, c, s)
        cases.flux_sweeps(impl, c, s)
        cases.integrate_and_step(impl, c, s, it)
        abortsum = int(s.abort.sum()) if it >= c.foreplay else 0
        condition1 = abortsum < L + 1
        if c.physical_tstep != 0 and runtime_limit is not None:      # :941-943, with the iteration index before the increment
            condition3 = (it + 1) * c.physical_tstep < runtime_limit
            if not condition3:
                reason = "runtime limit"
        if it % 100 == 0:
            condition2 = bool(c.T_lay[L] < c.plancktable_dim * c.plancktable_step - 2)
            if not condition2:
                reason = "surface temperature beyond the Planck table"
        it += 1
        if it in snap_at:
            snaps[it] = snapshot(c, s)
        if on_iteration is not None:
            on_iteration(it, c, s)
        if it in crit_relaxation_numbers:
            c.rad_convergence_limit *= 10.0
        if it > max_nr_iterations:
            reason = "iteration limit"
            break
    snaps["end"] = snapshot(c, s)
    return it, snaps, reason


def loop_case(name):
    """the small columns whose loops are committed as goldens (tests/golden/loop_<name>.npz)"""
    kw = dict(nbin=6, nlayer=8, ntemp=6, npress=5, plancktable_dim=400, plancktable_step=10)
    relax = ()
    if name == "default":
        pass
    elif name == "dirbeam_albedo":
        kw.update(dir_beam=1, albedo=0.3)
    elif name == "noscat_relax":
        # criterion relaxed at iteration 60 (computation.py:974-975): the loop must end because of it
        kw.update(scat=0)
        relax = (60,)
    elif name == "clouds_g0_i2s":
        kw.update(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0)
    elif name == "onthefly":
        # opacities mixed on the fly at every refresh: 4 absorbers (one CIA pair), water-vapour and H2 scattering
        kw.update(nbin=5)
    else:
        raise KeyError(name)
    c = cases.make_case(**kw)
    if name == "noscat_relax":
        c.rad_convergence_limit = 1e-9
    if name == "onthefly":
        cases.add_species(c, nspecies=4)
    return c, relax


def loop_refresh(c):
    return cases.refresh_onthefly if c.get("species") else cases.refresh_premixed


LOOP_NAMES = ("default", "dirbeam_albedo", "noscat_relax", "clouds_g0_i2s", "onthefly")
SNAP_AT = (1, 10, 11, 50)
