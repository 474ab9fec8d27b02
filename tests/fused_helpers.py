"""Runs the same problem through the fused fast path (RTBatch) and through the CPU oracle chain."""
import numpy as np

import cases

FUSED_KEYS = ["T_lay", "T_int", "F_up_band", "F_down_band", "F_dir_band", "F_up_tot", "F_down_tot",
              "F_net", "planckband_lay", "planckband_int", "opac_wg_lay", "opac_wg_int",
              "scat_cross_lay", "scat_cross_int", "meanmolmass_lay", "meanmolmass_int", "delta_z_lay",
              "z_lay", "F_up_wg", "Fc_up_wg", "F_down_wg", "Fc_down_wg", "abort", "delta_t_prefactor"]


def run_oracle(port, c0, n_iter, planck_grid=None, refresh=None):
    c = c0.copy()
    s = cases.alloc_state(c)
    if planck_grid is None:
        cases.setup_planck(port, c, s)
    else:
        s.planck_grid[:] = planck_grid
    cases.radiation_iterations(port, c, s, n_iter, refresh=refresh or cases.refresh_premixed)
    out = dict(s)
    out["T_lay"], out["T_int"], out["z_lay"] = c.T_lay, c.T_int, c.z_lay
    out["delta_t_prefactor"] = s.deltat_prefactor
    L = c.nlayer
    out["opac_wg_lay"] = s.opac_wg_lay[:c.ny * c.nbin * L]
    return out


def run_fused(ctx, c0, n_iter, ncol=1, keys=FUSED_KEYS, col=0, T_per_col=None, with_planck_grid=False):
    from helios_amd.rt import batch_from_case
    species = c0.get("species")
    rt = batch_from_case(ctx, c0, ncol=ncol, nspecies=len(species) if species else 0)
    if species:
        for k, sp in enumerate(species):
            rt.set_species(k, sp["pretab"], sp["scat"], sp["weight"], is_h2o=2 if sp["is_h2o"] else 0,
                           is_cia=1 if sp["is_cia"] else 0, in_mu=0 if sp["is_cia"] else 1)
        vl, vi = cases.species_vmr_arrays(c0)
        rt.set_column_vmr(-1, vl, vi)
    try:
        rt.keep_down_fluxes(True)
        if T_per_col is not None:
            for i, T in enumerate(T_per_col):
                rt.set_temperatures(i, T)
        rt.build_planck_table(1 if c0.T_star > 10 else 0)
        rt.run(0, n_iter)
        if isinstance(col, int):
            out = {k: rt.get(k, col) for k in keys}
        else:
            out = [{k: rt.get(k, cc) for k in keys} for cc in col]
        if with_planck_grid:
            return out, rt.get("planck_grid")
        return out
    finally:
        rt.close()


def keys_for(c, keys=FUSED_KEYS):
    """with isothermal layers the reference computes neither interface quantities nor layer-centre fluxes"""
    if not int(c.get("iso", 0)):
        return list(keys)
    return [k for k in keys if not (k.startswith("Fc_") or (k.endswith("_int") and k != "T_int"))]


def compare(f, o, c0, rtol=1e-8, keys=None):
    keys = keys_for(c0) if keys is None else keys_for(c0, keys)
    scale = max(np.abs(o["F_down_wg"]).max(), np.abs(o["F_dir_wg"]).max(), np.abs(o["F_up_wg"]).max())
    nwg = c0.ny * c0.nbin * c0.nlayer
    for k in keys:
        a, b = f[k], o[k]
        if k == "abort":
            assert np.array_equal(a, b), k
            continue
        if k in ("Fc_up_wg", "Fc_down_wg"):
            a, b = a[:nwg], b[:nwg]
        atol = 1e-300
        if k.startswith(("F_", "Fc_")):
            atol = 1e-90 + 1e-13 * scale
        if k == "F_net":
            atol = 1e-12 * np.abs(o["F_up_tot"]).max()
        if k.startswith("planckband"):
            atol = 1e-13 * np.abs(b).max()
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=k)
