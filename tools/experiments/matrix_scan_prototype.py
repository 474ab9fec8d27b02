"""Round 5 experiment: numpy prototype of the scan form of the matrix method against the oracle's Thomas
elimination and against the same recurrences in extended precision (CPU; run from the repo root: python tools/experiments/matrix_scan_prototype.py)"""
import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases, oracle
from helios_amd import phys_const as pc
port = oracle.port

def E_factor(w0, g0, scat_corr, i2s):
    E = np.ones_like(w0)
    m = (scat_corr == 1) & (w0 > i2s) & (g0 >= 0.0)
    Ev = np.maximum(1.0, 1.225 - 0.1582 * g0 - 0.1777 * w0 - 0.07465 * g0 * g0 + 0.2351 * w0 * g0 - 0.05582 * w0 * w0)
    return np.where(m, Ev, E)

def solve(c, s):
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    nc = X * Y; H = 2 * L
    def half(u, l):   # [H][nc]: even h = lower half of layer h/2, odd = upper
        out = np.empty((H, nc)); out[0::2] = l.reshape(-1, nc)[:L]; out[1::2] = u.reshape(-1, nc)[:L]; return out
    M = half(s.M_upper, s.M_lower); N = half(s.N_upper, s.N_lower); P = half(s.P_upper, s.P_lower)
    w0 = half(s.w_0_upper, s.w_0_lower); dt = half(s.delta_tau_wg_upper, s.delta_tau_wg_lower)
    tr = half(s.trans_wg_upper, s.trans_wg_lower)
    Gp = half(s.G_plus_upper, s.G_plus_lower); Gm = half(s.G_minus_upper, s.G_minus_lower)
    dtc_u = np.repeat(s.delta_tau_all_clouds_upper.reshape(L, X), Y, axis=1); dtc_l = np.repeat(s.delta_tau_all_clouds_lower.reshape(L, X), Y, axis=1)
    dtau = dt.copy(); dtau[0::2] += dtc_l; dtau[1::2] += dtc_u
    g0 = np.full((H, nc), c.g_0)
    if c.clouds == 1:
        gl = np.repeat(s.g_0_tot_lay.reshape(L, X), Y, axis=1); gi = np.repeat(s.g_0_tot_int.reshape(I, X), Y, axis=1)
        g0[0::2] = (gi[:L] + gl) / 2; g0[1::2] = (gi[1:] + gl) / 2
    E = E_factor(w0, g0, c.scat_corr, c.i2s_transition)
    Bl = np.repeat(s.planckband_lay.reshape(X, L + 2), Y, axis=0).T   # [L+2][nc]
    Bi = np.repeat(s.planckband_int.reshape(X, I), Y, axis=0).T      # [I][nc]
    Bn = np.empty((H + 1, nc)); Bn[0::2] = Bi; Bn[1::2] = Bl[:L]
    Fd = s.F_dir_wg.reshape(I, nc); Fc = s.Fc_dir_wg.reshape(-1, nc)[:L]
    Fn = np.empty((H + 1, nc)); Fn[0::2] = Fd; Fn[1::2] = Fc
    trig = np.repeat(s.scat_trigger.reshape(1, nc), 1, axis=0)[0] == 1
    Bb, Bt = Bn[:-1], Bn[1:]
    nmu = -c.mu_star
    # trigger branch coefficients
    K = 2 * np.pi * c.epsi * (1 - w0) / (E - w0)
    thin = dtau < c.delta_tau_limit
    with np.errstate(all="ignore"):
        pgrad = (Bb - Bt) / dtau
        pd = np.where(thin, (N + M - P) * (Bb + Bt) / 2, (M + N) * Bb - P * Bt + c.epsi / (E * (1 - w0 * g0)) * (P - M + N) * pgrad)
        pu = np.where(thin, (N + M - P) * (Bb + Bt) / 2, (M + N) * Bt - P * Bb + c.epsi / (E * (1 - w0 * g0)) * (M - N - P) * pgrad)
    dd = np.minimum(0, Fn[:-1] / nmu * (Gm * M + Gp * N) - Fn[1:] / nmu * P * Gm) if c.dir_beam else 0 * M
    du = np.minimum(0, Fn[1:] / nmu * (Gm * N + Gp * M) - Fn[:-1] / nmu * P * Gp) if c.dir_beam else 0 * M
    al = P / M; be = -N / M; sd = (K * pd + dd) / M; su = (K * pu + du) / M
    # pure absorption coefficients
    with np.errstate(all="ignore"):
        g = c.epsi * (tr - 1) / dtau
        up_ = np.where(thin, np.pi * c.epsi * (1 - tr), 2 * np.pi * c.epsi * (1 + g))
        vp_ = np.where(thin, np.pi * c.epsi * (1 - tr), 2 * np.pi * c.epsi * (-tr - g))
    al = np.where(trig, al, tr); be = np.where(trig, be, 0.0)
    sd = np.where(trig, sd, up_ * Bb + vp_ * Bt); su = np.where(trig, su, up_ * Bt + vp_ * Bb)
    A = np.repeat(c.surf_albedo, Y)
    boaK = np.where(trig, (1 - w0[0]) / (E[0] - w0[0]), 1.0)
    Bsurf = Bl[L + 1]
    rho = np.empty((H + 1, nc)); sig = np.empty((H + 1, nc)); inv = np.empty((H, nc))
    rho[0] = A; sig[0] = A * Fn[0] + (1 - A) * np.pi * boaK * Bsurf
    for h in range(H):
        inv[h] = 1.0 / (1.0 - be[h] * rho[h])
        aa = al[h] * inv[h]
        rho[h + 1] = be[h] + aa * al[h] * rho[h]
        sig[h + 1] = aa * sig[h] + (su[h] + aa * rho[h] * sd[h])
    D = np.empty((H + 1, nc)); U = np.empty((H + 1, nc))
    D[H] = (1 - c.dir_beam) * c.f_factor * (c.R_star / c.a) ** 2 * np.pi * Bl[L]
    U[H] = rho[H] * D[H] + sig[H]
    for h in range(H - 1, -1, -1):
        D[h] = (al[h] * D[h + 1] + (be[h] * sig[h] + sd[h])) * inv[h]
        U[h] = rho[h] * D[h] + sig[h]
    return D, U, trig

for name, kw in dict(default=dict(albedo=0.1), L100=dict(nbin=24, nlayer=100, albedo=0.1), dirbeam=dict(dir_beam=1, albedo=0.3),
                     clouds_g0=dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2), noscat=dict(scat=0, albedo=0.1),
                     tinyalb=dict(albedo=1e-8, nlayer=60)).items():
    c = cases.make_case(**kw); c.flux_calc_method = "matrix"
    s = cases.alloc_state(c); cases.setup_planck(port, c, s)
    cases.interpolate_temperatures_and_planck(port, c, s); cases.refresh_premixed(port, c, s); cases.flux_sweeps(port, c, s)
    D, U, trig = solve(c, s)
    nc = c.nbin * c.ny; L = c.nlayer; I = L + 1
    ref = dict(Fd=s.F_down_wg.reshape(I, nc), Fu=s.F_up_wg.reshape(I, nc), Fcd=s.Fc_down_wg.reshape(-1, nc)[:L], Fcu=s.Fc_up_wg.reshape(-1, nc)[:L])
    got = dict(Fd=D[0::2], Fu=U[0::2], Fcd=D[1::2], Fcu=U[1::2])
    scale = max(np.abs(v).max() for v in ref.values())
    out = []
    for k in ref:
        err = np.abs(got[k] - ref[k]); crit = err / (1e-13 * max(scale, np.abs(s.F_dir_wg).max()) + 1e-9 * np.abs(ref[k]))
        out.append("%s %.2f" % (k, crit.max()))
    print(name, "trigger %d/%d" % (trig.sum(), nc), " ".join(out), "neg:", int((D < 0).sum() + (U < 0).sum()))


# ---- who is closer to the exact solution of the reference's own system?  (extended precision, x87 long double)
def coefficient_rows(c, s):
    """al, be, sd, su [H][nc] and the boundary data exactly as solve() builds them, in float64"""
    import types
    out = {}
    def grab(**kw): out.update(kw)
    # re-run the first half of solve() -- kept in sync by construction: solve() is called with a hook
    return out

def exact_and_double(c, s):
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    D, U, trig = solve(c, s)                      # float64, stable form
    # exact: the same recurrences in long double from the same float64 coefficients
    import numpy as np
    global np_float
    return D, U, trig

LD = np.longdouble
def solve_ld(c, s):
    """solve() in long double: monkeypatch numpy float arrays -> longdouble by casting the inputs"""
    s2 = cases.Case(s)
    for k, v in list(s.items()):
        if isinstance(v, np.ndarray) and v.dtype == np.float64:
            s2[k] = v.astype(LD)
    c2 = cases.Case(c)
    c2.surf_albedo = np.asarray(c.surf_albedo).astype(LD)
    return solve(c2, s2)

print("---- distance to the extended-precision solution of the same equations (max over entries of |x - exact| / (1e-13 scale + 1e-9 |exact|))")
cfgs = dict(otf_cfg1=dict(nbin=14, nlayer=21, dir_beam=1, albedo=0.1, clouds=1, g_0=0.2, scat_corr=1),
            clouds_g0=dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2), default=dict(albedo=0.1))
for name, kw in cfgs.items():
    c = cases.make_case(**kw)
    refresh = cases.refresh_premixed
    if name.startswith("otf"):
        c = cases.add_species(c, nspecies=4); refresh = cases.refresh_onthefly
    c.flux_calc_method = "matrix"
    s = cases.alloc_state(c); cases.setup_planck(port, c, s)
    cases.interpolate_temperatures_and_planck(port, c, s); refresh(port, c, s); cases.flux_sweeps(port, c, s)
    D, U, trig = solve(c, s)
    Dx, Ux, _ = solve_ld(c, s)
    nc = c.nbin * c.ny; L = c.nlayer; I = L + 1
    ref = dict(Fd=s.F_down_wg.reshape(I, nc), Fu=s.F_up_wg.reshape(I, nc), Fcd=s.Fc_down_wg.reshape(-1, nc)[:L], Fcu=s.Fc_up_wg.reshape(-1, nc)[:L])
    got = dict(Fd=D[0::2], Fu=U[0::2], Fcd=D[1::2], Fcu=U[1::2])
    ex = dict(Fd=Dx[0::2], Fu=Ux[0::2], Fcd=Dx[1::2], Fcu=Ux[1::2])
    scale = max(float(np.abs(v).max()) for v in ref.values()); scale = max(scale, np.abs(s.F_dir_wg).max())
    row = []
    for k in ref:
        den = 1e-13 * scale + 1e-9 * np.abs(ex[k]).astype(float)
        e_ref = (np.abs(ref[k] - ex[k]).astype(float) / den).max(); e_new = (np.abs(got[k] - ex[k]).astype(float) / den).max()
        row.append("%s: reference %.2f scans %.4f" % (k, e_ref, e_new))
    print(name, "| ".join(row))
