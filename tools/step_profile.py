"""Per-kernel HIP-event averages of one fused iteration (tuning aid).

    python tools/step_profile.py [--workload c2] [--steps 9]

Prints the average duration of every profiled scope of hx_rt_step over `steps` E-iterations (no refresh)
and of one refresh.  Environment knobs of rt_fused.hip (HELIOS_RT_*) apply."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--columns", type=int, default=1)
    ap.add_argument("--nbin", type=int, default=0, help="override the workload's number of bins")
    ap.add_argument("--nlayer", type=int, default=0, help="override the workload's number of layers")
    ap.add_argument("--convection", action="store_true", help="also time iterations of the convection loop")
    args = ap.parse_args()
    import bench
    from helios_amd.device import Context
    w = dict(bench.WORKLOADS[args.workload])
    if args.nbin:
        w["nbin"] = args.nbin
    if args.nlayer:
        w["nlayer"] = args.nlayer
    c = bench.build_case(w, 20242)
    ctx = Context(0)
    rt = bench.make_batch(ctx, c, args.columns)
    rt.build_planck_table(1)
    rt.run(0, 11)
    ctx.synchronize()
    rt.profile(True)
    rt.run(11, min(args.steps, 9))
    rt.profile(False)
    out = []
    for k in ("rt_nodes", "rt_flux", "rt_totals_a", "rt_totals_b"):
        ms, n = rt.profile_read(k)
        out.append("%s %.1f us (n=%d)" % (k, ms * 1e3, n))
    ctx.timer_start()
    rt.run(21, 9)
    ev = ctx.timer_stop_ms() / 9
    rt.profile(True)
    rt.refresh()
    rt.profile(False)
    for k in ("refresh_total", "opac_interpol", "rt_coef", "add_to_mixed_opac", "opac_species_interpol"):
        ms, n = rt.profile_read(k)
        if n:
            out.append("%s %.1f us (n=%d)" % (k, ms * 1e3, n))
    print("step(no refresh) %.1f us | " % (ev * 1e3) + " | ".join(out))
    if args.convection:
        import numpy as np
        from helios_amd import phys_const as pc
        L = c.nlayer
        kap = 2.0 / 7.0
        T = 2500.0 * (np.asarray(c.p_lay) / c.p_lay[0]) ** 0.4          # super-adiabatic below, like a hot interior
        T = np.maximum(T, 600.0)
        rt.set_temperatures(0, np.append(T, 2600.0))
        for name, v in (("kappa_lay", np.full(L, kap)), ("kappa_int", np.full(L + 1, kap)),
                        ("c_p_lay", np.full(L, pc.R_UNIV / kap))):
            rt.set_state(0, name, v)
        rt.set_state(0, "conv_layer", np.zeros(L + 1, np.int32))
        rt.set_state(0, "conv_unstable", np.zeros(L + 1, np.int32))
        rt.set_state(0, "dampara", np.array([-1.0]))
        rt.set_state(0, "done", np.zeros(1, np.int32))
        rt.conv_run(0, 11)
        ctx.synchronize()
        rt.profile(True)
        rt.conv_run(11, 9)
        rt.profile(False)
        out = []
        for k in ("rt_conv_adjust", "rt_nodes", "rt_flux", "rt_totals_a", "rt_totals_c"):
            ms, n = rt.profile_read(k)
            out.append("%s %.1f us (n=%d)" % (k, ms * 1e3, n))
        ctx.timer_start()
        rt.conv_run(21, 9)
        ev = ctx.timer_stop_ms() / 9
        print("convection step(no refresh) %.1f us | " % (ev * 1e3) + " | ".join(out),
              "| convective layers:", int(rt.get("conv_layer").sum()), "done:", int(rt.get("done")[0]))
    rt.close()


if __name__ == "__main__":
    main()
