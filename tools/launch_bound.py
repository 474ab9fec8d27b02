"""Is the iteration loop bound by kernel launches on the host?  Times the ENQUEUE of nine iterations (hx_rt_run
returns when the kernels are queued) against their execution on the GPU, for a small and for the headline grid.

    python tools/launch_bound.py [--workload c1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c1")
    args = ap.parse_args()
    import bench
    from helios_amd.device import Context
    c = bench.build_case(dict(bench.WORKLOADS[args.workload]), 20242)
    ctx = Context(0)
    rt = bench.make_batch(ctx, c, 1)
    rt.build_planck_table(1)
    rt.run(0, 11)
    ctx.synchronize()
    enq, tot = [], []
    for rep in range(20):
        t0 = time.perf_counter()
        rt.run(11 + 10 * rep, 9)            # iterations 11..19 of a decade: no refresh, no read-back inside
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) / 9)
        tot.append((t2 - t0) / 9)
    enq.sort(); tot.sort()
    print("%s: enqueue %.1f us per iteration (4 launches), queued + executed %.1f us per iteration (medians of 20)"
          % (args.workload, 1e6 * enq[len(enq) // 2], 1e6 * tot[len(tot) // 2]))


if __name__ == "__main__":
    main()
