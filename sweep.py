#!/usr/bin/env python3
"""Parameter sweep over independent atmosphere columns, batched on the device and sharded over the GPUs of a node.

    python sweep.py -sweep "internal_temperature=100,300,1000;f_factor=0.25,0.5" -opacity_mixing synthetic -name grid
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 sweep.py -sweep "..." ...

All other options are those of helios.py.  HELIOS_SWEEP_PARTITION = cyclic (default: columns dealt out in turn) | block |
dynamic[:chunk] (ranks claim chunks of columns from a shared work list as they retire the ones they hold).  Column k writes its files to <output>/<name>_<k>/; rank 0 also writes
<output>/<name>_sweep_spectra.npz with the emission spectra of all columns and the swept parameter values.
"""
import os
import sys

import numpy as np


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if "-sweep" not in argv:
        raise SystemExit("usage: sweep.py -sweep \"key=v1,v2;key2=...\" [helios.py options]")
    k = argv.index("-sweep")
    spec = argv[k + 1]
    base = argv[:k] + argv[k + 2:]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist, coll_device = None, "cpu"
    if world > 1:
        backend = os.environ.get("HELIOS_BENCH_BACKEND", "nccl")     # "gloo": all ranks on GPU 0 (single-GPU machines)
        device_index = local_rank if backend == "nccl" else 0
        # every rank like bench.py's: onto the host cores of its GPU's NUMA node before its first GPU call, the process group
        # with a deadline, a roll call and a first collective that must count every rank (helios_amd/parallel.py)
        from helios_amd import parallel
        parallel.bind_to_gpu_numa_node(device_index)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        parallel.init_process_group_checked(dist, backend, int(os.environ.get("RANK", "0")), world,
                                            device=torch.device("cuda", device_index),
                                            timeout_s=float(os.environ.get("HELIOS_INIT_TIMEOUT", "180")))
        if backend == "nccl":
            coll_device = "cuda"
            os.environ["HELIOS_DEVICE"] = str(local_rank)
    from helios_amd import sweep as sw
    overrides = sw.expand_sweep(spec)
    columns, spectra = sw.run_sweep(base, overrides, dist, coll_device)
    rank = dist.get_rank() if dist is not None else 0
    if rank == 0:
        # (with HELIOS_SWEEP_PARTITION=dynamic the other ranks may have retired every column before rank 0 claimed one)
        wavelength = columns[0].opac_wave if columns else sw.wavelength_grid(base, overrides[0])
        out_dir = None
        for i, a in enumerate(base):
            if a == "-output_directory":
                out_dir = base[i + 1]
        out_dir = out_dir or "./output/"
        os.makedirs(out_dir, exist_ok=True)
        keys = sorted({k_ for o in overrides for k_ in o})
        np.savez(os.path.join(out_dir, sw._base_name(base) + "_sweep_spectra.npz"), F_up_TOA=spectra,
                 wavelength=np.asarray(wavelength), **{"param_" + k_: np.array([str(o.get(k_, "")) for o in overrides])
                                                        for k_ in keys})
        print("\nSweep of %d columns on %d GPU(s) finished." % (len(overrides), world))
    if dist is not None:
        dist.destroy_process_group()
    return columns, spectra


if __name__ == "__main__":
    main()
