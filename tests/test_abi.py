"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol the header
declares, its struct layouts match the ctypes mirrors, and the product never touches the oracle."""
import ctypes
import os
import re

from helios_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    protos = _lib.prototypes()
    assert len(protos) >= 65
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(raw, n)]
    assert not missing, missing


def test_header_covers_every_hot_path_kernel():
    """one hx_<kernel> per reference kernel on the hot path"""
    protos = _lib.prototypes()
    for k in ("plancktable", "corr_inc_energy", "temp_inter", "planck_interpol_layer",
              "planck_interpol_interface", "opac_interpol", "meanmolmass_interpol", "opac_species_interpol",
              "add_to_mixed_opac", "calc_h2o_scat", "add_to_mixed_scat", "calc_total_g_0_of_gas_and_clouds",
              "calc_trans_iso", "calc_trans_noniso", "calc_delta_z", "fdir_iso", "fdir_noniso", "fband_iso",
              "fband_noniso", "integrate_flux", "rad_temp_iter", "conv_temp_iter",
              "integrate_optdepth_transmission_iso", "integrate_optdepth_transmission_noniso",
              "calc_contr_func_iso", "calc_contr_func_noniso", "calc_mean_opacities", "integrate_beamflux"):
        assert "hx_" + k in protos, k


def test_struct_layouts_match():
    from helios_amd import rt
    from helios_amd.device import HxDiag
    rt._check_struct_sizes(_lib.lib())
    assert ctypes.sizeof(HxDiag) == 64          # 8 slots of 8 bytes (static_assert in context.hip)


def test_no_gpu_means_loud_failure():
    """there is no CPU fallback: creating a context without a GPU raises"""
    import pytest
    from helios_amd.device import Context
    raw = _lib.lib()
    n = ctypes.c_void_p()
    rc = raw.hx_create(0, ctypes.byref(n))
    if rc == 0:          # running on a GPU box
        raw.hx_destroy(n)
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HeliosHipError):
        Context(0)


def test_product_does_not_import_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle/|libhelios_oracle|libhelios_ref", re.M)
    for d, _dirs, files in os.walk(os.path.join(ROOT, "helios_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(d, f)).read()
                assert not pat.search(text), os.path.join(d, f)
    text = open(os.path.join(ROOT, "helios.py")).read()
    assert not pat.search(text)


def test_every_symbol_of_the_header_is_bound():
    """every hx_ name that appears as a function in include/helios_hip.h has a parsed prototype (pointer-returning ones
    included: hx_stream returns void*, which a missing prototype would truncate to a 32-bit int) and is exported"""
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    names = set(re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", text))
    protos = _lib.prototypes()
    assert names and names <= set(protos), sorted(names - set(protos))
    assert protos["hx_stream"][0] is ctypes.c_void_p
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n


def test_reference_gpu_build_exports_every_launcher():
    """oracle/_ref/libhelios_ref_gfx950.so (the reference's kernels.cu built by hipcc, the pin): one ref_<kernel> entry
    per launcher of the host build, same names"""
    path = os.path.join(ROOT, "oracle", "_ref", "libhelios_ref_gfx950.so")
    if not os.path.exists(path):
        import pytest
        pytest.skip("built where /root/reference exists")
    from helios_amd._cproto import parse_prototypes
    g = parse_prototypes(open(os.path.join(ROOT, "oracle", "ref_driver_gfx950.hip")).read(), "ref_")
    h = parse_prototypes(open(os.path.join(ROOT, "oracle", "ref_driver.cpp")).read(), "ref_")
    assert set(g) == set(h) and len(g) >= 34
    raw = ctypes.CDLL(path)
    for n in g:
        assert hasattr(raw, n), n


def _kernel_notes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_object_notes", os.path.join(ROOT, "tools", "code_object_notes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {k["name"]: k for k in mod.kernel_notes()}


def test_no_selected_flux_kernel_keeps_registers_in_scratch():
    """the tiling chosen for any column of 1 ... 416 layers (and any isothermal column of up to 512) is an instantiation
    of k_rt_flux whose code object reports no spilled VGPRs and no scratch (read from the library's own notes, no GPU
    needed); beyond 416 layers only k = 64 with 14-16 rows exists"""
    import ctypes
    from helios_amd import _lib
    lib = _lib.lib()
    notes = _kernel_notes()
    flux = {n: k for n, k in notes.items() if "k_rt_flux<" in n}
    assert len(flux) >= 4 * 16
    seen = set()
    k, r = ctypes.c_int(), ctypes.c_int()
    for iso, top in ((0, 416), (1, 512)):
        for beam in (0, 1):
            for L in range(1, top + 1):
                assert lib.hx_rt_flux_geometry(L, iso, beam, 20, 10000, 1, ctypes.byref(k), ctypes.byref(r)) == 0
                seen.add((r.value, k.value if k.value >= 16 else 0))
                if beam:        # (round 5: the beam no longer changes the choice -- its planes' rows are requested in groups)
                    kb, rb = ctypes.c_int(), ctypes.c_int()
                    assert lib.hx_rt_flux_geometry(L, iso, 0, 20, 10000, 1, ctypes.byref(kb), ctypes.byref(rb)) == 0
                    assert (kb.value, rb.value) == (k.value, r.value), (L, iso)
    for rows, K in sorted(seen):
        # the sweeps (`false`) and the direct solve of the matrix method (`true`) share the tiling and its selection
        for matrix in ("false", "true"):
            n = [v for name, v in flux.items() if "k_rt_flux<%d, %d, %s>" % (rows, K, matrix) in name]
            assert len(n) == 1, (rows, K, matrix)
            assert n[0]["vgpr_spill_count"] == 0 and n[0]["private_segment_fixed_size"] == 0, (rows, K, matrix, n[0])
    # and the instantiations that do spill are known to the selection
    spilling = {(int(name.split("<")[1].split(",")[0]), int(name.split(",")[1].split(">")[0]))
                for name, v in flux.items() if v["vgpr_spill_count"] > 0}
    assert spilling and not (spilling & seen)


def test_no_flux_kernel_serialises_its_tile_loads():
    """round 4 found the loads of the two direct-beam planes of k_rt_flux compiled into ONE register pair that is loaded,
    waited for and added once per row -- 14 to 26 dependent memory round trips per tile, 9 to 23 % of the kernel -- and
    whether the compiler did so depended on unrelated code.  The disassembly of the library's code objects is held to it:
    no instantiation loads more than six times into the same destination (rows requested in groups leave three to six)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_object_notes", os.path.join(ROOT, "tools", "code_object_notes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dest = mod.load_destinations(name_filter="k_rt_flux<")
    assert len(dest) >= 4 * 16
    worst = {n: max(c.values()) for n, c in dest.items() if c}
    assert len(worst) == len(dest)
    # (the tilings of 15 and 16 rows keep part of their register image in scratch and are never selected: not held to it)
    selectable = {n: w for n, w in worst.items() if int(n.split("<")[1].split(",")[0]) <= 14}
    assert max(selectable.values()) <= 6, sorted(selectable.items(), key=lambda kv: -kv[1])[:5]
    # (round 6: the tilings of 20-32 rows -- columns of 513-1024 layers, 64 lanes only -- live partly in scratch and reload
    # through few registers by construction; they are what such a column gets instead of the per-stage path, not held to it)
    assert max(w for n, w in worst.items() if int(n.split("<")[1].split(",")[0]) <= 16) <= 8
    # rows x 7 tile loads (three to six coefficient planes, the state) are all there
    rows = int([n for n in dest if "k_rt_flux<13, 16, false>" in n][0].split("<")[1].split(",")[0])
    assert sum(dest[[n for n in dest if "k_rt_flux<13, 16, false>" in n][0]].values()) >= 7 * rows


def test_no_coef_kernel_serialises_its_beam_loads():
    """the same pattern in the refresh's coefficient kernel (round 4, found in the listing of k_rt_coef<7, 8>: ten loads of
    the beam values into one register pair, seven into another): the beam is now read once per NODE -- a row's top value
    is the next row's bottom value -- one row of arithmetic ahead of its use.  Every instantiation: at most eight loads
    into one destination (the unrolled tilings share address registers, not results), config 5's tiling at most six."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_object_notes", os.path.join(ROOT, "tools", "code_object_notes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dest = mod.load_destinations(name_filter="k_rt_coef<")
    assert len(dest) >= 16 * 4
    worst = {n: max(c.values()) for n, c in dest.items() if c}
    assert len(worst) == len(dest)
    assert max(worst.values()) <= 8, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    assert worst[[n for n in worst if "k_rt_coef<7, 8>" in n][0]] <= 6


def test_only_the_unselected_flux_tilings_use_scratch():
    """every other kernel of the library -- the species loop with random overlap, the coefficient kernel, all per-stage
    kernels -- runs without a private segment (k_rt_mix_species kept 7 VGPRs in scratch in round 2: 2.1 GB of stores per
    launch); the mixing kernel has to leave room for FIVE wavefronts per SIMD, twenty per CU (round 6: at most 96 VGPRs --
    the allocation granule is 8, 512 / 96 = 5 -- and 160 KB / 20 = 8 KB of LDS; rounds 2-5: 128 VGPRs, 10 KB, four per SIMD),
    and so does the per-stage random-overlap kernel built on the same device function"""
    notes = _kernel_notes()
    with_scratch = sorted(n for n, k in notes.items() if k["private_segment_fixed_size"] > 0 or k["vgpr_spill_count"] > 0)
    assert with_scratch and all("k_rt_flux<" in n for n in with_scratch), with_scratch
    mix = [k for n, k in notes.items() if "k_rt_mix_species" in n]
    assert len(mix) == 1 and mix[0]["vgpr_count"] <= 96 and mix[0]["group_segment_fixed_size"] <= 8192, mix
    lean = [k for n, k in notes.items() if "k_add_to_mixed_opac_lean" in n]
    assert len(lean) == 1 and lean[0]["vgpr_count"] <= 96 and lean[0]["group_segment_fixed_size"] <= 8192, lean


def test_the_planck_tables_hand_issued_scalar_load_is_left_alone_until_its_wait():
    """csrc/stage_interp.hip requests a term's constants with an inline `s_load_dwordx16` a term ahead and waits for them
    with an inline `s_waitcnt lgkmcnt(0)`: the compiler does not know that a load is outstanding in between.  The built code
    object is held to what the source relies on -- between every such load and the next wait on the scalar-memory counter,
    no instruction reads or writes one of its sixteen destination registers (an SGPR copy or spill there would see the
    registers before the data; the advisor's finding of round 5)"""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("code_object_notes", os.path.join(ROOT, "tools", "code_object_notes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dis = mod.disassembly(name_filter="k_plancktable")
    assert dis
    checked = 0
    for name, ins in dis.items():
        for n, line in enumerate(ins):
            m = re.match(r"s_load_dwordx16 s\[(\d+):(\d+)\]", line)
            if not m:
                continue
            lo, hi = int(m.group(1)), int(m.group(2))
            for later in ins[n + 1:]:
                if later.startswith("s_waitcnt") and "lgkmcnt(0)" in later:
                    break
                assert not later.startswith(("s_endpgm", "s_branch", "s_cbranch")) or True
                used = [int(x) for x in re.findall(r"\bs(\d+)\b", later)]
                for a, b in re.findall(r"s\[(\d+):(\d+)\]", later):
                    used += list(range(int(a), int(b) + 1))
                assert not [u for u in used if lo <= u <= hi], (name, line, later)
            else:
                raise AssertionError("no wait behind %s in %s" % (line, name))
            checked += 1
    assert checked >= 2
