#!/usr/bin/env python3
"""Resource usage of every kernel in libhelios_hip.so, read from the code objects' own notes (amdhsa metadata):
name, VGPRs, spilled VGPRs, scratch bytes per lane, SGPRs, LDS.

    python tools/code_object_notes.py [path/to/lib.so] [--filter k_rt_flux]

The gfx950 code objects sit in the library's .hip_fatbin section as clang offload bundles; each is cut out and handed to
llvm-readelf --notes.  tests/test_abi.py uses `kernel_notes()` to hold the fused kernels to "no scratch"."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(path):
    out = subprocess.run([READELF, "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out)
    if not m:
        raise RuntimeError("no .hip_fatbin section in %s" % path)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(path, "rb") as f:
        f.seek(off)
        return f.read(size)


def code_objects(path):
    """the device ELF images (bytes) of every offload bundle in the library"""
    blob = _fatbin(path)
    pos, out = 0, []
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, blob[pos + off:pos + off + size]))
        pos += len(MAGIC)
    return out


def kernel_notes(path=None):
    """[{name, vgpr_count, vgpr_spill_count, sgpr_count, sgpr_spill_count, private_segment_fixed_size,
    group_segment_fixed_size}] for every kernel of the library (names demangled)"""
    path = path or os.path.join(ROOT, "helios_amd", "libhelios_hip.so")
    kernels = []
    for _triple, elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(elf)
            tf.flush()
            txt = subprocess.run([READELF, "--notes", tf.name], capture_output=True, text=True, check=True).stdout
        for b in txt.split("  - .agpr_count:")[1:]:
            k = {}
            for key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                        "private_segment_fixed_size", "group_segment_fixed_size"):
                k[key] = int(re.search(r"\.%s:\s+(\d+)" % key, b).group(1))
            k["symbol"] = re.search(r"\.name:\s+(\S+)", b).group(1)
            kernels.append(k)
    names = subprocess.run(["c++filt"], input="\n".join(k["symbol"] for k in kernels), capture_output=True,
                           text=True).stdout.split("\n")
    for k, n in zip(kernels, names):
        k["name"] = n
    return kernels


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def load_destinations(path=None, name_filter="k_rt_flux"):
    """{demangled kernel name: {destination register (pair): number of global loads into it}} from the disassembly of
    the library's gfx950 code objects.  Many loads into ONE register pair are loads the compiler serialised -- each has
    to be waited for before the next can be issued (round 4: the direct-beam planes of k_rt_flux)"""
    from collections import Counter
    path = path or os.path.join(ROOT, "helios_amd", "libhelios_hip.so")
    out = {}
    for _triple, elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(elf)
            tf.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", tf.name], capture_output=True, text=True,
                                 check=True).stdout
        for block in re.split(r"\n(?=[0-9a-f]+ <)", txt):
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", block)
            if not m:
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            if name_filter not in name:
                continue
            out[name] = dict(Counter(re.findall(r"global_load_dword(?:x2|x3|x4)?\s+(v\[?\d+(?::\d+)?\]?),", block)))
    return out


def disassembly(path=None, name_filter=""):
    """{demangled kernel name: [instruction lines]} of the library's gfx950 code objects"""
    path = path or os.path.join(ROOT, "helios_amd", "libhelios_hip.so")
    out = {}
    for _triple, elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(elf)
            tf.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", tf.name], capture_output=True, text=True,
                                 check=True).stdout
        for block in re.split(r"\n(?=[0-9a-f]+ <)", txt):
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", block)
            if not m:
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            if name_filter in name:
                out[name] = [ln.split("//")[0].strip() for ln in block.split("\n")[1:] if ln.strip()]
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    if flt in args:
        args.remove(flt)
    for k in sorted(kernel_notes(args[0] if args else None), key=lambda k: k["name"]):
        if flt in k["name"]:
            print("%-90s vgpr %3d spilled %2d scratch %4d B  sgpr %3d  lds %6d" %
                  (k["name"][:90], k["vgpr_count"], k["vgpr_spill_count"], k["private_segment_fixed_size"],
                   k["sgpr_count"], k["group_segment_fixed_size"]))
