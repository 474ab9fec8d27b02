#!/usr/bin/env python3
"""VGPR liveness over a hipcc -S listing (gfx950): where a kernel's register pressure peaks.

    hipcc -O3 --offload-arch=gfx950 -gline-tables-only --save-temps -c file.hip
    python tools/vgpr_liveness.py file-hip-amdgcn-amd-amdhsa-gfx950.s k_rt_mix_species [--top 15]

Builds the control-flow graph of the kernel from its labels and branches, runs the usual backward data-flow over the
vector registers (a write under a partial EXEC mask is counted as a definition: an upper bound on kills, a lower bound on
pressure), and prints the live count at the start of every phase marker (`; RO_MARK`), the instructions with the highest
pressure and the source lines (.loc) they belong to."""
import re
import sys

NO_DST = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "s_", "v_cmp_", "v_cmpx_",
          "v_readlane", "v_readfirstlane", "v_nop", "ds_nop", "global_atomic_add_u64", "global_atomic_add_x2", "ds_bpermute_b32__never")
TWO_DST = ("v_swap_b32",)


PARTIAL_KILLS = "--partial-kills" in sys.argv      # lower bound: every write ends a live range


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def parse(path, flt):
    t = open(path).read()
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(flt), t, re.M)
    body = t[m.end():t.find(".Lfunc_end", m.end())].split("\n")      # (a kernel may hold several s_endpgm)
    ins, labels, loc, mark, depth = [], {}, None, None, 0
    files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', t)) or dict(re.findall(r'\.file\s+(\d+)\s+"([^"]+)"', t))
    for l in body:
        s = l.strip()
        if not s:
            continue
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if mm:
            loc = (files.get(mm.group(1), mm.group(1)).split("/")[-1], int(mm.group(2)))
            continue
        mm = re.search(r"; RO_MARK (\w+)", s)
        if mm:
            mark = mm.group(1)
            ins.append(dict(op="mark", text=s, defs=[], uses=[], loc=loc, mark=mark, target=None))
            continue
        mm = re.match(r"([.\w$]+):", s)
        if mm:
            labels[mm.group(1)] = len(ins)
            continue
        if s.startswith((";", ".")):
            continue
        s = s.split(";")[0].strip()
        op, _, rest = s.partition(" ")
        toks = [x for x in re.split(r",\s*", rest) if x]
        vs = [regs(x.split(" ")[0]) for x in toks]
        defs, uses = [], []
        if op.startswith(NO_DST) and not op.startswith("s_"):
            for v in vs:
                uses += v
        elif op.startswith("s_"):
            pass
        elif op in TWO_DST:
            defs = vs[0] + vs[1]
            uses = vs[0] + vs[1]
        else:
            if vs:
                defs = vs[0]
                for v in vs[1:]:
                    uses += v
            # read-modify-write forms: DPP with old value, v_mac/v_fmac, *_sdwa preserve, d16 loads, dst also a source
            if "dpp" in op or "row_" in rest or "quad_perm" in rest or op.startswith(("v_fmac", "v_mac", "v_movrel", "v_writelane")):
                uses += defs
        # EXEC nesting in layout order (structured control flow): inside a saveexec region a vector write leaves the other
        # lanes' old value alive -- it does not end the live range
        if op.startswith(("s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec")):
            depth += 1
        elif op.startswith("s_or_b64") and toks and toks[0].strip() == "exec" and depth > 0:
            depth -= 1
        if depth > 0 and not PARTIAL_KILLS:
            uses = uses + defs
        target = None
        if op.startswith(("s_cbranch", "s_branch")):
            target = toks[-1].strip()
        ins.append(dict(op=op, text=s, defs=defs, uses=uses, loc=loc, mark=mark, target=target))
    return ins, labels


def liveness(ins, labels):
    n = len(ins)
    succ = []
    for i, x in enumerate(ins):
        s = []
        if x["op"] not in ("s_branch", "s_endpgm") and i + 1 < n:
            s.append(i + 1)
        if x["target"] in labels:
            s.append(labels[x["target"]])
        succ.append(s)
    live_in = [frozenset()] * n
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = frozenset((out - set(ins[i]["defs"])) | set(ins[i]["uses"]))
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    return live_in


def main():
    path, flt = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    ins, labels = parse(path, flt)
    live = liveness(ins, labels)
    print("instructions %d, max live VGPRs %d" % (len(ins), max(len(v) for v in live)))
    for i, x in enumerate(ins):
        if x["op"] == "mark":
            print("  mark %-12s live %3d" % (x["mark"], len(live[i])))
    # pressure by source line: the maximum over the instructions of a line
    by = {}
    for i, x in enumerate(ins):
        k = x["loc"]
        if k and len(live[i]) > by.get(k, (0, 0))[0]:
            by[k] = (len(live[i]), i)
    for k, (v, i) in sorted(by.items(), key=lambda kv: -kv[1][0])[:top]:
        print("  %3d live at %s:%d   %s" % (v, k[0], k[1], ins[i]["text"][:70]))


if __name__ == "__main__":
    main()
