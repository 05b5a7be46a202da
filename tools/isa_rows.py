#!/usr/bin/env python3
"""Diagnostic: instruction counts of the row bodies of a kernel in a `hipcc --save-temps` assembly listing.
    python tools/isa_rows.py file.s [mangled kernel name] [depth]
Lists every basic block at loop depth >= `depth` (default 4 = k_poa's row loop) with its instruction mix; blocks with a full
set of scan steps (>= 10 DPP instructions) are the row bodies.  Used to see what an edit elsewhere did to the fast row
(register allocation: copies, lane spills of scalars) before spending GPU time on it."""
import re
import sys


def function(path, name):
    s = open(path).read().split('\n')
    st = [i for i, l in enumerate(s) if l.startswith(name + ':')][0]
    en = [i for i in range(st, len(s)) if s[i].strip().startswith('.Lfunc_end')][0]
    return s[st:en]


def blocks(f):
    out, cur = [], None
    for l in f:
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
        if m:
            d = re.search(r'Depth=(\d)', m.group(2))
            cur = {"label": m.group(1), "depth": int(d.group(1)) if d else 0, "ins": []}
            out.append(cur)
        elif cur is not None and l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur["ins"].append(re.sub(r'\s+;.*', '', l.strip()))
    return out


def stat(b):
    i = b["ins"]
    return {"n": len(i), "valu": sum(x.startswith('v_') for x in i), "salu": sum(x.startswith('s_') for x in i),
            "dpp": sum('row_shr' in x or 'row_bcast' in x or 'wave_sh' in x for x in i), "bperm": sum('ds_bpermute' in x for x in i),
            "lds": sum(x.startswith('ds_') for x in i), "vmem": sum(x.startswith(('global_', 'scratch_', 'buffer_', 'flat_')) for x in i),
            "scratch": sum(x.startswith('scratch_') for x in i), "mov": sum(x.startswith('v_mov_b32') for x in i),
            "lanespill": sum(x.startswith(('v_readlane_b32', 'v_writelane_b32')) for x in i)}


if __name__ == "__main__":
    path = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "_Z5k_poaILb0ELb1ELb0EEv7PoaArgs"
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    bl = blocks(function(path, name))
    tot = {}
    for b in bl:
        s = stat(b)
        if b["depth"] >= depth:
            for k, v in s.items():
                tot[k] = tot.get(k, 0) + v
        if s["dpp"] >= 10 or (b["depth"] >= depth and s["n"] >= 40):
            print("%-10s depth %d  %s" % (b["label"], b["depth"], "  ".join("%s %d" % kv for kv in s.items())))
    print("all blocks at depth >= %d: %s" % (depth, "  ".join("%s %d" % kv for kv in tot.items())))
