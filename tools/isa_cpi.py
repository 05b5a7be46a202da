#!/usr/bin/env python3
"""Issue cost of a kernel's vector instructions from ITS OWN instruction mix (round 6, review item 6).

`roofline.valu.busy_frac` used one constant (2.9 cycles per vector wave-instruction) for every kernel, which made k_conk fill
106 % of its issue cycles.  The cost table measured on this chip (profiles/r02_valu_cost_gfx950.txt, tools/ubench/valu_cost.hip)
has two classes -- 2.33 cycles (add / sub / logic / shift-right / mov / 16-bit VOP2 min-max-add-sub / fp32 add-mul-fma) and 4.18
cycles (DPP and SDWA forms, compares, v_cndmask, lane reads / writes, every three-operand VOP3, 32-bit and float min / max,
variable shift-left, packed math, fp64, anything that names an SGPR operand) -- so the mean cost of a kernel's instructions follows
from which of them it issues.  This tool compiles the kernel sources with --save-temps, classifies every vector instruction of
the kernel's function(s) and weights a basic block by 8^(loop depth) (the DP rows sit in the innermost loops; a static listing
has no execution counts), and writes the per-kernel mean:

    python tools/isa_cpi.py [--json out.json]

tools/pmc_sq.sh multiplies the counted vector instructions per SIMD cycle by it.  The figure is an ESTIMATE of the dynamic mix:
the tool prints both classes' shares so that a reader can see what it rests on."""
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "c3poa_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
FAST, SLOW, QUARTER, CNDMASK_VCC = 2.33, 4.18, 16.0, 2.8
# kernel -> (source file, regexes of the mangled functions that are the kernel and its out-of-line callees in the shipped instance)
KERNELS = {
    "k_poa": ("k_poa.hip", [r"^_Z5k_poaILb0ELb1ELb0EE"]),
    "k_poa_wide": ("k_poa.hip", [r"^_Z5k_poaILb0ELb1ELb1EE", r"^_Z14poa_align_callILb0ELb1ELb1EE"]),
    "k_window": ("k_polish.hip", [r"^_Z8k_windowILb0EE", r"^_Z13win_rows_bandILi\dELb0EE", r"^_Z8win_rowsILi\d+ELb0EE", r"^_Z18win_traceback_bandILb0EE"]),
    "k_prep": ("k_polish.hip", [r"^_Z6k_prep"]),
    "k_conk": ("k_conk.hip", [r"^_Z6k_conkILi5ELb0EE"]),
    "k_peaks": ("k_peaks.hip", [r"^_Z7k_peaks"]),
}
THREE_OP = ("v_max3", "v_min3", "v_med3", "v_add3", "v_bfe", "v_bfi", "v_lshl_or", "v_lshl_add", "v_add_lshl", "v_and_or", "v_or3", "v_xad",
            "v_perm", "v_mad", "v_alignbit", "v_alignbyte", "v_sad", "v_pk_", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64",
            "v_cvt", "v_mbcnt", "v_bcnt", "v_ffb", "v_add_co", "v_addc", "v_sub_co", "v_subb", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64",
            "v_mul_u32_u24", "v_mul_i32_i24", "v_mul_hi", "v_dot", "v_div", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_ldexp", "v_frexp",
            "v_trunc", "v_floor", "v_ceil", "v_rndne", "v_fract", "v_cmp", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")
MINMAX32 = re.compile(r"^v_(max|min)_(i32|u32|f32|f64)")


def cost(ins):
    op = ins.split()[0]
    if op.startswith("v_mul_lo_u32") or op.startswith("v_mul_lo_i32"):
        return QUARTER
    if "_dpp" in op or "_sdwa" in op or "row_shr" in ins or "row_bcast" in ins or "wave_sh" in ins or "quad_perm" in ins or "row_mask" in ins:
        return SLOW
    if op.startswith("v_cndmask_b32_e32"):                                             # (VCC form: the table's cmp + cndmask pair costs 7.0 -> 7.0 - 4.19)
        return CNDMASK_VCC
    if op.startswith(THREE_OP) or MINMAX32.match(op):
        return SLOW
    args = ins[len(op):]
    if op.startswith("v_lshlrev_b32") and re.match(r"\s*v\d+,\s*v\d+", args):          # variable shift-left (the shift amount is the first source)
        return SLOW
    srcs = args.split(",")[1:]                                                           # everything after the destination
    if any(re.match(r"\s*(s\d+|s\[|vcc|exec|m0|ttmp)", s) for s in srcs):
        return SLOW
    return FAST


def functions(path):
    out, cur = {}, None
    for l in open(path):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", l)
        if m:
            cur = m.group(1); out[cur] = []
        elif l.strip().startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            out[cur].append(l.rstrip("\n"))
    return out


# loop depth of the call site of an out-of-line function (its own listing starts again at depth 0)
CALL_DEPTH = [(r"^_Z13win_rows_band|^_Z8win_rows", 3), (r"^_Z18win_traceback_band", 3), (r"^_Z14poa_align_call", 2)]


def mix(funcs):
    """funcs: [(mangled name, lines)] -> (weighted mean cost, share of the slow class, vector instructions in the listing)"""
    wsum, csum, slow, n = 0.0, 0.0, 0.0, 0
    for fn, lines in funcs:
        base = max([d for p, d in CALL_DEPTH if re.search(p, fn)] or [0])
        a, b, c_, k = mix1(lines, base)
        wsum += a; csum += b; slow += c_; n += k
    return (csum / wsum if wsum else None, slow / wsum if wsum else None, n)


def mix1(lines, base):
    depth, wsum, csum, slow, n = 0, 0.0, 0.0, 0.0, 0
    head = False                                       # inside the comment lines that follow a block label
    for l in lines:
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            head = True
            d = re.search(r"(?:in Loop: Header=\S+|Loop Header:) Depth=(\d)", l); depth = int(d.group(1)) if d else 0
            continue
        if head and l.strip().startswith(";"):
            d = re.search(r"(?:in Loop: Header=\S+|This (?:Inner )?Loop Header:) Depth=(\d)", l)
            if d:
                depth = int(d.group(1))
            continue
        head = False
        s = l.strip()
        if not l.startswith("\t") or s.startswith((".", ";")) or not s.startswith("v_"):
            continue
        ins = re.sub(r"\s+;.*", "", s)
        if ins.split()[0] in ("v_nop",):
            continue
        w = 8.0 ** (base + depth)
        c = cost(ins)
        wsum += w; csum += w * c; slow += w * (c > FAST + 0.1); n += 1
    return wsum, csum, slow, n


def kernel_src_sha():
    hsh = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return hsh.hexdigest()[:16]


def main():
    out = {"note": "tools/isa_cpi.py: mean issue cycles per vector wave-instruction from the kernel's own listing (hipcc --save-temps), classes of "
                   "profiles/r02_valu_cost_gfx950.txt (%.2f / %.2f cycles; v_mul_lo_u32 %.0f), basic blocks weighted 8^(loop depth)" % (FAST, SLOW, QUARTER),
           "kernel_src_sha": kernel_src_sha(), "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        listings = {}
        for kn, (src, pats) in KERNELS.items():
            if src not in listings:
                subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-save-temps", "-c", os.path.join(CSRC, src), "-o", os.path.join(td, src + ".o")],
                                      cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                listings[src] = functions(glob.glob(os.path.join(td, src.replace(".hip", "") + "-hip-amdgcn-*.s"))[0])
            cpi, slow, n = mix([(fn, body) for fn, body in listings[src].items() if any(re.search(p, fn) for p in pats)])
            if cpi is None:
                continue
            out["kernels"][kn] = {"cycles_per_inst": round(cpi, 3), "slow_class_share": round(slow, 3), "vector_insts_in_listing": n}
            print("%-11s %.2f cycles per vector instruction (%.0f %% in the 4.2-cycle class; %d vector instructions in the listing)" % (kn, cpi, 100 * slow, n))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    return out


if __name__ == "__main__":
    main()
