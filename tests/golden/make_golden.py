#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN PYTHON (read-only, /root/reference).

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_golden.py
Outputs (committed): tests/golden/signal.npz, tests/golden/cases.json

What is pinned here:
  * bin/savitzky_golay.py + bin/call_peaks.py  -> smoothed tracks and peak indices
  * scipy.signal.find_peaks micro-cases (as installed here)
  * C3POa.rounding, the shift/clip/split block of C3POa.analyze_reads (C3POa.py:127-155)
  * bin/consensus.py pairwise_consensus / normalizeLen
  * determine_consensus dispatcher: subread naming, FASTQ/FASTA text, racon argv (stubs for
    mappy/pyabpoa/conk supply canned results: control flow only, not dependency arithmetic)
Nothing from /root/reference is copied: only inputs we generate and the outputs it computes.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

# numpy>=1.24 removed np.int / np.mat which bin/savitzky_golay.py:19-20,30 uses
np.int = int
np.mat = np.asmatrix

sys.path.insert(0, os.path.join(REF, "bin"))
from call_peaks import call_peaks            # noqa: E402
from savitzky_golay import savitzky_golay    # noqa: E402
from consensus import pairwise_consensus, normalizeLen  # noqa: E402
from scipy.signal import find_peaks          # noqa: E402

from c3poa_amd import synth                  # noqa: E402
from c3poa_amd.seqio import revcomp          # noqa: E402
from oracle import oracle_py as O            # noqa: E402  (only to make realistic integer tracks)


def make_tracks():
    tracks, names = [], []
    # realistic score tracks (integer) from synthetic reads of three shapes
    for cfg, n in (("cfg1", 4), ("cfg3", 3), ("cfg4", 1)):
        for rec in synth.generate(cfg, n_reads=n):
            sp = synth.SPLINT1 if rec[3] == "+" else revcomp(synth.SPLINT1)
            tracks.append(O.conk(sp, rec[1]).astype(np.int64))
            names.append("%s_%s" % (cfg, rec[0]))
    rng = np.random.default_rng(7)
    # pure noise (gated -> [])
    tracks.append(np.abs(rng.normal(1000, 300, 3000)).astype(np.int64)); names.append("noise")
    # noise + 4 gaussian bumps
    x = np.arange(5000)
    y = np.abs(rng.normal(10, 3, 5000))
    for c in (400, 1900, 3400, 4900):
        y += 400 * np.exp(-0.5 * ((x - c) / 15.0) ** 2)
    tracks.append(y.astype(np.int64)); names.append("bumps")
    # short track, plateau-heavy track
    tracks.append((rng.integers(0, 5, 200) * 100).astype(np.int64)); names.append("short_plateau")
    z = np.zeros(1500, dtype=np.int64); z[300:310] = 5000; z[900:905] = 7000; z[1200] = 9000
    tracks.append(z); names.append("spikes")
    return names, tracks


def main():
    out = {}
    names, tracks = make_tracks()
    npz = {}
    sig = []
    for nm, t in zip(names, tracks):
        for md in (500, 1500, 100):
            pk = call_peaks(t, md, 3, 41, 2)
            sig.append(dict(name=nm, min_dist=md, peaks=[int(p) for p in pk]))
        sm1 = savitzky_golay(t, 41, 2)
        sm3 = savitzky_golay(savitzky_golay(sm1, 41, 2), 41, 2)
        npz["track_" + nm] = t.astype(np.int32)
        if nm in ("cfg1_r00000000", "cfg3_r00000001", "bumps", "short_plateau", "spikes"):
            npz["sg1_" + nm] = np.asarray(sm1, dtype=np.float64)
            npz["sg3_" + nm] = np.asarray(sm3, dtype=np.float64)
    out["call_peaks"] = sig
    np.savez_compressed(os.path.join(HERE, "signal.npz"), **npz)

    # scipy find_peaks micro-cases
    fp = []
    rng = np.random.default_rng(11)
    cases = [([0, 3, 0, 0, 3, 0, 0, 0, 5, 0], 3, 1), ([0, 3, 0, 0, 3, 0, 0, 0, 5, 0], 3, 4),
             ([0, 3, 0, 0, 3, 0, 0, 0, 5, 0], 3, 5), ([0, 1, 1, 1, 1, 0], 0.5, 1),
             ([0, 2, 2, 0, 2, 2, 2, 0], 1, 2), ([5, 4, 3, 4, 5], 0, 1), ([1, 2, 3, 4], 0, 1)]
    for _ in range(40):
        n = int(rng.integers(5, 120))
        # distinct heights: numpy's argsort is not stable, so equal-priority order inside
        # scipy's _select_by_peak_distance is implementation-defined for larger arrays;
        # plateaus (equal neighbours) are still exercised by repeating samples
        x = np.repeat(rng.permutation(n).astype(float), rng.integers(1, 3, n)).tolist()
        cases.append((x, float(rng.integers(0, n // 2 + 1)), int(rng.integers(1, 12))))
    for x, h, d in cases:
        pk, _ = find_peaks(np.asarray(x, dtype=float), distance=d, height=h)
        fp.append(dict(x=x, height=h, distance=d, peaks=[int(p) for p in pk]))
    out["find_peaks"] = fp

    # stubs so that C3POa.py / determine_consensus.py import
    stub = tempfile.mkdtemp(prefix="c3stub_")
    os.makedirs(os.path.join(stub, "conk"))
    open(os.path.join(stub, "conk", "__init__.py"), "w").write("from . import conk\n")
    open(os.path.join(stub, "conk", "conk.py"), "w").write("TRACK=None\ndef conk(a,b,p):\n    return TRACK\n")
    open(os.path.join(stub, "mappy.py"), "w").write(
        "class _Hit:\n"
        "    def __init__(s,**k): s.__dict__.update(k)\n"
        "HITS={}\n"
        "class Aligner:\n"
        "    def __init__(s, seq=None, preset=None, scoring=None): s.seq=seq\n"
        "    def map(s, q):\n"
        "        return [ _Hit(q_st=0,q_en=len(q),strand=1,ctg_len=len(s.seq),r_st=0,r_en=len(s.seq),mlen=len(q),blen=len(q),mapq=60) ]\n"
        "def fastx_read(path, read_comment=False):\n"
        "    name=None\n"
        "    for line in open(path):\n"
        "        line=line.rstrip()\n"
        "        if line.startswith('>'): name=line[1:]\n"
        "        elif name is not None and line: yield (name, line, None); name=None\n"
        "def revcomp(s):\n"
        "    return s.translate(str.maketrans('ACGT','TGCA'))[::-1]\n")
    open(os.path.join(stub, "pyabpoa.py"), "w").write(
        "CALLS=[]\nCANNED={}\n"
        "class _R:\n    pass\n"
        "class msa_aligner:\n"
        "    def __init__(s, match=2): s.match=match\n"
        "    def msa(s, seqs, out_cons, out_msa):\n"
        "        CALLS.append((len(seqs), out_cons, out_msa, s.match))\n"
        "        r=_R(); r.cons_seq=CANNED.get('cons', []) if seqs else []; r.msa_seq=CANNED.get('msa', []) if seqs else []\n"
        "        return r\n")
    sys.path.insert(0, stub)
    sys.argv = ["C3POa.py"]
    spec = importlib.util.spec_from_file_location("C3POa_ref", os.path.join(REF, "C3POa.py"))
    c3 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(c3)

    out["rounding"] = [dict(x=x, base=50, r=c3.rounding(x, 50)) for x in
                       list(range(0, 200, 1)) + [1474, 1475, 1476, 1524, 1525, 1526, 1574, 1575, 1576, 2025, 2075]]

    # split: drive C3POa.analyze_reads with a canned track; capture determine_consensus args
    import conk as conk_stub
    captured = []

    def capture(args, read, subreads, sub_qual, dang, dang_q, racon, tmp_dir, subread_file):
        captured.append(dict(sub_lens=[len(s) for s in subreads], dang_lens=[len(s) for s in dang],
                             subs=[read[1].find(s) for s in subreads]))
        return "", 0
    c3.determine_consensus = capture
    split_cases = []
    L = 5000
    seq = "".join("ACGT"[i % 4] if i % 7 else "G" for i in range(L))
    # make seq positions unique enough: use index-coded sequence instead
    rng2 = np.random.default_rng(5)
    seq = "".join("ACGT"[i] for i in rng2.integers(0, 4, L))
    qual = "I" * L
    args = types.SimpleNamespace(mdistcutoff=500, out_path=tempfile.mkdtemp(prefix="c3out_") + "/", zero=True)
    os.makedirs(args.out_path + "Splint1")
    splint = "A" * 284

    def run_split(raw_peaks):
        # build a track whose call_peaks result is exactly raw_peaks: bypass via monkeypatch
        captured.clear()
        c3.call_peaks = lambda scores, md, it, w, o: np.array(raw_peaks, dtype=np.int64) if raw_peaks else []
        conk_stub.conk.TRACK = [0] * L
        c3.analyze_reads(args, [("rd", seq, qual)], {"Splint1": [splint, splint]}, {"rd": ["Splint1", "+"]},
                         set(["Splint1"]), 1, "racon")
        return dict(raw_peaks=raw_peaks, S=284, L=L, called=bool(captured),
                    sub_lens=captured[0]["sub_lens"] if captured else None,
                    dang_lens=captured[0]["dang_lens"] if captured else None)
    for raw in ([108, 1608, 3108, 4608], [1408, 2908, 4408], [108, 1608, 2608, 4608], [858, 2858], [2358],
                [108, 1608, 3108, 4848], [], [4900], [4857, 4990], [10, 1510, 3010, 4510], [108, 1583, 3108, 4683],
                [0, 100], [108, 1133, 2608, 4608], [58, 1083, 2108, 3133, 4158]):
        split_cases.append(run_split(raw))
    for _ in range(60):
        k = int(rng2.integers(1, 9))
        raw = sorted(set(int(v) for v in rng2.integers(0, L, k)))
        split_cases.append(run_split(raw))
    out["split"] = split_cases

    # header formatting (C3POa.py:167-173)
    hdr = []
    for q, Lq, rep, cl in (("I" * 5000, 5000, 3, 1600), ("5" * 333 + "I" * 667, 1000, 2, 812), ("#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI" * 7, 273, 12, 99)):
        avg = round(sum([ord(x) - 33 for x in q]) / Lq, 2)
        hdr.append(dict(qual=q, L=Lq, repeats=rep, cons_len=cl,
                        header=">" + "nm" + "_" + "_".join([str(x) for x in [avg, Lq, rep, cl]])))
    out["header"] = hdr

    # pairwise consensus / normalizeLen on random 2-row MSAs
    pw = []
    rng3 = np.random.default_rng(3)
    fixed = [(["ACGT", "AGGT"], ["IIII", "5555"]), (["ACGT", "AGGT"], ["5555", "5555"]),
             (["AC--GT", "ACTTGT"], ["5555", "IIIIII"]), (["AC--GT", "ACTTGT"], ["IIII", "555555"]),
             (["--ACGT", "TTACGT"], ["5I5I", "IIIIII"]), (["ACGT--", "ACGTTT"], ["5I5I", "555555"]),
             (["ACGT--", "ACGTTT"], ["IIII", "555555"]), (["ACGTAC", "ACGTAC"], ["555555", "IIIIII"])]
    for rows, quals in fixed:
        subs = [r.replace("-", "") for r in rows]
        pw.append(dict(rows=rows, quals=quals, cons=pairwise_consensus(rows, subs, quals),
                       norm=[normalizeLen(rows[0], quals[0]), normalizeLen(rows[1], quals[1])]))
    for _ in range(120):
        n = int(rng3.integers(3, 60))
        a, b = [], []
        for _c in range(n):
            r = rng3.random()
            if r < 0.12:
                a.append("-"); b.append("ACGT"[rng3.integers(0, 4)])
            elif r < 0.24:
                b.append("-"); a.append("ACGT"[rng3.integers(0, 4)])
            elif r < 0.4:
                x = int(rng3.integers(0, 4)); a.append("ACGT"[x]); b.append("ACGT"[(x + int(rng3.integers(1, 4))) % 4])
            else:
                x = "ACGT"[rng3.integers(0, 4)]; a.append(x); b.append(x)
        rows = ["".join(a), "".join(b)]
        subs = [r.replace("-", "") for r in rows]
        if not subs[0] or not subs[1]:
            continue
        quals = ["".join(chr(33 + int(v)) for v in rng3.integers(2, 41, len(s))) for s in subs]
        pw.append(dict(rows=rows, quals=quals, cons=pairwise_consensus(rows, subs, quals),
                       norm=[normalizeLen(rows[0], quals[0]), normalizeLen(rows[1], quals[1])]))
    out["pairwise"] = pw

    # determine_consensus dispatcher: naming / file texts / racon argv with a fake racon
    sys.path.insert(0, os.path.join(REF, "bin"))
    import determine_consensus as dc
    import pyabpoa
    fake_racon = os.path.join(stub, "fake_racon.sh")
    open(fake_racon, "w").write("#!/bin/sh\necho \"$@\" > \"$(dirname $1)/argv.txt\"\ncat \"$3\"\n")
    os.chmod(fake_racon, 0o755)
    disp = []
    for nsub, ndang in ((3, 2), (1, 0), (2, 1), (0, 2), (0, 1), (1, 2), (4, 0)):
        td = tempfile.mkdtemp(prefix="c3dc_") + "/"
        subs = ["ACGTACGTAC" * 3 + "ACGT"[i % 4] for i in range(nsub)]
        sq = ["I" * len(s) for s in subs]
        dang = ["TTTTGGGGCC" + "A" * j for j in range(ndang)]
        dq = ["5" * len(s) for s in dang]
        pyabpoa.CALLS.clear()
        pyabpoa.CANNED["cons"] = ["ACGTACGTACACGTACGTACACGTACGTAC"]
        pyabpoa.CANNED["msa"] = [subs[0], subs[1]] if nsub >= 2 else []
        a2 = types.SimpleNamespace(mdistcutoff=5, zero=False)
        cons, rep = dc.determine_consensus(a2, ("rd", "N" * 100, "I" * 100), subs, sq, dang, dq, fake_racon, td, td + "subreads.fastq")
        files = sorted(os.listdir(td))
        disp.append(dict(nsub=nsub, ndang=ndang, cons=cons, repeats=rep,
                         subreads_fastq=open(td + "subreads.fastq").read() if os.path.exists(td + "subreads.fastq") else None,
                         argv=open(td + "argv.txt").read().replace(td, "<tmp>/") if os.path.exists(td + "argv.txt") else None,
                         files_left=files, msa_calls=[list(c) for c in pyabpoa.CALLS]))
    out["dispatch"] = disp

    json.dump(out, open(os.path.join(HERE, "cases.json"), "w"), indent=0)
    print("wrote", len(out["call_peaks"]), "call_peaks,", len(fp), "find_peaks,", len(split_cases), "split,", len(pw), "pairwise,", len(disp), "dispatch")


if __name__ == "__main__":
    main()
