#!/usr/bin/env python3
"""Pin the four external stages against the REAL libraries, wherever they are installed.

The reference's arithmetic for these stages lives in dependencies that are neither vendored in the reference repository
nor installed in the build container (conk: setup.sh:11-14 unpinned HEAD; pyabpoa==1.0.5: setup.sh:8; mappy; racon:
setup.sh:16-23).  This script is the one command that turns "parity unpinned" into golden vectors the moment a container
has them:

    python tools/pin_external.py            # writes tests/golden/external_<stage>.json for every stage it can import/run

Inputs are generated here (c3poa_amd.synth, fixed seeds); outputs are whatever the real library computes at the
reference's own call shapes:
    conk     conk.conk(splint, seq, 20)                                   C3POa.py:123
    pyabpoa  poa.msa_aligner(match=5).msa(seqs, out_cons, out_msa)        bin/determine_consensus.py:30,34,43
    mappy    mm.Aligner(seq=cons, preset='map-ont').map(subread)          bin/determine_consensus.py:56,63-67
    racon    racon reads.fastq overlaps.paf draft.fasta -q 5 -t 1          bin/determine_consensus.py:87-99
tests/test_external_pins.py compares the oracle with every file that exists (and skips the stages that have none).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import synth  # noqa: E402
from c3poa_amd.seqio import revcomp  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _mut(rng, s, **kw):
    seq, q = synth._mutate(rng, np.frombuffer(s.encode(), dtype=np.uint8), **kw)
    return seq.decode(), q.decode()


def subread_sets(seed=11, n_sets=8):
    """(subreads, quals) sets shaped like the kept subreads of one R2C2 read: 2..7 noisy copies of a 300..1500-nt unit"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_sets):
        truth = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(300, 1500))))
        pairs = [_mut(rng, truth) for _ in range(2 + k % 6)]
        out.append({"truth": truth, "subs": [p[0] for p in pairs], "quals": [p[1] for p in pairs]})
    return out


def pin_conk():
    try:
        from conk import conk
    except Exception as e:                                              # noqa: BLE001
        return "conk not importable (%s)" % e
    cases = []
    for name, seq, _q, strand, _t in synth.generate("cfg1", n_reads=6):
        sp = synth.SPLINT1 if strand == "+" else revcomp(synth.SPLINT1)
        cases.append({"splint": sp, "seq": seq, "penalty": 20, "track": [int(x) for x in conk.conk(sp, seq, 20)]})
    json.dump({"stage": "conk", "call": "conk.conk(splint, seq, 20)", "cases": cases}, open(os.path.join(GOLD, "external_conk.json"), "w"))
    return "%d tracks" % len(cases)


def pin_pyabpoa():
    try:
        import pyabpoa as poa
    except Exception as e:                                              # noqa: BLE001
        return "pyabpoa not importable (%s)" % e
    cases = []
    for s in subread_sets():
        a = poa.msa_aligner(match=5)
        subs = s["subs"]
        if len(subs) == 2:
            res = a.msa(subs, out_cons=False, out_msa=True)
            cases.append({"subs": subs, "msa": list(res.msa_seq), "cons": None})
        else:
            res = a.msa(subs, out_cons=True, out_msa=True)
            cases.append({"subs": subs, "msa": list(res.msa_seq), "cons": res.cons_seq[0]})
    ver = getattr(poa, "__version__", "unknown")
    json.dump({"stage": "pyabpoa", "version": ver, "call": "msa_aligner(match=5).msa", "cases": cases}, open(os.path.join(GOLD, "external_pyabpoa.json"), "w"))
    return "%d MSAs (pyabpoa %s)" % (len(cases), ver)


def pin_mappy():
    try:
        import mappy as mm
    except Exception as e:                                              # noqa: BLE001
        return "mappy not importable (%s)" % e
    cases = []
    for s in subread_sets(seed=12, n_sets=4):
        al = mm.Aligner(seq=s["truth"], preset="map-ont")
        hits = []
        for sub in s["subs"]:
            hits.append([[h.q_st, h.q_en, h.strand, h.ctg_len, h.r_st, h.r_en, h.mlen, h.blen, h.mapq] for h in al.map(sub)])
        cases.append({"draft": s["truth"], "subs": s["subs"], "hits": hits})
    json.dump({"stage": "mappy", "version": getattr(mm, "__version__", "unknown"), "cases": cases}, open(os.path.join(GOLD, "external_mappy.json"), "w"))
    return "%d drafts" % len(cases)


def pin_racon():
    racon = shutil.which("racon")
    if not racon:
        return "racon not on PATH"
    try:
        import mappy as mm
        import pyabpoa as poa
    except Exception:                                                   # noqa: BLE001
        return "racon needs mappy (overlaps) and pyabpoa (the draft it polishes), as bin/determine_consensus.py does"
    cases = []
    for s in subread_sets(seed=13, n_sets=6):
        if len(s["subs"]) < 3:
            continue                                                    # the 2-subread draft is the pairwise merge, not abPOA's
        draft = poa.msa_aligner(match=5).msa(s["subs"], out_cons=True, out_msa=False).cons_seq[0]   # determine_consensus.py:43-47
        with tempfile.TemporaryDirectory() as td:
            fq, paf, fa = os.path.join(td, "r.fastq"), os.path.join(td, "o.paf"), os.path.join(td, "d.fasta")
            open(fa, "w").write(">cons\n%s\n" % draft)
            al = mm.Aligner(seq=draft, preset="map-ont")
            with open(fq, "w") as f, open(paf, "w") as p:
                for i, (sub, q) in enumerate(zip(s["subs"], s["quals"])):
                    f.write("@s_%d\n%s\n+\n%s\n" % (i + 1, sub, q))
                    for h in al.map(sub):                               # the PAF rows of bin/determine_consensus.py:63-67
                        p.write("s_%d\t%d\t%d\t%d\t%s\tcons\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
                            i + 1, len(sub), h.q_st, h.q_en, h.strand, h.ctg_len, h.r_st, h.r_en, h.mlen, h.blen, h.mapq))
            out = subprocess.run([racon, fq, paf, fa, "-q", "5", "-t", "1"], capture_output=True, text=True).stdout
            polished = "".join(out.splitlines()[1:]) if out.startswith(">") else ""
        cases.append({"draft": draft, "subs": s["subs"], "quals": s["quals"], "polished": polished})
    ver = subprocess.run([racon, "--version"], capture_output=True, text=True).stdout.strip()
    json.dump({"stage": "racon", "version": ver, "argv": "-q 5 -t 1", "cases": cases}, open(os.path.join(GOLD, "external_racon.json"), "w"))
    return "%d drafts (racon %s)" % (len(cases), ver)


if __name__ == "__main__":
    for name, fn in (("conk", pin_conk), ("pyabpoa", pin_pyabpoa), ("mappy", pin_mappy), ("racon", pin_racon)):
        print("%-8s %s" % (name, fn()))
