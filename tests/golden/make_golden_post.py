#!/usr/bin/env python3
"""Golden vectors for the post-processing host logic, made by RUNNING THE REFERENCE'S OWN PYTHON
(/root/reference/C3POa_postprocessing.py: parse_blat, match_index, write_fasta_file).

Run in the build container only:  python tests/golden/make_golden_post.py   ->  tests/golden/post_cases.json
The reference imports two packages that are not installed here; they are stubbed with generic utilities only
(mappy.revcomp / mappy.fastx_read from c3poa_amd.seqio, editdistance.eval = textbook Levenshtein), so what is pinned is
the reference's control flow, coordinate arithmetic and record formats -- not blat (the PSL rows are inputs here).
Nothing from /root/reference is copied: inputs are generated, outputs are what the reference computes from them.
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
from c3poa_amd.seqio import fastx_read, revcomp  # noqa: E402


def _lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


mm = types.ModuleType("mappy"); mm.revcomp = revcomp; mm.fastx_read = lambda p, read_comment=False: fastx_read(p)
ed = types.ModuleType("editdistance"); ed.eval = _lev
sys.modules["mappy"] = mm; sys.modules["editdistance"] = ed
spec = importlib.util.spec_from_file_location("ref_post", os.path.join(REF, "C3POa_postprocessing.py"))
ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)

rng = np.random.default_rng(20)
rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))  # noqa: E731
A5, A3 = rnd(33), rnd(36)
INDEXES = [("idx%d" % k, rnd(16)) for k in range(4)]


def psl(read, rlen, adapter, alen, strand, matches, qins, qs, qe, ts, te):
    return "\t".join(str(c) for c in [matches, alen - matches, 0, 0, 1 if qins else 0, qins, 0, 0, strand, read, rlen, qs, qe,
                                      adapter, alen, ts, te, 1, "%d," % (qe - qs), "%d," % qs, "%d," % ts])


def make_reads(with_index):
    reads, rows = {}, []
    for i in range(40):
        cdna = rnd(int(rng.integers(300, 900)))
        flip = i % 2 == 1
        left, right = (A3, A5) if flip else (A5, A3)
        lname, rname = ("3Prime_adapter", "5Prime_adapter") if flip else ("5Prime_adapter", "3Prime_adapter")
        idx = INDEXES[i % 4][1] if with_index and i % 5 else rnd(16)
        body = (revcomp(idx) + cdna) if (with_index and flip) else (cdna + idx if with_index else cdna)
        pre, post = rnd(int(rng.integers(0, 60))), rnd(int(rng.integers(0, 60)))
        seq = pre + left + body + revcomp(right) + post
        name = "read%03d_12.5_4000_3_%d" % (i, len(seq))
        reads[name] = seq
        p0 = len(pre); m0 = len(pre) + len(left) + len(body)
        kind = i % 10
        if kind == 7:                                  # only one adapter found
            rows.append(psl(name, len(seq), lname, len(left), "+", len(left) - 1, 0, p0, p0 + len(left), 0, len(left)))
        elif kind == 8:                                # two '+' hits -> dropped
            rows.append(psl(name, len(seq), lname, len(left), "+", len(left), 0, p0, p0 + len(left), 0, len(left)))
            rows.append(psl(name, len(seq), rname, len(right), "+", 14, 0, p0 + 100, p0 + 114, 3, 17))
            rows.append(psl(name, len(seq), rname, len(right), "-", len(right), 0, m0, m0 + len(right), 0, len(right)))
        elif kind == 9:                                # weak / gappy rows are ignored, same adapter on both sides -> dropped
            rows.append(psl(name, len(seq), lname, len(left), "+", len(left), 0, p0, p0 + len(left), 0, len(left)))
            rows.append(psl(name, len(seq), lname, len(left), "-", len(left) - 2, 0, m0, m0 + len(right), 0, len(left)))
            rows.append(psl(name, len(seq), rname, len(right), "-", 9, 0, m0, m0 + 9, 0, 9))
            rows.append(psl(name, len(seq), rname, len(right), "-", 30, 55, m0, m0 + 30, 0, 30))
        else:                                          # partial alignments: positions are projected to the adapter ends
            ts = int(rng.integers(0, 4)); te = len(left) - int(rng.integers(0, 4))
            rows.append(psl(name, len(seq), lname, len(left), "+", te - ts - 1, 0, p0 + ts, p0 + te, ts, te))
            ts2 = int(rng.integers(0, 4)); te2 = len(right) - int(rng.integers(0, 4))
            rows.append(psl(name, len(seq), rname, len(right), "-", te2 - ts2, 1, m0 + (len(right) - te2), m0 + (len(right) - ts2), ts2, te2))
    return reads, rows


def run_ref(flags, reads, rows, with_index):
    with tempfile.TemporaryDirectory() as d:
        path = d + "/"
        with open(path + "adapter_to_consensus_alignment.psl", "w") as fh:
            fh.write("\n".join(rows) + "\n")
        idx_to_seq, seq_to_idx = ({n: s for n, s in INDEXES}, {s: n for n, s in INDEXES}) if with_index else ({}, {})
        args = types.SimpleNamespace(undirectional=flags.get("u", False), barcoded=flags.get("b", False), trim=flags.get("t", False), threads=2)
        ad = ref.parse_blat(path, reads)
        ref.write_fasta_file(args, path, ad, reads, seq_to_idx, idx_to_seq)
        out = {}
        for root, _dirs, files in os.walk(path):
            for f in files:
                if f.endswith(".psl"):
                    continue
                out[os.path.relpath(os.path.join(root, f), path)] = open(os.path.join(root, f)).read()
        return out


cases = []
for flags, with_index in (({}, False), ({"t": True}, False), ({"u": True, "t": True}, False), ({"b": True}, False),
                          ({"t": True}, True), ({}, True)):
    reads, rows = make_reads(with_index)
    cases.append({"flags": flags, "with_index": with_index, "reads": reads, "psl": rows,
                  "indexes": INDEXES if with_index else [], "files": run_ref(flags, reads, rows, with_index)})

# match_index micro-cases straight from the reference function
mi = []
s2i = {s: n for n, s in INDEXES}
for k in range(60):
    base = INDEXES[k % 4][1]
    q = list(base)
    for _ in range(int(rng.integers(0, 4))):
        q[int(rng.integers(0, 16))] = "ACGT"[int(rng.integers(0, 4))]
    piece = rnd(int(rng.integers(0, 5))) + "".join(q) + rnd(int(rng.integers(0, 5)))
    piece = piece[:int(rng.integers(16, 25))]
    mi.append({"seq": piece, "result": ref.match_index(piece, s2i)})
json.dump({"cases": cases, "match_index": mi, "indexes": INDEXES}, open(os.path.join(HERE, "post_cases.json"), "w"), indent=0)
print("wrote", len(cases), "cases;", sum(len(c["files"]) for c in cases), "files;", len(mi), "match_index cases;",
      {r["result"] for r in mi})
