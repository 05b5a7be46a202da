"""Per-stage shims with the reference's own call shapes (SURVEY.md 8(b) "stage seams"), each backed by
the HIP library through the C ABI.  They exist so that the reference's call sites can be pointed at
this package one stage at a time and so that the parity tests read like the
reference's code:

    conk.conk(splint, seq, penalty)                      C3POa.py:123
    call_peaks(scores, min_dist, iters, window, order)   bin/call_peaks.py:8
    msa_aligner(match=5).msa(seqs, out_cons, out_msa)    bin/determine_consensus.py:30,34,43
    determine_consensus(args, read, subreads, ...)       bin/determine_consensus.py:10
"""
import numpy as np

from . import _lib
from .records import subread_records

_H = {}


def _handle(**cfg):
    key = tuple(sorted(cfg.items()))
    if key not in _H:
        _H[key] = _lib.Handle(**cfg)
    return _H[key]


class conk:
    """`from c3poa_amd.shims import conk; conk.conk(splint, seq, 20)`"""

    @staticmethod
    def conk(splint, seq, penalty=20):
        h = _handle(conk_penalty=penalty)
        h.set_splints([splint])
        h.upload([seq], ["I" * len(seq)], ["+"])
        h.run(_lib.STAGE_CONK)
        return h.track(0)


def call_peaks(scores, min_dist, iters=3, window=41, order=2):
    """returns an int64 ndarray, or [] when the median gate rejects the track (as the reference)"""
    h = _handle(sg_iters=iters, sg_window=window, sg_order=order)
    pk = h.call_peaks(np.asarray(scores, dtype=np.int32), min_dist)
    return pk if len(pk) else []


class _MsaResult:
    def __init__(self, cons, msa):
        self.cons_seq, self.msa_seq = cons, msa
        self.n_seq = len(msa)


class msa_aligner:
    """pyabpoa.msa_aligner look-alike (only the parameters the reference sets are exposed)"""

    def __init__(self, match=2, mismatch=4, gap_open1=4, gap_ext1=2, gap_open2=24, gap_ext2=1, extra_b=10, extra_f=0.01):
        self.cfg = dict(poa_match=match, poa_mismatch=mismatch, poa_o1=gap_open1, poa_e1=gap_ext1, poa_o2=gap_open2,
                        poa_e2=gap_ext2, poa_band_b=extra_b, poa_band_f=extra_f)

    def msa(self, seqs, out_cons, out_msa):
        if not seqs:
            return _MsaResult([], [])
        cons, msa = _handle(**self.cfg).poa_msa(list(seqs), out_cons=out_cons, out_msa=out_msa)
        return _MsaResult(cons, msa)


def pairwise_consensus(msa_rows, subreads, quals):
    """bin/consensus.py:76 (imported at bin/determine_consensus.py:5, called :36-40)"""
    return _handle().pairwise_consensus(msa_rows, subreads, quals)


def determine_consensus(args, read, subreads, sub_qual, dangling_subreads, qual_dangling_subreads, racon=None,
                        tmp_dir=None, subread_file=None):
    """(final_cons, repeats) as bin/determine_consensus.py:10-104; appends the subread FASTQ records to
    `subread_file` exactly as the reference does.  The dangling list is [front][, tail] in read order; a
    single dangling piece is classified by where it sits in the read."""
    name, seq, qual = read[0], read[1], read[2]
    repeats = len(subreads)
    if repeats == 0:
        # determine_consensus.py:14-18 / zero_repeats :106-136
        if getattr(args, "zero", True) and len(dangling_subreads) == 2:
            if subread_file:
                from .records import zero_repeat_records
                with open(subread_file, "a+") as fh:
                    fh.write(zero_repeat_records(name, dangling_subreads, qual_dangling_subreads))
            h = _handle(mdistcutoff=getattr(args, "mdistcutoff", 500))
            cons = h.zero_repeats(dangling_subreads[0], qual_dangling_subreads[0], dangling_subreads[1],
                                  qual_dangling_subreads[1], getattr(args, "mdistcutoff", 500))
            if cons:
                return cons, 0
        return "", 0
    front = tail = None
    d = list(zip(dangling_subreads, qual_dangling_subreads))
    if len(d) == 2:
        front, tail = d
    elif len(d) == 1:
        if seq.startswith(d[0][0]) and not seq.endswith(d[0][0]):
            front = d[0]
        else:
            tail = d[0]
    if subread_file:
        with open(subread_file, "a+") as fh:
            fh.write(subread_records(name, subreads, sub_qual, dangling_subreads, qual_dangling_subreads))
    h = _handle(mdistcutoff=getattr(args, "mdistcutoff", 500))
    cons = h.determine_consensus(subreads, sub_qual, front, tail)
    return cons, repeats
