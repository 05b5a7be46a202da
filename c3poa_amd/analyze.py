"""Host-side mirror of the reference's batch seam, backed by the HIP library.

analyze_reads(args, reads, splint_dict, adapter_dict, adapter_set, iteration, racon)
    same signature, same side effects as /root/reference/C3POa.py:110-173: appends to
    <out>/<splint>/tmp<iteration>/R2C2_Consensus.fasta and .../subreads.fastq.  `racon` is accepted
    and ignored (the polish runs on the GPU).  One call = one c3_batch_upload + c3_batch_run.
"""
import os

import numpy as np

from . import _lib
from .records import consensus_record, subread_records, zero_repeat_records

_HANDLES = {}


def get_handle(device=0, mdistcutoff=500, zero=True):
    key = (device, mdistcutoff, bool(zero))
    if key not in _HANDLES:
        _HANDLES[key] = _lib.Handle(device=device, mdistcutoff=mdistcutoff, zero=1 if zero else 0)
    return _HANDLES[key]


def run_batch(handle, reads, splint_names, splint_dict, adapter_dict):
    """reads: list of (name, seq, qual).  returns (results structured array, list of consensus str)."""
    sid = {n: i for i, n in enumerate(splint_names)}
    strands, ids = [], []
    for name, _seq, _qual in reads:
        ad = adapter_dict.get(name)
        if not ad:
            strands.append("?"); ids.append(0)          # C3POa.py:115-116
        else:
            strands.append("-" if ad[1] == "-" else "+"); ids.append(sid[ad[0]])
    handle.upload([r[1] for r in reads], [r[2] for r in reads], strands, np.array(ids, dtype=np.int16))
    handle.run()
    return handle.results()


def analyze_reads(args, reads, splint_dict, adapter_dict, adapter_set, iteration, racon=None, device=0):
    if not reads:
        return
    splint_names = sorted(splint_dict)
    h = get_handle(device, args.mdistcutoff, getattr(args, "zero", True))
    h.set_splints([splint_dict[n][0] for n in splint_names])
    res, cons = run_batch(h, reads, splint_names, splint_dict, adapter_dict)
    write_group(args, reads, res, cons, adapter_dict, iteration)


def write_group(args, reads, res, cons, adapter_dict, iteration):
    """file side effects of analyze_reads + determine_consensus for one group"""
    handles = {}

    def fh(splint, fname):
        tmp_dir = args.out_path + splint + "/tmp" + str(iteration) + "/"
        key = (splint, fname)
        if key not in handles:
            os.makedirs(tmp_dir, exist_ok=True)
            handles[key] = open(tmp_dir + fname, "a+")
        return handles[key]

    for i, (name, seq, qual) in enumerate(reads):
        r = res[i]
        if r["status"] in (_lib.ST_NOT_ASSIGNED, _lib.ST_NO_PEAKS, _lib.ST_TOO_SHORT):
            continue                                     # C3POa.py:115,125,131: no output at all
        splint = adapter_dict[name][0]
        ns = int(r["n_sub"])
        subs = [seq[r["sub_beg"][k]:r["sub_end"][k]] for k in range(ns)]
        squal = [qual[r["sub_beg"][k]:r["sub_end"][k]] for k in range(ns)]
        dang, dqual = [], []
        if r["has_front"]:
            dang.append(seq[:r["front_end"]]); dqual.append(qual[:r["front_end"]])
        if r["has_tail"]:
            dang.append(seq[r["tail_beg"]:]); dqual.append(qual[r["tail_beg"]:])
        if ns == 0:
            # determine_consensus.py:14-18: zero-repeat pieces are written before the rescue is tried
            if getattr(args, "zero", True) and len(dang) == 2:
                fh(splint, "subreads.fastq").write(zero_repeat_records(name, dang, dqual))
                if r["status"] == _lib.ST_OK and cons[i]:      # rescue succeeded: repeats == 0 (determine_consensus.py:18)
                    fh(splint, "R2C2_Consensus.fasta").write(consensus_record(name, qual, len(seq), 0, cons[i]))
            continue
        if r["status"] == _lib.ST_LIMIT:
            continue
        fh(splint, "subreads.fastq").write(subread_records(name, subs, squal, dang, dqual))
        if r["status"] == _lib.ST_OK and cons[i]:
            fh(splint, "R2C2_Consensus.fasta").write(consensus_record(name, qual, len(seq), ns, cons[i]))
    for f in handles.values():
        f.close()
