"""Post-processing of consensus reads: adapter finding, trimming / re-orientation, oligo-dT and 10x demultiplexing.

Host-side mirror of /root/reference/C3POa_postprocessing.py.  The reference aligns the adapters to
the consensus reads with blat (:229-236) and reads six PSL columns back (:238-264); here `find_adapters_gpu` gets the
same information from k_adapter (c3_scan_adapters) and writes the same `adapter_to_consensus_alignment.psl`, which
`parse_blat` -- and a rerun -- consume exactly as upstream.  `match_index` (:266-285, the editdistance loop) is the
native c3_match_index.  Output files and record formats follow write_fasta_file (:287-398).
"""
import gzip
import os
import shutil
import sys

import numpy as np

from . import _lib
from .seqio import fastx_read, revcomp

FLC = "R2C2_full_length_consensus_reads.fasta"
FLC_LEFT = "R2C2_full_length_consensus_reads_left_splint.fasta"
FLC_RIGHT = "R2C2_full_length_consensus_reads_right_splint.fasta"
FLC_10X = "R2C2_full_length_consensus_reads_10X_sequences.fasta"
MUX_TSV = "R2C2_oligodT_multiplexing.tsv"
PSL_NAME = "adapter_to_consensus_alignment.psl"
MIN_SCORE = 22          # > 10 matching bases under the 2 / -4 scoring: the `matches > 10` of parse_blat (:248)


def read_fasta(path, indexes):
    """:218-227 -- {name: seq} (and {seq: name} for the index file), file order"""
    reads, index_dict = {}, {}
    for rec in fastx_read(path):
        reads[rec[0]] = rec[1]
        if indexes:
            index_dict[rec[1]] = rec[0]
    return (reads, index_dict) if indexes else reads


def psl_line(read_name, read_len, adapter_name, adapter_len, strand, e):
    """21 PSL columns from one k_adapter record e = (score, qS, qE, tS, tE, matches, mism, qBaseIns, tBaseIns, qNumIns, tNumIns, L)"""
    cols = [e[5], e[6], 0, 0, e[9], e[7], e[10], e[8], strand, read_name, read_len, e[1], e[2],
            adapter_name, adapter_len, e[3], e[4], 1, "%d," % (e[2] - e[1]), "%d," % e[1], "%d," % e[3]]
    return "\t".join(str(int(c)) if not isinstance(c, str) else c for c in cols)


def find_adapters_gpu(reads, adapter_file, psl_path, batch=200000, handle=None):
    """writes the PSL the reference gets from blat: one row per (read, adapter, strand) whose local alignment reaches
    MIN_SCORE.  reads: {name: seq} in file order."""
    adapters = [(r[0], r[1]) for r in fastx_read(adapter_file)]
    h = handle or _lib.Handle()
    h.set_splints([a[1] for a in adapters])
    names = list(reads)
    with open(psl_path + ".part", "w") as out:
        for b0 in range(0, len(names), batch):
            chunk = names[b0:b0 + batch]
            seqs = [reads[n] for n in chunk]
            keep = [i for i, s in enumerate(seqs) if len(s) > 0]
            if not keep:
                continue
            h.upload([seqs[i] for i in keep], ["!" * len(seqs[i]) for i in keep], "?" * len(keep))
            tab = h.scan_adapters()
            hit = np.argwhere(tab[:, :, :, 0] >= MIN_SCORE)
            rows = []
            for i, a, rc in hit:
                name = chunk[keep[i]]
                rows.append(psl_line(name, len(reads[name]), adapters[a][0], len(adapters[a][1]), "-" if rc else "+", tab[i, a, rc]))
            if rows:
                out.write("\n".join(rows) + "\n")
    os.replace(psl_path + ".part", psl_path)
    if handle is None:
        h.close()


def parse_blat(psl_path, reads):
    """:238-264 -- per read and strand the list of (adapter, matches, projected position)"""
    adapter_dict = {}
    for name, sequence in reads.items():
        adapter_dict[name] = {"+": [("-", 1, 0)], "-": [("-", 1, len(sequence))]}
    with open(psl_path) as fh:
        for line in fh:
            a = line.strip().split("\t")
            if len(a) < 17:
                continue
            read_name, adapter, strand = a[9], a[13], a[8]
            if int(a[5]) < 50 and float(a[0]) > 10:
                if strand == "+":
                    position = int(a[12]) + (int(a[14]) - int(a[16]))        # projected END of the adapter on the read
                else:
                    position = int(a[11]) - (int(a[14]) - int(a[16]))        # projected START
                adapter_dict[read_name][strand].append((adapter, float(a[0]), position))
    return adapter_dict


def match_index(seq, seq_to_idx):
    """:266-285"""
    seqs = list(seq_to_idx)
    k = _lib.match_index(seq, seqs)
    return seq_to_idx[seqs[k]] if k >= 0 else "-"


class _Outputs:
    """the three (or four) FASTA streams of one destination directory, opened lazily in append mode"""

    def __init__(self):
        self.fh = {}

    def get(self, path, name):
        key = path + name
        if key not in self.fh:
            os.makedirs(path, exist_ok=True)
            self.fh[key] = open(key, "a")
        return self.fh[key]

    def close(self):
        for f in self.fh.values():
            f.close()
        self.fh = {}


def match_batch_host(pieces, index_seqs):
    """one native c3_match_index call per piece (host; what the CPU tests use)"""
    return np.array([_lib.match_index(p, index_seqs) for p in pieces], dtype=np.int32)


def match_batch_gpu(pieces, index_seqs, handle=None):
    """c3_match_index_batch: one lane per piece on the GPU"""
    h = handle or _lib.Handle()
    out = h.match_index_batch(pieces, index_seqs)
    if handle is None:
        h.close()
    return out


def write_fasta_file(args, path, adapter_dict, reads, seq_to_idx, idx_to_seq, match_batch=match_batch_gpu):
    """:287-398 -- classification, trimming, orientation, demultiplexing; returns the number of reads written.
    Two passes instead of the reference's one: the reads that pass the adapter rules are collected first so that all
    oligo-dT pieces are matched in one batch (match_index, :266-285), then the records are written in read order."""
    undirectional, barcoded, trim = args.undirectional, args.barcoded, args.trim
    odt = bool(seq_to_idx)
    keep = []                                              # (name, p_pos, m_pos, direction)
    for name, sequence in reads.items():
        plus = sorted((x for x in adapter_dict[name]["+"] if x[0] != "-"), key=lambda x: x[2])
        minus = sorted((x for x in adapter_dict[name]["-"] if x[0] != "-"), key=lambda x: x[2])
        if len(plus) != 1 or len(minus) != 1:
            continue
        p_pos, m_pos = plus[0][2], minus[0][2]
        if m_pos <= p_pos:
            continue
        if undirectional:
            direction = "+"
        elif plus[0][0] != minus[0][0]:
            direction = "+" if plus[0][0] == "5Prime_adapter" else "-"
        else:
            continue
        keep.append((name, p_pos, m_pos, direction))
    fwd_pieces, rev_pieces, fwd_idx, rev_idx = [], [], [], []
    if odt:
        index_seqs = list(seq_to_idx)
        for name, p_pos, m_pos, _d in keep:
            sequence = reads[name]
            fwd_pieces.append(sequence[p_pos - 4:p_pos + 16])
            rev_pieces.append(revcomp(sequence[m_pos - 16:m_pos + 4]))
        hits = match_batch(fwd_pieces + rev_pieces, index_seqs) if keep else np.zeros(0, dtype=np.int32)
        names = [seq_to_idx[x] for x in index_seqs]
        fwd_idx = [names[k] if k >= 0 else "-" for k in hits[:len(keep)]]
        rev_idx = [names[k] if k >= 0 else "-" for k in hits[len(keep):]]
    outs = _Outputs()
    if odt:
        for idx in idx_to_seq:
            if os.path.exists(path + idx):
                shutil.rmtree(path + idx)
        mux = open(path + MUX_TSV, "w")
    else:
        for nm in (FLC, FLC_LEFT, FLC_RIGHT):
            open(path + nm, "w").close()
    if barcoded:
        open(path + FLC_10X, "w").close()
    for k, (name, p_pos, m_pos, direction) in enumerate(keep):
        sequence = reads[name]
        dest = path
        if odt:
            mux.write("%s\t%s\t%s\n" % (name, rev_pieces[k], fwd_pieces[k]))
            forward_index, reverse_index = fwd_idx[k], rev_idx[k]
            idx_name = "no_index_found"
            if forward_index in idx_to_seq and reverse_index not in idx_to_seq:
                direction, idx_name = "-", forward_index
            if reverse_index in idx_to_seq and forward_index not in idx_to_seq:
                direction, idx_name = "+", reverse_index
            dest = path + idx_name + "/"
        seq = sequence[p_pos:m_pos]
        ada = sequence[max(p_pos - 40, 0):m_pos + 40]
        out_name = name + "_" + str(len(seq))
        out, out3, out5 = outs.get(dest, FLC), outs.get(dest, FLC_LEFT), outs.get(dest, FLC_RIGHT)
        if direction == "+":
            out.write(">%s\n%s\n" % (out_name, seq if trim else ada))
            out5.write(">%s\n%s\n" % (out_name, revcomp(sequence[:p_pos])))
            out3.write(">%s\n%s\n" % (out_name, sequence[m_pos:]))
            if barcoded:
                outs.get(path, FLC_10X).write(">%s\n%splus\n" % (out_name, revcomp(sequence[m_pos - 40:m_pos])))
        else:
            out.write(">%s\n%s\n" % (out_name, revcomp(seq) if trim else revcomp(ada)))
            out3.write(">%s\n%s\n" % (out_name, revcomp(sequence[:p_pos + 40])))
            out5.write(">%s\n%s\n" % (out_name, sequence[m_pos:]))
            if barcoded:
                outs.get(path, FLC_10X).write(">%s\n%sminus\n" % (out_name, sequence[p_pos:p_pos + 40]))
    outs.close()
    if odt:
        mux.close()
    return len(keep)


def _gzip_in_place(path):
    # (native, every core: c3_compress_file writes independent gzip members; C3POa_postprocessing.py -co gzips its outputs)
    from . import _lib
    _lib.compress_file(path, path + ".gz", level=6)


def run(args):
    """main() of the reference (:400-426) with the blat step on the GPU.  -n > 1 keeps the reference's multi-process
    output conventions (every index directory exists, -co compresses); the work itself is one GPU pass either way."""
    if not args.output_path.endswith("/"):
        args.output_path += "/"
    os.makedirs(args.output_path, exist_ok=True)
    if args.undirectional and args.barcoded:
        print("Error: undirectional and barcoded are mutually exclusive.")
        sys.exit(1)
    reads = read_fasta(args.input_fasta_file, False)
    if args.index_file:
        idx_to_seq, seq_to_idx = read_fasta(args.index_file, True)
    else:
        idx_to_seq, seq_to_idx = {}, {}
    psl = args.output_path + PSL_NAME
    if not os.path.exists(psl) or os.stat(psl).st_size == 0:
        if getattr(args, "adapter_finder", "gpu") == "gpu":
            find_adapters_gpu(reads, args.adapter_file, psl)
        else:
            raise RuntimeError("no %s: run with --adapter-finder gpu (blat is not bundled)" % psl)
    else:
        print("Reading existing psl file", file=sys.stderr)
    adapter_dict = parse_blat(psl, reads)
    n = write_fasta_file(args, args.output_path, adapter_dict, reads, seq_to_idx, idx_to_seq)
    if args.threads > 1:
        names = [FLC, FLC_LEFT, FLC_RIGHT]
        dirs = [args.output_path]
        if idx_to_seq:
            dirs = [args.output_path + idx + "/" for idx in list(idx_to_seq) + ["no_index_found"]]
        elif args.barcoded:
            names = names + [FLC_10X]
        for d in dirs:                                          # chunk_process cats into every destination (:165-214)
            os.makedirs(d, exist_ok=True)
            for nm in names:
                if not os.path.exists(d + nm):
                    open(d + nm, "w").close()
                if args.compress_output:
                    _gzip_in_place(d + nm)
    return n
