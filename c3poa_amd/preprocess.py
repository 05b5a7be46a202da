"""Splint assignment from the blat PSL (host side).

Mirrors /root/reference/bin/preprocess.py:12-45: the PSL at <out>/tmp/splint_to_read_alignments.psl is
reused when it exists and is non-empty (that is also how the synthetic benchmark bypasses blat); rows
with qBaseInsert (col 5) < 50 and matches (col 0) > 50 count, the best-by-matches row decides splint
and strand.  Running blat itself (bin/preprocess.py:61-112) is outside the accelerated path
(SURVEY.md 8(f)-1): it is attempted only if the configured binary exists.
"""
import os
import shutil
import subprocess
import sys


def parse_psl(align_psl, tmp_adapter_dict):
    """returns (adapter_dict{name: [splint, strand]}, adapter_set, no_splint_reads)"""
    adapter_set = set()
    with open(align_psl) as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            cols = line.split("\t")
            read_name, adapter, strand = cols[9], cols[13], cols[8]
            gaps, score = float(cols[5]), float(cols[0])
            if gaps < 50 and score > 50:
                tmp_adapter_dict[read_name].append([adapter, float(cols[0]), strand])   # KeyError as upstream (App. A.16)
                adapter_set.add(adapter)
    adapter_dict, no_splint = {}, 0
    for name, alignments in tmp_adapter_dict.items():
        best = sorted(alignments, key=lambda x: x[1], reverse=True)[0]
        if not best[0]:
            no_splint += 1
            continue
        adapter_set.add(best[0])
        adapter_dict[name] = [best[0], best[2]]
    return adapter_dict, adapter_set, no_splint


def equivalent_matches(score, match=5):
    """length m of the perfect ungapped match whose diagonal sum match*m*(m+1)/2 equals `score` -- the figure
    written to PSL column 0 so that the reference's `matches > 50` filter (bin/preprocess.py:32) keeps its meaning"""
    return int(((1.0 + 8.0 * max(0, int(score)) / match) ** 0.5 - 1.0) / 2.0)


def psl_row(read_name, read_len, splint_name, splint_len, strand, score, offset, match=5):
    m = min(equivalent_matches(score, match), splint_len)
    q_start = max(0, min(int(offset), read_len))
    q_end = min(read_len, q_start + splint_len)
    cols = [m, splint_len - m, 0, 0, 0, 0, 0, 0, strand, read_name, read_len, q_start, q_end,
            splint_name, splint_len, 0, splint_len, 1, "%d," % splint_len, "%d," % q_start, "0,"]
    return "\t".join(str(c) for c in cols)


def gpu_find_splints(args, align_psl, batch_reads=131072, handle=None):
    """GPU splint/strand finder: one PSL row per accepted read (best candidate only, like the reference keeps the
    best row per read, bin/preprocess.py:39).  Reads are streamed by the native reader in page-locked batches."""
    import numpy as np
    from . import _lib
    from .seqio import fastx_read
    splints = [(s[0], s[1]) for s in fastx_read(args.splint_file)]
    h = handle or _lib.Handle()
    h.set_splints([s[1] for s in splints])
    rd = _lib.Reader(args.reads, n_sets=1)
    n_rows = 0
    open(align_psl + ".part", "w").close()
    names_s, lens_s = [s[0] for s in splints], [len(s[1]) for s in splints]
    while True:
        hb = rd.next(batch_reads, args.lencutoff, 1 << 30)
        if hb.n == 0:
            break
        h.upload_host(hb, b"?" * hb.n, np.zeros(hb.n, dtype=np.int16))
        tab, sid, st = h.scan_splints()
        # rows formatted natively (c3_write_splint_psl); psl_row() above is the scalar statement of the same row
        n_rows += _lib.write_splint_psl(hb, tab, sid, st, names_s, lens_s, h.cfg.conk_match, align_psl + ".part")
    rd.close()
    os.replace(align_psl + ".part", align_psl)
    if handle is None:
        h.close()
    return n_rows


def ensure_psl(blat, args, tmp_dir):
    """make sure <tmp_dir>/splint_to_read_alignments.psl exists (bin/preprocess.py:13-20): reuse it, or create it with the
    GPU finder (default) or the blat binary.  Returns its path."""
    align_psl = tmp_dir + "splint_to_read_alignments.psl"
    finder = getattr(args, "splint_finder", "gpu")
    if (not os.path.exists(align_psl) or os.stat(align_psl).st_size == 0) and finder == "gpu":
        print("Assigning splints to reads on the GPU", file=sys.stderr)
        gpu_find_splints(args, align_psl)
    elif not os.path.exists(align_psl) or os.stat(align_psl).st_size == 0:
        if shutil.which(blat) is None:
            raise RuntimeError("no %s and no blat binary (%r); use --splint-finder gpu" % (align_psl, blat))
        print("Aligning splints to reads with blat", file=sys.stderr)
        fa = tmp_dir + "R2C2_temp_for_BLAT.fasta"
        from .seqio import fastx_read
        with open(fa, "w") as fh:
            for rd in fastx_read(args.reads):
                if len(rd[1]) >= args.lencutoff:
                    fh.write(">%s\n%s\n" % (rd[0], rd[1]))
        with open(tmp_dir + "blat_messages.log", "w") as log:
            subprocess.run([blat, "-noHead", "-stepSize=1", "-t=DNA", "-q=DNA", "-minScore=15", "-minIdentity=10",
                            args.splint_file, fa, align_psl], stdout=log, check=True)
        os.remove(fa)
    else:
        print("Reading existing psl file", file=sys.stderr)
    return align_psl


def preprocess(blat, args, tmp_dir, tmp_adapter_dict, num_reads):
    """same signature and return value as bin/preprocess.py:12 (the CLI itself uses the native assignment,
    c3_assign_*, which returns the same splint / strand per read without a Python object per read)"""
    return parse_psl(ensure_psl(blat, args, tmp_dir), tmp_adapter_dict)
