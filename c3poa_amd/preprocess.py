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


def preprocess(blat, args, tmp_dir, tmp_adapter_dict, num_reads):
    align_psl = tmp_dir + "splint_to_read_alignments.psl"
    if not os.path.exists(align_psl) or os.stat(align_psl).st_size == 0:
        if shutil.which(blat) is None:
            raise RuntimeError(
                "no %s and no blat binary (%r): the splint/strand assignment step is outside the accelerated "
                "path; provide the PSL (c3poa_amd.synth.write_psl does for synthetic data) or install blat"
                % (align_psl, blat))
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
    return parse_psl(align_psl, tmp_adapter_dict)
