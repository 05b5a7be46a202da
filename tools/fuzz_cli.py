#!/usr/bin/env python3
"""Randomised END-TO-END sweep of the command line against the oracle:
   python tools/fuzz_cli.py [n_reads] [seed] [gpu|psl]
The reads of tools/fuzz_parity2.py's generator (random splint, low-complexity inserts, hundreds of repeats, 100+ kb reads,
chimeras) go into a FASTQ file; C3POa.py runs on it with SMALL GPU batches, several byte-range readers and several writer threads
(so that reads cross batch, range and writer boundaries); both output files are then compared, record by record, with what the
Python formatter of the reference's record formats (c3poa_amd/analyze.write_group, pinned by the golden cases) makes of the
ORACLE's results for the same reads.  Mode `psl` (default) hands the splint / strand of every read over in the PSL as upstream's
blat step would; mode `gpu` lets the GPU finder assign them (reads it leaves unassigned or assigns to the other strand are
reported, not compared)."""
import importlib.util
import os
import shutil
import sys
import tempfile
import types
import numpy as np
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fuzz_parity2", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity2.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)


def parse_fasta(path):
    from c3poa_amd.seqio import fastx_read
    return {n: s for n, s, _q in fastx_read(path)} if os.path.exists(path) else {}


def parse_fastq(path):
    from c3poa_amd.seqio import fastx_read
    out = {}
    if os.path.exists(path):
        for n, s, q in fastx_read(path):
            assert n not in out, "duplicate subread record " + n
            out[n] = (s, q)
    return out


if __name__ == "__main__":
    import C3POa
    from c3poa_amd import _lib, analyze, synth
    from oracle import oracle_py as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "psl"
    splint, mdist, reads, strands = fz.generate(n, 70_000 + seed)
    rng = np.random.default_rng(seed)
    lencut = int(rng.choice([0, 500, 1000, 3000]))
    group = int(rng.choice([1, 7, 64, 1000]))
    names = ["read%d_%s" % (i, "x" * int(rng.integers(0, 30))) for i in range(len(reads))]
    tmp = tempfile.mkdtemp(prefix="fuzzcli_", dir=os.environ.get("FUZZ_TMP", "/tmp"))
    try:
        fq, fa, out = tmp + "/reads.fastq", tmp + "/splint.fasta", tmp + "/out"
        # the FASTQ text: sometimes multi-line records, CRLF line ends, no final newline (kseq.h semantics, as mm.fastx_read)
        style = int(rng.integers(0, 4))
        parts = []
        for nm, (s, q) in zip(names, reads):
            if style == 1 and len(s) > 200:
                w = int(rng.integers(60, 5000))
                s_txt = "\n".join(s[i:i + w] for i in range(0, len(s), w)); q_txt = "\n".join(q[i:i + w] for i in range(0, len(q), w))
            else:
                s_txt, q_txt = s, q
            parts.append("@%s some comment\n%s\n+%s\n%s\n" % (nm, s_txt, nm if style == 1 else "", q_txt))
        text = "".join(parts)
        if style == 2:
            text = text.replace("\n", "\r\n")
        if style == 3:
            text = text[:-1]
        data = text.encode()
        comp = int(rng.integers(0, 3))                              # plain / gzip (one stream) / BGZF (parallel inflate)
        if comp == 1:
            import gzip
            fq += ".gz"
            with gzip.open(fq, "wb", compresslevel=1) as fh:
                fh.write(data)
        elif comp == 2:
            import struct, zlib
            fq += ".gz"
            block = int(rng.choice([0xff00, 4096, 20000]))
            with open(fq, "wb") as fh:
                for i in range(0, len(data), block):
                    chunk = data[i:i + block]
                    co = zlib.compressobj(1, zlib.DEFLATED, -15)
                    c_ = co.compress(chunk) + co.flush()
                    fh.write(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(c_) + 8 - 1)
                             + c_ + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
                fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        else:
            open(fq, "wb").write(data)
        open(fa, "w").write(">Sp1\n%s\n" % splint)
        os.makedirs(out + "/tmp")
        if mode == "psl":
            unassigned = set(int(x) for x in rng.choice(len(reads), max(1, len(reads) // 25), replace=False))       # reads blat found nothing for
            synth.write_psl(out + "/tmp/splint_to_read_alignments.psl",
                            [(nm, r[0], r[1], st, None) for i, (nm, r, st) in enumerate(zip(names, reads, strands)) if i not in unassigned], "Sp1")
        os.environ["C3_GPU_BATCH_READS"] = str(int(rng.choice([16, 48, 200])))
        os.environ["C3_READERS_PER_GPU"] = str(int(rng.choice([1, 3, 5])))
        os.environ["C3_MIN_RANGE_BYTES"] = "4096"
        argv = ["-r", fq, "-s", fa, "-o", out, "-l", str(lencut), "-d", str(mdist), "-g", str(group)]
        zero = rng.random() < 0.8
        if not zero:
            argv.append("-z")
        C3POa.main(C3POa.parse_args(argv))
        got_c = parse_fasta(out + "/Sp1/R2C2_Consensus.fasta")
        got_s = parse_fastq(out + "/Sp1/R2C2_Subreads.fastq")
        # expectation: oracle results of the reads the reference's loop would process, through the Python record formatter
        if mode == "psl":
            keep = [i for i in range(len(reads)) if i not in unassigned and len(reads[i][0]) >= lencut]
            st_used = {i: strands[i] for i in keep}
        else:                                                           # the finder's own assignment, read back from the PSL it wrote
            st_used = {}
            byname = {nm: i for i, nm in enumerate(names)}
            for line in open(out + "/tmp/splint_to_read_alignments.psl"):
                c = line.rstrip("\n").split("\t")
                st_used[byname[c[9]]] = c[8]
            keep = [i for i in sorted(st_used) if len(reads[i][0]) >= lencut]
            wrong = sum(1 for i in keep if st_used[i] != strands[i])
            print("gpu finder: %d of %d reads assigned, %d to the other strand than the generator's" % (len(st_used), len(reads), wrong))
        P = O.default_params(mdistcutoff=mdist, zero=1 if zero else 0)
        ores, ocons = O.process_batch(splint, [reads[i] for i in keep], [st_used[i] for i in keep], params=P, threads=16)
        res = np.zeros(len(keep), dtype=_lib.RESULT_DTYPE)
        for k, o in enumerate(ores):
            r = res[k]
            r["status"], r["n_peaks"], r["n_sub"], r["has_front"], r["has_tail"] = o.status, o.n_peaks, o.n_sub, o.has_front, o.has_tail
            r["front_end"], r["tail_beg"], r["cons_len"] = o.front_end, o.tail_beg, o.cons_len
            for x in range(o.n_sub):
                r["sub_beg"][x], r["sub_end"][x] = o.sub_beg[x], o.sub_end[x]
        recs = [(names[i], reads[i][0], reads[i][1]) for i in keep]
        ad = {names[i]: ["Sp1", st_used[i]] for i in keep}
        exp_dir = tmp + "/exp/"
        analyze.write_group(types.SimpleNamespace(out_path=exp_dir, zero=zero), recs, res, ocons, ad, 1)
        exp_c = parse_fasta(exp_dir + "Sp1/tmp1/R2C2_Consensus.fasta")
        exp_s = parse_fastq(exp_dir + "Sp1/tmp1/subreads.fastq")
        bad = 0
        for what, g, e in (("consensus", got_c, exp_c), ("subread", got_s, exp_s)):
            for k in sorted(set(g) | set(e)):
                if g.get(k) != e.get(k):
                    bad += 1
                    if bad <= 8:
                        print("DIFFERENT %s record %s: cli %s | expected %s" % (what, k[:60], "missing" if k not in g else "present", "missing" if k not in e else "present"))
        log = open(out + "/c3poa.log").read().splitlines()
        n_short = sum(1 for r in reads if len(r[0]) < lencut)
        if log[1] != "Total reads: %d" % len(reads) or (mode == "psl" and log[3].split(" (")[0] != "Under len cutoff: %d" % n_short):
            bad += 1
            print("DIFFERENT log:", log[1:4], "expected total", len(reads), "short", n_short)
        print("seed %d mode %s input %s/%s: %d reads (%d processed), splint %d nt, -l %d -d %d -g %d%s, batch %s readers %s: %d consensus + %d subread records, differences %d" % (
            seed, mode, ("one-line", "multi-line", "CRLF", "no final newline")[style], ("plain", "gzip", "BGZF")[comp], len(reads), len(keep), len(splint), lencut, mdist, group, "" if zero else " -z", os.environ["C3_GPU_BATCH_READS"], os.environ["C3_READERS_PER_GPU"],
            len(got_c), len(got_s), bad))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(1 if bad else 0)
