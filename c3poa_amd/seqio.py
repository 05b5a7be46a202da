"""FASTA/FASTQ reading and reverse complement for the host side.

Replaces what the reference takes from mappy (`mm.fastx_read`, `mm.revcomp`;
/root/reference/C3POa.py:201,232,234,239) with dependency-free Python.
"""
import gzip

_COMP = str.maketrans("ACGTUNacgtunRYKMBDHVrykmbdhv", "TGCAANtgcaanYRMKVHDByrmkvhdb")


def revcomp(seq):
    """mappy.revcomp equivalent (C3POa.py:234)."""
    return seq.translate(_COMP)[::-1]


def _open(path):
    if str(path).endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def fastx_read(path, read_comment=False):
    """Yield (name, seq, qual) like mappy.fastx_read; qual is None for FASTA records."""
    with _open(path) as fh:
        line = fh.readline()
        while line:
            if not line.strip():
                line = fh.readline()
                continue
            if line[0] == ">":
                hdr = line[1:].rstrip("\n")
                parts = []
                line = fh.readline()
                while line and line[0] != ">":
                    parts.append(line.strip())
                    line = fh.readline()
                name, _, comment = hdr.partition(" ")
                name = name.split("\t")[0]
                yield (name, "".join(parts), None, comment) if read_comment else (name, "".join(parts), None)
            elif line[0] == "@":
                hdr = line[1:].rstrip("\n")
                seq = fh.readline().strip()
                plus = fh.readline()
                if not plus.startswith("+"):
                    raise ValueError("malformed FASTQ near %r" % hdr)
                qual = fh.readline().strip()
                name, _, comment = hdr.partition(" ")
                name = name.split("\t")[0]
                yield (name, seq, qual, comment) if read_comment else (name, seq, qual)
                line = fh.readline()
            else:
                raise ValueError("not FASTA/FASTQ: %r" % line[:40])
