"""Pure host-side formatting of the reference's output records (no GPU needed).

Mirrors, byte for byte:
  * the consensus FASTA record of /root/reference/C3POa.py:167-173
  * the subread FASTQ records and their naming asymmetry of
    /root/reference/bin/determine_consensus.py:57-62, 69-77 (kept: _1.._n; first dangling: _0;
    second dangling: _<repeats+1>) and :108-114 (zero-repeat pieces: _0, _1)
"""


def consensus_header(name, qual, seq_len, repeats, cons_len):
    """C3POa.py:168-171 -- avg_qual is over the WHOLE raw read, Python float formatting."""
    avg_qual = round(sum([ord(x) - 33 for x in qual]) / seq_len, 2)
    return ">" + name + "_" + "_".join([str(x) for x in [avg_qual, seq_len, repeats, cons_len]])


def consensus_record(name, qual, seq_len, repeats, consensus):
    return consensus_header(name, qual, seq_len, repeats, len(consensus)) + "\n" + consensus + "\n"


def fastq_record(qname, sub, q):
    return "@{name}\n{sub}\n+\n{q}\n".format(name=qname, sub=sub, q=q)


def subread_records(name, subreads, sub_qual, dangling, qual_dangling):
    """FASTQ text determine_consensus appends to the group subread file for repeats >= 1."""
    out = []
    repeats = len(subreads)
    for i in range(repeats):
        out.append(fastq_record(name + "_" + str(i + 1), subreads[i], sub_qual[i]))
    for j in range(len(dangling)):
        qname = name + "_" + str(j) if j == 0 else name + "_" + str(repeats - 1 + 2)
        out.append(fastq_record(qname, dangling[j], qual_dangling[j]))
    return "".join(out)


def zero_repeat_records(name, dangling, qual_dangling):
    """determine_consensus.py:108-114 -- written before the rescue is even attempted."""
    return "".join(fastq_record(name + "_" + str(i), dangling[i], qual_dangling[i]) for i in range(len(dangling)))
