"""Deterministic synthetic R2C2 concatemer generator.

clean read = insert[-k0:] + (splint + insert) * n + splint + insert[:k1]; 50 % of reads are
reverse-complemented (strand '-'); i.i.d. errors sub 4 % / ins 2.5 % / del 3.5 %;
Phred ~ clip(N(12,4),2,40) on correct bases and clip(N(7,3),2,40) on erroneous ones.
The ground-truth consensus of a read is the error-free splint-centre-to-splint-centre unit in
read orientation.
"""
import numpy as np

from .seqio import revcomp

SPLINT1 = ("TGAGGCTGATGAGTTCCATATTTGAAAAGTTTTCATCACTACTTAGTTTTTTGATAGCTTCAAGCCAGAGTTGTCTTTTTCTATCTACTC"
           "TCATACAACCAATAAATGCTGAAATGAATTCTAAGCGGAGATCGCCTAGTGATTTTAAACTATTGCTGGCAGCATTCTTGAGTCCAATAT"
           "AAAAGTATTGTGTACCTTTTGCTGGGTCAGGTTGTTCTTTAGGAGGAGTAAAAGGATCAAATGCACTAAACGAAACTGAAACAAGCGATC"
           "GAAAATATCCCTTT")

CONFIGS = {
    # name: (reads, seed, insert, n_lo, n_hi, L_target or None, k)
    "cfg1": dict(reads=1000, seed=1, insert=1216, n_lo=3, n_hi=3, k0=108, k1=108, mdist=500),
    "cfg2": dict(reads=100000, seed=2, insert=1216, n_lo=3, n_hi=3, k0=108, k1=108, mdist=500),
    "cfg3": dict(reads=1000000, seed=3, insert=1000, n_lo=2, n_hi=10, k0=108, k1=108, mdist=500),
    "cfg4": dict(reads=100000, seed=4, insert=1256, n_lo=12, n_hi=12, k0=618, k1=618, mdist=1500),
    "cfg5": dict(reads=10000000, seed=5, insert=1216, n_lo=3, n_hi=3, k0=108, k1=108, mdist=500),
    # not a BASELINE config: long cDNA inserts (real R2C2 libraries hold 2-6 kb molecules; abPOA has no length limit,
    # bin/determine_consensus.py:30-47).  Insert drawn per read from the list; 3-5 repeats; reads of 10-32 kb
    "cfgL": dict(reads=20000, seed=6, insert=(3000, 6000), n_lo=3, n_hi=5, k0=108, k1=108, mdist=500),
    # not BASELINE configs: the cfg2 shape with every error rate x 1.5 (6 / 3.75 / 5.25 % = 15 %: where raw ONT R2C2 reads live) and x 2.0 --
    # the polish band's certificate fails more often there, the POA bands drift further (round 6: the bench line watches x 1.5)
    "cfg2e15": dict(reads=100000, seed=12, insert=1216, n_lo=3, n_hi=3, k0=108, k1=108, mdist=500, err=1.5),
    "cfg2e20": dict(reads=100000, seed=13, insert=1216, n_lo=3, n_hi=3, k0=108, k1=108, mdist=500, err=2.0),
}

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, clean, sub=0.04, ins=0.025, dele=0.035):
    """clean: uint8 ASCII array -> (seq bytes, qual bytes)"""
    n = len(clean)
    r = rng.random(n)
    is_del = r < dele
    is_sub = (r >= dele) & (r < dele + sub)
    base = clean.copy()
    if is_sub.any():
        # substitute with one of the 3 other bases
        cur = np.searchsorted(_ACGT, base[is_sub])  # A C G T sorted in ASCII
        base[is_sub] = _ACGT[(cur + rng.integers(1, 4, is_sub.sum())) % 4]
    n_ins = (rng.random(n) < ins).astype(np.int64)
    keep = ~is_del
    out_len = int(keep.sum() + n_ins.sum())
    seq = np.empty(out_len, dtype=np.uint8)
    err = np.zeros(out_len, dtype=bool)
    # position of each kept base = number of kept bases + inserted bases before it
    pos = np.cumsum(keep.astype(np.int64) + n_ins) - (keep.astype(np.int64) + n_ins)
    kp = pos[keep]
    seq[kp] = base[keep]
    err[kp] = is_sub[keep]
    ip = pos[n_ins > 0] + keep[n_ins > 0].astype(np.int64)
    seq[ip] = _ACGT[rng.integers(0, 4, len(ip))]
    err[ip] = True
    q = np.where(err, rng.normal(7, 3, out_len), rng.normal(12, 4, out_len))
    q = np.clip(np.rint(q), 2, 40).astype(np.uint8) + 33
    return seq.tobytes(), q.tobytes()


def make_read(rng, splint, insert_len, n, k0, k1, err=1.0):
    """returns (seq str, qual str, strand '+'/'-', truth str in read orientation)"""
    ins = _ACGT[rng.integers(0, 4, insert_len)].tobytes().decode()
    clean = ins[len(ins) - k0:] + (splint + ins) * n + splint + ins[:k1]
    h = len(splint) // 2
    truth = splint[h:] + ins + splint[:h]
    strand = "+"
    if rng.random() < 0.5:
        clean, truth, strand = revcomp(clean), revcomp(truth), "-"
    seq, qual = _mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=0.04 * err, ins=0.025 * err, dele=0.035 * err)
    return seq.decode(), qual.decode(), strand, truth


def generate(cfg="cfg2", n_reads=None, seed=None, splint=SPLINT1, start=0):
    """Yield (name, seq, qual, strand, truth).  `start` skips that many reads of the stream
    deterministically (each read has its own child RNG) so shards can be generated independently."""
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    total = c["reads"] if n_reads is None else n_reads
    sd = c["seed"] if seed is None else seed
    for i in range(start, start + total):
        rng = np.random.default_rng([sd, i])
        n = int(rng.integers(c["n_lo"], c["n_hi"] + 1))
        ins = c["insert"] if isinstance(c["insert"], int) else int(c["insert"][int(rng.integers(0, len(c["insert"])))])
        seq, qual, strand, truth = make_read(rng, splint, ins, n, c["k0"], c["k1"], c.get("err", 1.0))
        yield ("r%08d" % i, seq, qual, strand, truth)


def write_psl(path, records, splint_name="Splint1"):
    """Side input that lets preprocess() skip blat (bin/preprocess.py:17-34): 21 tab-separated
    PSL columns, matches(col0) > 50, qBaseInsert(col5) < 50, strand col8, read name col9,
    splint name col13."""
    with open(path, "w") as fh:
        for name, seq, _q, strand, _t in records:
            cols = ["280", "4", "0", "0", "0", "0", "0", "0", strand, name, str(len(seq)), "0", "284",
                    splint_name, "284", "0", "284", "1", "284,", "0,", "0,"]
            fh.write("\t".join(cols) + "\n")


def identity(a, b):
    """1 - Levenshtein(a,b)/max(len)"""
    if not a or not b:
        return 0.0
    la, lb = len(a), len(b)
    prev = np.arange(lb + 1)
    bb = np.frombuffer(b.encode(), dtype=np.uint8)
    for i in range(1, la + 1):
        cur = np.empty(lb + 1, dtype=np.int64)
        cur[0] = i
        sub = prev[:-1] + (bb != ord(a[i - 1]))
        best = np.minimum(sub, prev[1:] + 1)
        # left dependency: cur[j] = min(best[j-1], cur[j-1]+1) -> prefix min of (best - j) + j
        x = np.concatenate(([cur[0]], best)) - np.arange(lb + 1)
        cur = np.minimum.accumulate(x) + np.arange(lb + 1)
        prev = cur
    return 1.0 - prev[-1] / max(la, lb)
