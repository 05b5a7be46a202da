"""Independent checkers of the oracle stages that no reference fixture can pin (abPOA alignment / consensus, racon-style polish).

Nothing here is derived from oracle/*.c: the graph is rebuilt from the MSA ROWS the oracle prints (one node per occupied
(column, base), one edge per pair of consecutive bases of a row), and

  * an unbanded, textbook sequence-to-DAG aligner with the two-piece ("convex") affine gap min(4+2k, 24+k), match +5,
    mismatch -4 (pyabpoa defaults with match=5, bin/determine_consensus.py:30) gives the optimal global score of every
    sequence against the graph of the sequences before it.  The oracle's adaptive-band DP can only lose against it; the
    test requires equality on well-behaved inputs and reports how often the band loses the optimum on ragged ones;
  * a naive heaviest-bundle walk over the rebuilt, weighted graph gives the consensus wherever no tie has to be broken;
  * the polish is checked through a property no alignment detail can fake: layers that agree outvote a wrong draft.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from c3poa_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

NEG = -10 ** 9
MATCH, MISMATCH, O1, E1, O2, E2 = 5, -4, 4, 2, 24, 1


class Dag:
    """graph of the first k MSA rows: node = occupied (column, base); edges between consecutive bases of a row"""

    def __init__(self, rows):
        self.cols = [c for c in range(len(rows[0])) if any(r[c] != "-" for r in rows)]
        self.id = {}
        self.base, self.col = [], []
        for c in self.cols:
            for r in rows:
                if r[c] != "-" and (c, r[c]) not in self.id:
                    self.id[(c, r[c])] = len(self.base)
                    self.base.append(r[c]); self.col.append(c)
        n = len(self.base)
        self.preds = [[] for _ in range(n)]          # in first-traversal order
        self.w = {}                                  # (u, v) -> number of rows walking the edge; u = -1: source, v = -2: sink
        self.out = [[] for _ in range(n)]
        self.starts, self.ends = [], []
        for r in rows:
            path = [self.id[(c, r[c])] for c in range(len(r)) if r[c] != "-"]
            for u, v in zip([-1] + path, path + [-2]):
                if (u, v) not in self.w:
                    self.w[(u, v)] = 0
                    if u == -1:
                        self.starts.append(v)
                    elif v == -2:
                        self.ends.append(u)
                    else:
                        self.preds[v].append(u); self.out[u].append(v)
                self.w[(u, v)] += 1
        self.order = sorted(range(n), key=lambda v: self.col[v])      # columns are a topological order


def gap(k):
    return min(O1 + E1 * k, O2 + E2 * k)


def best_global_score(g, q):
    """optimal global alignment score of q against any source->sink path of the DAG (no band), textbook recurrences"""
    Q = len(q)
    qa = np.frombuffer(q.encode(), dtype=np.uint8)
    srcH = np.array([0] + [-gap(k) for k in range(1, Q + 1)], dtype=np.int64)
    srcN = np.full(Q + 1, NEG, dtype=np.int64)
    H, EA, EB = {}, {}, {}
    for v in g.order:
        ps = [(H[p], EA[p], EB[p]) for p in g.preds[v]]
        if v in g.starts:
            ps.append((srcH, srcN, srcN))
        sub = np.where(qa == ord(g.base[v]), MATCH, MISMATCH)
        M = np.full(Q + 1, NEG, dtype=np.int64)
        e1 = np.full(Q + 1, NEG, dtype=np.int64)
        e2 = np.full(Q + 1, NEG, dtype=np.int64)
        for hp, ap, bp in ps:
            M[1:] = np.maximum(M[1:], hp[:-1] + sub)
            e1 = np.maximum(e1, np.maximum(hp - (O1 + E1), ap - E1))       # the node is skipped (deletion)
            e2 = np.maximum(e2, np.maximum(hp - (O2 + E2), bp - E2))
        h = np.maximum(M, np.maximum(e1, e2))
        f1 = f2 = NEG
        hl = [int(x) for x in h]
        for j in range(1, Q + 1):                                            # query base inserted after the node
            f1 = max(hl[j - 1] - (O1 + E1), f1 - E1)
            f2 = max(hl[j - 1] - (O2 + E2), f2 - E2)
            if f1 > hl[j]:
                hl[j] = f1
            if f2 > hl[j]:
                hl[j] = f2
        H[v] = np.array(hl, dtype=np.int64); EA[v] = e1; EB[v] = e2
    return max(int(H[v][Q]) for v in g.ends)


def _mut(rng, s, **kw):
    return synth._mutate(rng, np.frombuffer(s.encode(), dtype=np.uint8), **kw)[0].decode()


def _scores_vs_bruteforce(seqs):
    _c, rows, _cells = O.poa_msa(seqs, out_cons=False, out_msa=True)
    got = O.poa_last_scores()
    assert len(got) == len(seqs) - 1
    for r, s in zip(rows, seqs):
        assert r.replace("-", "") == s
    out = []
    for k in range(1, len(seqs)):
        out.append((got[k - 1], best_global_score(Dag(rows[:k]), seqs[k])))
    return out


def test_banded_graph_alignment_is_optimal_for_3_to_6_sequences():
    """n = 3..6 noisy copies of one template (the R2C2 case): the oracle's adaptive-band score of every sequence-to-graph
    alignment equals the unbanded optimum"""
    rng = np.random.default_rng(101)
    checked = 0
    for n in (3, 4, 5, 6):
        truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 140))
        seqs = [_mut(rng, truth) for _ in range(n)]
        for got, opt in _scores_vs_bruteforce(seqs):
            assert got == opt, (n, got, opt)
            checked += 1
    assert checked == 2 + 3 + 4 + 5


def test_band_never_beats_the_optimum_and_how_often_it_loses_on_ragged_input():
    """ragged subreads (a 25-40 base chunk missing or duplicated): the band (w = 10 + 0.01 * len) is allowed to lose the
    optimum, never to beat it; the rate is printed (-s) and bounded so that a regression of the band rule shows up"""
    rng = np.random.default_rng(7)
    lost = total = 0
    for trial in range(6):
        truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 150))
        cut = int(rng.integers(30, 90)); ln = int(rng.integers(25, 40))
        ragged = truth[:cut] + truth[cut + ln:] if trial % 2 else truth[:cut] + truth[cut - ln:cut] + truth[cut:]
        seqs = [_mut(rng, truth), _mut(rng, ragged), _mut(rng, truth), _mut(rng, ragged)]
        for got, opt in _scores_vs_bruteforce(seqs):
            assert got <= opt, (trial, got, opt)
            total += 1; lost += got < opt
    print("band lost the optimum in %d of %d ragged alignments" % (lost, total))
    assert total == 18 and lost <= total // 2


def _naive_heaviest_bundle(g):
    """walk from the source along the heaviest out-edge; downstream score = edge weight + score of its target (abPOA's
    heaviest bundling).  Returns (consensus, tie_free): ties are not resolved here"""
    score, nxt, tie_free = {-2: 0}, {}, True
    for v in list(reversed(g.order)) + [-1]:
        outs = g.starts if v == -1 else g.out[v] + ([-2] if (v, -2) in g.w else [])
        best = max(g.w[(v, t)] for t in outs)
        cands = [t for t in outs if g.w[(v, t)] == best]
        if len(cands) > 1:
            top = max(score[t] for t in cands)
            cands = [t for t in cands if score[t] == top]
            tie_free = tie_free and len(cands) == 1
        nxt[v] = cands[-1]
        score[v] = best + score[nxt[v]]
    out, v = [], nxt[-1]
    while v != -2:
        out.append(g.base[v]); v = nxt[v]
    return "".join(out), tie_free


def test_heaviest_bundle_consensus_against_a_naive_walk():
    rng = np.random.default_rng(23)
    checked = 0
    for trial in range(40):
        truth = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(12, 16))))
        seqs = [_mut(rng, truth, sub=0.06, ins=0.03, dele=0.03) for _ in range(int(rng.integers(3, 8)))]
        if any(len(s) < 4 for s in seqs):
            continue
        cons, rows, _ = O.poa_msa(seqs)
        g = Dag(rows)
        assert len(g.base) <= 50
        naive, tie_free = _naive_heaviest_bundle(g)
        if tie_free:
            assert cons == [naive], (seqs, cons, naive)
            checked += 1
    assert checked >= 10


def test_polish_layers_outvote_a_wrong_draft():
    """five agreeing subreads over a draft that carries substitutions, a deletion and an insertion: the polished consensus
    is the subreads' sequence (bin/determine_consensus.py:87-99 -- what racon is run for)"""
    rng = np.random.default_rng(5)
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 1300))
    q = "I" * len(truth)
    clean = O.determine_consensus([truth] * 5, [q] * 5)
    assert clean == truth
    # one bad subread cannot pull the consensus away from four good ones
    bad = list(truth)
    for p in (100, 400, 401, 900):
        bad[p] = "ACGT"[("ACGT".index(bad[p]) + 1) % 4]
    bad = "".join(bad[:600] + bad[603:])
    out = O.determine_consensus([truth, truth, bad, truth, truth], [q, q, "5" * len(bad), q, q])
    assert out == truth
    assert synth.identity(O.determine_consensus([bad, truth, truth], ["5" * len(bad), q, q]), truth) > 0.995


# ---------------------------------------------------------------------------------------------------------------
# Window polish (racon / spoa semantics): independent brute force of the window DP.
# The oracle's capture hook (oracle_py.win_capture) reports, per window alignment, the backbone, the layer (codes, begin,
# end), the end score and the alignment path.  Everything else is rebuilt HERE, from the spec: the graph (fusing the
# reported paths: match -> same node, mismatch -> the aligned sibling with that base or a new one, insertion -> new node),
# spoa's sub-graph rule (depth-first from the end node over in-edges and aligned nodes, ids >= begin), and a textbook
# UNBANDED global linear-gap sequence-to-DAG alignment (match 3, mismatch -5, gap -4).  The oracle's DP score must equal
# the brute-force optimum, its path must be a walk of the graph that re-scores to that optimum, and the two graphs must
# stay identical layer after layer.

PM, PX, PG = 3, -5, -4


class WinGraph:
    def __init__(self, backbone):
        self.base = [int(b) for b in backbone]
        n = len(self.base)
        self.preds = [[i - 1] if i else [] for i in range(n)]       # in first-traversal order
        self.aligned = [[] for _ in range(n)]                        # mismatch siblings (one MSA column)

    def add_node(self, b):
        self.base.append(int(b)); self.preds.append([]); self.aligned.append([])
        return len(self.base) - 1

    def fuse(self, ops, query):
        prev = None
        for node, q in ops:
            if q < 0:
                continue
            b = int(query[q])
            if node >= 0:
                col = [node] + self.aligned[node]
                tgt = next((x for x in ([node] if self.base[node] == b else []) + [y for y in col if self.base[y] == b]), None)
                if tgt is None:
                    tgt = self.add_node(b)
                    for y in col:
                        self.aligned[y].append(tgt)
                    self.aligned[tgt] = list(col)
            else:
                tgt = self.add_node(b)
            if prev is not None and prev not in self.preds[tgt]:
                self.preds[tgt].append(prev)
            prev = tgt

    def subgraph(self, begin, end):
        """spoa Graph::Subgraph: depth-first from `end` over in-edges and aligned nodes, node ids >= begin"""
        inside, stack = set(), [end]
        while stack:
            v = stack.pop()
            if v in inside or v < begin:
                continue
            inside.add(v)
            stack.extend(self.preds[v]); stack.extend(self.aligned[v])
        return inside

    def topo(self, nodes):
        """any topological order of the induced sub-graph in which aligned nodes do not matter (Kahn)"""
        nodes = set(nodes)
        indeg = {v: sum(1 for u in self.preds[v] if u in nodes) for v in nodes}
        succ = {v: [] for v in nodes}
        for v in nodes:
            for u in self.preds[v]:
                if u in nodes:
                    succ[u].append(v)
        ready = sorted(v for v in nodes if indeg[v] == 0)
        out = []
        while ready:
            v = ready.pop()
            out.append(v)
            for w in succ[v]:
                indeg[w] -= 1
                if indeg[w] == 0:
                    ready.append(w)
        assert len(out) == len(nodes)
        return out, succ

    def best_score(self, nodes, query):
        """unbanded optimum of the global alignment of `query` against the induced sub-graph: the path may start at any node
        without predecessor inside and must end at a node without successor inside"""
        order, succ = self.topo(nodes)
        nodes = set(nodes)
        Q = len(query)
        qa = np.asarray(query)
        j = np.arange(Q + 1)
        start = j * PG
        H = {}
        for v in order:
            ps = [H[u] for u in self.preds[v] if u in nodes] or [start]
            sub = np.where(qa == self.base[v], PM, PX)
            key = np.full(Q + 1, NEG, dtype=np.int64)
            for hp in ps:
                key[1:] = np.maximum(key[1:], hp[:-1] + sub)
                key = np.maximum(key, hp + PG)
            H[v] = np.maximum.accumulate(key - PG * j) + PG * j
        return max(int(H[v][Q]) for v in order if not succ[v])

    def path_score(self, nodes, ops, query):
        """re-score a reported path; checks that it is a walk of the sub-graph that consumes the whole query"""
        nodes = set(nodes)
        score, prev, nq = 0, None, 0
        for node, q in ops:
            if node >= 0:
                assert node in nodes
                if prev is None:
                    assert not [u for u in self.preds[node] if u in nodes], "path does not start at a source"
                else:
                    assert prev in self.preds[node], "path uses an edge that does not exist"
                prev = node
            if q >= 0:
                assert q == nq
                nq += 1
            score += (PM if self.base[node] == int(query[q]) else PX) if (node >= 0 and q >= 0) else PG
        assert nq == len(query)
        return score, prev


def _check_windows(subs_quals, min_windows, front=None, tail=None):
    O.win_capture(True)
    try:
        for subs, quals in subs_quals:
            O.determine_consensus(subs, quals, front=front, tail=tail)
        als = O.win_captured()
    finally:
        O.win_capture(False)
    graphs, n_full, n_sub = {}, 0, 0
    for a in als:
        if a["layer"] == 0:
            g = graphs[a["win"]] = WinGraph(a["base"][:a["blen"]])
        g = graphs[a["win"]]
        # the graph rebuilt from the reported paths is the graph the oracle aligned against
        assert len(g.base) == a["n"] and g.base == [int(b) for b in a["base"]]
        assert [sorted(p) for p in g.preds] == [sorted(p) for p in a["preds"]]
        nodes = set(range(len(g.base))) if a["full"] else g.subgraph(a["begin"], a["end"])
        assert nodes == set(int(v) for v in np.nonzero(a["mask"])[0]), "sub-graph rule"
        q = a["query"]
        opt = g.best_score(nodes, q)
        assert a["score"] == opt, (a["win"], a["layer"], a["score"], opt)
        ps, last = g.path_score(nodes, [(int(x), int(y)) for x, y in a["ops"]], q)
        assert ps == opt and last == a["end_node"]
        g.fuse([(int(x), int(y)) for x, y in a["ops"]], q)
        n_full += a["full"]; n_sub += not a["full"]
    assert n_full + n_sub >= min_windows
    return n_full, n_sub


def test_window_dp_against_an_unbanded_brute_force():
    """full-span layers and sub-graph layers (dangling pieces, ragged subreads) of >= 30 window alignments: the oracle's score
    is the unbanded optimum of an independently rebuilt graph, its path re-scores to it, spoa's sub-graph rule holds"""
    rng = np.random.default_rng(77)
    jobs = []
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 1080))
    for n in (3, 4, 5):
        subs = [_mut(rng, truth) for _ in range(n)]
        jobs.append((subs, ["".join(chr(33 + int(x)) for x in rng.integers(6, 30, len(s))) for s in subs]))
    n_full, n_sub = _check_windows(jobs, 12)
    # dangling pieces and a subread that misses a chunk: layers that do not span their window -> sub-graph alignments
    t2 = "".join("ACGT"[i] for i in rng.integers(0, 4, 560))
    subs = [_mut(rng, t2), _mut(rng, t2[:200] + t2[260:]), _mut(rng, t2), _mut(rng, t2)]
    quals = ["".join(chr(33 + int(x)) for x in rng.integers(6, 30, len(s))) for s in subs]
    fr, tl = _mut(rng, t2[330:]), _mut(rng, t2[:240])
    f2, s2 = _check_windows([(subs, quals)], 10, front=(fr, "5" * len(fr)), tail=(tl, "5" * len(tl)))
    assert n_full + f2 >= 12 and n_sub + s2 >= 10, (n_full, n_sub, f2, s2)
    assert n_full + n_sub + f2 + s2 >= 30


def test_window_rules_copy_and_tgs_trim():
    """racon's window rules on constructed cases: fewer than 3 sequences -> the window is copied and, with no
    polished window at all, the target is dropped; TGS windows (mean layer length > 1000) are trimmed to the span covered by
    at least (n-1)/2 layers"""
    rng = np.random.default_rng(9)
    t = "".join("ACGT"[i] for i in rng.integers(0, 4, 700))
    q = "I" * len(t)
    # one subread: backbone + 1 layer = 2 sequences per window -> nothing polished -> no consensus (racon without -u)
    assert O.determine_consensus([t], [q]) == ""
    # two subreads: 3 sequences per window -> polished
    assert O.determine_consensus([t, t], [q, q]) == t
    # TGS trim: five long layers (mean > 1000) of which only two reach the last 150 bases: the consensus ends where the
    # coverage drops below (6 - 1) / 2 = 2 ... both still reach, so nothing is lost; with one reaching, the tail is trimmed
    long_t = "".join("ACGT"[i] for i in rng.integers(0, 4, 1400))
    ql = "I" * len(long_t)
    full = O.determine_consensus([long_t] * 5, [ql] * 5)
    assert full == long_t
    short = long_t[:1250]
    out = O.determine_consensus([long_t] + [short] * 4, [ql] + ["I" * len(short)] * 4)
    assert out.startswith(long_t[:1000]) and len(out) < len(long_t) and synth.identity(out, short) > 0.98
