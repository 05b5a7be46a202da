"""ctypes binding of the CPU ORACLE (oracle/libc3oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under c3poa_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libc3oracle.so")


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "conk_match", "conk_mismatch", "conk_penalty", "sg_iters", "sg_window", "sg_order",
        "mdistcutoff", "poa_match", "poa_mismatch", "poa_o1", "poa_e1", "poa_o2", "poa_e2",
        "poa_band_b")] + [("poa_band_f", C.c_double)] + [(n, C.c_int) for n in (
        "pol_match", "pol_mismatch", "pol_gap", "pol_window", "pol_q", "dang_band", "zero", "zr_match", "zr_mismatch",
        "zr_gapo", "zr_gape", "zr_min_score", "zr_max_cells")]


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_int), ("n_peaks", C.c_int), ("peaks", C.c_int64 * 256),
                ("n_sub", C.c_int), ("sub_beg", C.c_int * 256), ("sub_end", C.c_int * 256),
                ("has_front", C.c_int), ("has_tail", C.c_int), ("front_end", C.c_int),
                ("tail_beg", C.c_int), ("cons_len", C.c_int),
                ("cells_conk", C.c_int64), ("cells_poa", C.c_int64), ("cells_polish", C.c_int64)]


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "libc3oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.c3o_conk.restype = C.c_int64
    return _lib


def default_params(**kw):
    p = Params()
    lib().c3o_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def conk(splint, seq, penalty=20, match=5, mismatch=-4):
    sp, sq = _b(splint), _b(seq)
    out = np.zeros(len(sq), dtype=np.int32)
    lib().c3o_conk(sp, len(sp), sq, len(sq), match, mismatch, penalty,
                   out.ctypes.data_as(C.c_void_p))
    return out


def scan_splints(seqs, splints, penalty=20, match=5, mismatch=-4):
    """ORACLE of c3_scan_splints: conk track of every read against every splint on both strands;
    table[n][n_splints][2][4] = max, first argmax, floor(mean), L and the assignment rule
    (best max; accepted when max >= 6 * mean and max >= match*51*52/2)."""
    comp = bytes.maketrans(b"ACGTUacgtu", b"TGCAAtgcaa")
    n = len(seqs)
    tab = np.zeros((n, len(splints), 2, 4), dtype=np.int32)
    sid = np.full(n, -1, dtype=np.int16)
    strand = bytearray(b"?" * n)
    for i, rd in enumerate(seqs):
        best, bs = -1, -1
        for s, sp in enumerate(splints):
            for rc in (0, 1):
                q = _b(sp) if rc == 0 else _b(sp).translate(comp)[::-1]
                t = conk(q, rd, penalty, match, mismatch)
                L = len(t)
                tab[i, s, rc] = (int(t.max()), int(t.argmax()), int(t.astype(np.int64).sum() // max(L, 1)), L) if L else (-1, 0, 0, 0)
                if tab[i, s, rc, 0] > bs:
                    bs, best = int(tab[i, s, rc, 0]), s * 2 + rc
        if best >= 0 and bs >= match * 51 * 52 // 2 and bs >= 6 * int(tab[i, best >> 1, best & 1, 2]):
            sid[i] = best >> 1
            strand[i] = ord("-" if best & 1 else "+")
    return tab, sid, bytes(strand)


def adapter_align(read, adapter, rc=False, params=None):
    """ORACLE of c3_scan_adapters for one (read, adapter, strand): 12 ints (score, qStart, qEnd, tStart, tEnd, matches,
    misMatches, qBaseInsert, tBaseInsert, qNumInsert, tNumInsert, read length)"""
    P = params or default_params()
    out = np.zeros(12, dtype=np.int32)
    rd, ad = _b(read), _b(adapter)
    lib().c3o_adapter_align(rd, len(rd), ad, len(ad), 1 if rc else 0, C.byref(P), out.ctypes.data_as(C.c_void_p))
    return out


def savgol(y, window=41, order=2):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(y)
    rc = lib().c3o_savgol(y.ctypes.data_as(C.c_void_p), len(y), window, order,
                          out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("savgol")
    return out


def call_peaks(scores, min_dist, iters=3, window=41, order=2, return_smoothed=False):
    s = np.ascontiguousarray(scores, dtype=np.int32)
    cap = len(s) // 2 + 4
    pk = np.zeros(cap, dtype=np.int64)
    sm = np.zeros(len(s), dtype=np.float64)
    n = lib().c3o_call_peaks(s.ctypes.data_as(C.c_void_p), len(s), int(min_dist), iters, window, order,
                             pk.ctypes.data_as(C.c_void_p), cap, sm.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise ValueError("track too short")
    return (pk[:n].copy(), sm) if return_smoothed else pk[:n].copy()


def find_peaks(x, height, distance):
    x = np.ascontiguousarray(x, dtype=np.float64)
    cap = len(x) // 2 + 4
    pk = np.zeros(cap, dtype=np.int64)
    lib().c3o_find_peaks.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]
    n = lib().c3o_find_peaks(x.ctypes.data_as(C.c_void_p), len(x), float(height), int(distance),
                             pk.ctypes.data_as(C.c_void_p), cap)
    return pk[:n].copy()


def rounding(x, base):
    return lib().c3o_rounding(int(x), int(base))


class SplitInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_peaks", "n_sub", "has_front", "has_tail", "front_end", "tail_beg")]


def split(peaks, S, L):
    pk = np.ascontiguousarray(peaks, dtype=np.int64)
    n = len(pk)
    po = np.zeros(n + 1, dtype=np.int64)
    sb = np.zeros(n + 1, dtype=np.int32)
    se = np.zeros(n + 1, dtype=np.int32)
    info = SplitInfo()
    lib().c3o_split(pk.ctypes.data_as(C.c_void_p), n, S, L, po.ctypes.data_as(C.c_void_p),
                    sb.ctypes.data_as(C.c_void_p), se.ctypes.data_as(C.c_void_p), C.byref(info))
    return dict(peaks=po[:info.n_peaks].tolist(),
                subs=[(int(sb[i]), int(se[i])) for i in range(info.n_sub)],
                has_front=info.has_front, has_tail=info.has_tail,
                front_end=info.front_end, tail_beg=info.tail_beg)


def poa_msa(seqs, out_cons=True, out_msa=True, params=None):
    P = params or default_params()
    n = len(seqs)
    bs = [_b(s) for s in seqs]
    arr = (C.c_char_p * max(n, 1))(*bs)
    lens = (C.c_int * max(n, 1))(*[len(s) for s in bs])
    tot = sum(len(s) for s in bs) + 8
    cons = C.create_string_buffer(tot)
    msa = C.create_string_buffer(tot * max(n, 1))
    cl, ml, cells = C.c_int(0), C.c_int(0), C.c_int64(0)
    rc = lib().c3o_poa_msa(arr, lens, n, C.byref(P), cons if out_cons else None, tot, C.byref(cl),
                           msa if out_msa else None, C.c_int64(tot * max(n, 1)), C.byref(ml), C.byref(cells))
    if rc:
        raise RuntimeError("poa_msa rc=%d" % rc)
    c = [cons.raw[:cl.value].decode()] if (out_cons and cl.value) else []
    m = [msa.raw[i * ml.value:(i + 1) * ml.value].decode() for i in range(n)] if (out_msa and ml.value) else []
    return c, m, cells.value


def poa_last_scores():
    """end-cell DP scores of the sequence-to-graph alignments of the last poa_msa call (one per sequence after the first)"""
    out = (C.c_int32 * 256)()
    n = lib().c3o_poa_last_scores(out, 256)
    return [int(out[i]) for i in range(min(n, 256))]


def win_capture(on=True):
    """checker hook: start (and clear) / stop the capture of the polish's window alignments on this thread"""
    lib().c3o_win_capture(1 if on else 0)


def win_captured():
    """the captured window alignments as dicts: win, layer, blen, begin, end, full, n, Q, score, end_node, base[n], grp[n], order[n],
    preds[n] (in-edge order), mask[n], query[Q], ops [(node, q)] (-1 = none)"""
    L = lib()
    L.c3o_win_capture_get.restype = C.c_int64
    L.c3o_win_capture_get.argtypes = [C.c_void_p, C.c_int64]
    n = L.c3o_win_capture_get(None, 0)
    buf = np.zeros(max(n, 1), dtype=np.int32)
    L.c3o_win_capture_get(buf.ctypes.data_as(C.c_void_p), n)
    out, i = [], 0
    while i < n:
        assert buf[i] == 0x57494e44
        win, layer, blen, begin, end, full, nn, Q, score, endn, nops = (int(x) for x in buf[i + 1:i + 12])
        i += 12
        base = buf[i:i + nn].copy(); i += nn
        grp = buf[i:i + nn].copy(); i += nn
        order = buf[i:i + nn].copy(); i += nn
        preds = []
        for _v in range(nn):
            k = int(buf[i]); preds.append([int(x) for x in buf[i + 1:i + 1 + k]]); i += 1 + k
        mask = buf[i:i + nn].astype(bool); i += nn
        query = buf[i:i + Q].copy(); i += Q
        ops = buf[i:i + 2 * nops].reshape(nops, 2).copy(); i += 2 * nops
        out.append(dict(win=win, layer=layer, blen=blen, begin=begin, end=end, full=bool(full), n=nn, Q=Q, score=score,
                        end_node=endn, base=base, grp=grp, order=order, preds=preds, mask=mask, query=query, ops=ops))
    return out


def normalize_len(row, qual):
    r, q = _b(row), _b(qual)
    out = C.create_string_buffer(len(r) + 8)
    n = lib().c3o_normalize_len(r, len(r), q, len(q), out)
    return out.raw[:n].decode("latin1")


def pairwise_consensus(msa_rows, subreads, quals):
    a, b = _b(msa_rows[0]), _b(msa_rows[1])
    sa, sb, qa, qb = _b(subreads[0]), _b(subreads[1]), _b(quals[0]), _b(quals[1])
    out = C.create_string_buffer(len(a) + 8)
    n = lib().c3o_pairwise_consensus(a, b, len(a), sa, len(sa), qa, sb, len(sb), qb, out, len(a) + 8)
    return out.raw[:n].decode()


def determine_consensus(subs, quals, front=None, tail=None, params=None, return_draft=False):
    """front/tail: (seq, qual) or None"""
    P = params or default_params()
    n = len(subs)
    bs, bq = [_b(s) for s in subs], [_b(q) for q in quals]
    arr = (C.c_char_p * n)(*bs)
    qarr = (C.c_char_p * n)(*bq)
    lens = (C.c_int * n)(*[len(s) for s in bs])
    tot = sum(len(s) for s in bs) + 64
    out = C.create_string_buffer(tot)
    draft = C.create_string_buffer(tot)
    dl = C.c_int(0)
    cells = (C.c_int64 * 2)(0, 0)
    f = (_b(front[0]), _b(front[1]), len(front[0])) if front else (None, None, 0)
    t = (_b(tail[0]), _b(tail[1]), len(tail[0])) if tail else (None, None, 0)
    n_out = lib().c3o_determine_consensus(arr, qarr, lens, n, f[0], f[1], f[2], t[0], t[1], t[2],
                                          C.byref(P), out, tot, draft, tot, C.byref(dl), cells)
    res = out.raw[:n_out].decode()
    if return_draft:
        return res, draft.raw[:dl.value].decode(), (cells[0], cells[1])
    return res


def zero_repeats(d0, q0, d1, q1, params=None):
    P = params or default_params()
    b0, b1 = _b(d0), _b(d1)
    out = C.create_string_buffer(len(b0) + len(b1) + 16)
    cells = C.c_int64(0)
    n = lib().c3o_zero_repeats(b0, _b(q0), len(b0), b1, _b(q1), len(b1), C.byref(P), out, len(b0) + len(b1) + 16, C.byref(cells))
    return out.raw[:n].decode()


def process_batch(splint, reads, strands, params=None, threads=1):
    """reads: list of (seq, qual) str; strands: list of '+'/'-'.  returns (results, consensi)"""
    from c3poa_amd.seqio import revcomp
    P = params or default_params()
    n = len(reads)
    lens = np.array([len(r[0]) for r in reads], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    seqs = "".join(r[0] for r in reads).encode()
    quals = "".join(r[1] for r in reads).encode()
    res = (ReadResult * n)()
    cons = C.create_string_buffer(int(off[-1]) + 8)
    st = "".join(strands).encode()
    sp = _b(splint)
    lib().c3o_process_batch(sp, _b(revcomp(splint)), len(sp), seqs, quals, off.ctypes.data_as(C.c_void_p), n,
                            st, C.byref(P), threads, res, cons, off.ctypes.data_as(C.c_void_p))
    outs = []
    raw = cons.raw          # one copy (cons.raw copies the whole buffer on every access)
    for i in range(n):
        outs.append(raw[off[i]:off[i] + res[i].cons_len].decode() if res[i].status == 0 else "")
    return res, outs
