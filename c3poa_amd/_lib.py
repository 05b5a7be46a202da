"""ctypes binding of the HIP backend (c3poa_amd/lib/libc3poa_hip.so, C ABI in include/c3poa.h).

There is NO CPU fallback: importing this module without the built library, or creating a handle
without an MI355X, raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C3POA_LIB", os.path.join(_HERE, "lib", "libc3poa_hip.so"))
MAX_PEAKS = 256

STAGE_CONK, STAGE_PEAKS, STAGE_POA, STAGE_POLISH, STAGES_ALL = 1, 2, 4, 8, 15
ST_OK, ST_NOT_ASSIGNED, ST_NO_PEAKS, ST_NO_CONSENSUS, ST_TOO_SHORT, ST_LIMIT = range(6)

EXPORTS = ["c3_default_config", "c3_version", "c3_device_count", "c3_warm_device", "c3_create", "c3_destroy", "c3_last_error", "c3_set_splints",
           "c3_batch_upload", "c3_batch_stage", "c3_batch_commit", "c3_batch_assign", "c3_batch_run", "c3_batch_sync", "c3_batch_results", "c3_batch_results_snapshot", "c3_batch_results_fetch", "c3_batch_timing",
           "c3_fetch_track", "c3_fetch_smoothed", "c3_fetch_raw_peaks", "c3_fetch_draft", "c3_fetch_msa2",
           "c3_call_peaks", "c3_poa_msa", "c3_pairwise_consensus", "c3_determine_consensus", "c3_zero_repeats", "c3_scan_splints",
           "c3_reader_open", "c3_reader_open_range", "c3_reader_close", "c3_reader_error", "c3_reader_names_only", "c3_reader_next", "c3_reader_next_set", "c3_reader_noqual", "c3_reader_reserved_bytes", "c3_reader_range_lost", "c3_bgzf_size", "c3_write_group",
           "c3_scan_adapters", "c3_match_index", "c3_match_index_batch",
           "c3_assign_open", "c3_assign_close", "c3_assign_batch", "c3_assign_seen", "c3_write_splint_psl",
           "c3_host_alloc", "c3_host_free", "c3_writer_reset", "c3_compress_file"]


class Config(C.Structure):
    _fields_ = ([("device", C.c_int), ("conk_match", C.c_int), ("conk_mismatch", C.c_int), ("conk_penalty", C.c_int),
                 ("sg_iters", C.c_int), ("sg_window", C.c_int), ("sg_order", C.c_int), ("mdistcutoff", C.c_int),
                 ("poa_match", C.c_int), ("poa_mismatch", C.c_int), ("poa_o1", C.c_int), ("poa_e1", C.c_int),
                 ("poa_o2", C.c_int), ("poa_e2", C.c_int), ("poa_band_b", C.c_int), ("poa_band_f", C.c_double),
                 ("pol_match", C.c_int), ("pol_mismatch", C.c_int), ("pol_gap", C.c_int), ("pol_window", C.c_int),
                 ("pol_q", C.c_int), ("dang_band", C.c_int), ("slots_poa", C.c_int), ("slots_win", C.c_int), ("zero", C.c_int)])


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_peaks", C.c_int32), ("n_sub", C.c_int32), ("has_front", C.c_int32),
                ("has_tail", C.c_int32), ("front_end", C.c_int32), ("tail_beg", C.c_int32), ("cons_len", C.c_int32),
                ("draft_len", C.c_int32), ("n_win", C.c_int32), ("peaks", C.c_int32 * MAX_PEAKS),
                ("sub_beg", C.c_int32 * MAX_PEAKS), ("sub_end", C.c_int32 * MAX_PEAKS)]


RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_peaks", "<i4"), ("n_sub", "<i4"), ("has_front", "<i4"),
                         ("has_tail", "<i4"), ("front_end", "<i4"), ("tail_beg", "<i4"), ("cons_len", "<i4"),
                         ("draft_len", "<i4"), ("n_win", "<i4"), ("peaks", "<i4", (MAX_PEAKS,)),
                         ("sub_beg", "<i4", (MAX_PEAKS,)), ("sub_end", "<i4", (MAX_PEAKS,))])
assert RESULT_DTYPE.itemsize == C.sizeof(ReadResult)


class Timing(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ms_pack", "ms_conk", "ms_peaks", "ms_poa", "ms_prep", "ms_window",
                                         "ms_stitch", "ms_total")] + \
               [(n, C.c_int64) for n in ("n_reads", "n_bases", "n_windows", "cells_conk", "cells_poa", "cells_polish", "n_poa_redo")] + \
               [(n, C.c_float) for n in ("ms_wall", "ms_host_worklist", "ms_alloc", "ms_host_gap")] + \
               [(n, C.c_int64) for n in ("cells_polish_computed", "n_band_layers", "n_band_fallback", "n_band_mismatch", "n_win_redo", "n_poa_redo16")] + \
               [(n, C.c_float) for n in ("ms_poa_tail", "pad_")]


class HostBatchStruct(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_short", C.c_int64), ("names", C.c_void_p), ("name_off", C.c_void_p),
                ("seqs", C.c_void_p), ("quals", C.c_void_p), ("off", C.c_void_p)]


_lib = None


def load():
    """Load the shared library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("c3poa_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, cp, i64p = C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_void_p
    lib.c3_default_config.argtypes = [C.POINTER(Config)]
    lib.c3_version.restype = C.c_char_p
    lib.c3_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.c3_destroy.argtypes = [vp]
    lib.c3_destroy.restype = None
    lib.c3_last_error.argtypes = [vp]
    lib.c3_last_error.restype = C.c_char_p
    lib.c3_set_splints.argtypes = [vp, C.c_int, cp, i64p]
    lib.c3_batch_upload.argtypes = [vp, C.c_int, vp, vp, i64p, vp, vp]
    lib.c3_batch_stage.argtypes = [vp, C.c_int, vp, vp, i64p, vp, vp]
    lib.c3_batch_commit.argtypes = [vp]
    lib.c3_batch_assign.argtypes = [vp, vp, vp]
    lib.c3_batch_run.argtypes = [vp, C.c_int]
    lib.c3_batch_sync.argtypes = [vp]
    lib.c3_batch_results.argtypes = [vp, vp, vp, C.c_int64, i64p]
    lib.c3_batch_results_snapshot.argtypes = [vp]
    lib.c3_batch_results_fetch.argtypes = [vp, vp, vp, C.c_int64, i64p]
    lib.c3_batch_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.c3_fetch_track.argtypes = [vp, C.c_int, vp, C.c_int64]
    lib.c3_fetch_smoothed.argtypes = [vp, C.c_int, vp, C.c_int64]
    lib.c3_fetch_raw_peaks.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.c3_fetch_draft.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.c3_fetch_msa2.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    lib.c3_call_peaks.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.c3_poa_msa.argtypes = [vp, C.c_int, C.POINTER(cp), ip, vp, C.c_int, ip, vp, C.c_int64, ip]
    lib.c3_zero_repeats.argtypes = [vp, cp, cp, C.c_int, cp, cp, C.c_int, C.c_int, vp, C.c_int, ip]
    lib.c3_scan_splints.argtypes = [vp, vp, vp, vp]
    lib.c3_pairwise_consensus.argtypes = [vp, cp, cp, C.c_int, cp, C.c_int, cp, cp, C.c_int, cp, vp, C.c_int, ip]
    lib.c3_scan_adapters.argtypes = [vp, vp]
    lib.c3_match_index.argtypes = [cp, C.c_int, C.c_int, cp, vp]
    lib.c3_match_index_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, cp, vp, vp]
    lib.c3_assign_open.argtypes = [cp, C.c_int, C.POINTER(cp), C.POINTER(vp)]
    lib.c3_assign_close.argtypes = [vp]
    lib.c3_assign_close.restype = None
    lib.c3_assign_batch.argtypes = [vp, C.POINTER(HostBatchStruct), vp, vp]
    lib.c3_assign_seen.argtypes = [vp, vp, vp]
    lib.c3_write_splint_psl.argtypes = [C.POINTER(HostBatchStruct), vp, vp, vp, C.c_int, C.POINTER(cp), vp, C.c_int, cp, vp]
    lib.c3_host_alloc.argtypes = [C.c_int64, C.POINTER(vp)]
    lib.c3_host_free.argtypes = [vp]
    lib.c3_host_free.restype = None
    lib.c3_writer_reset.restype = None
    lib.c3_compress_file.argtypes = [cp, cp, C.c_int, C.c_int]
    lib.c3_reader_open.argtypes = [cp, C.c_int, C.POINTER(vp)]
    lib.c3_reader_open_range.argtypes = [cp, C.c_int, C.c_int64, C.c_int64, C.POINTER(vp)]
    lib.c3_reader_close.argtypes = [vp]
    lib.c3_reader_close.restype = None
    lib.c3_reader_error.argtypes = [vp]
    lib.c3_reader_names_only.argtypes = [vp, C.c_int]
    lib.c3_reader_names_only.restype = None
    lib.c3_reader_error.restype = C.c_char_p
    for fn in (lib.c3_reader_noqual, lib.c3_reader_reserved_bytes):
        fn.argtypes = [vp]; fn.restype = C.c_int64
    lib.c3_reader_range_lost.argtypes = [vp]
    lib.c3_bgzf_size.argtypes = [C.c_char_p]; lib.c3_bgzf_size.restype = C.c_int64
    lib.c3_reader_next.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.POINTER(HostBatchStruct)]
    lib.c3_reader_next_set.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(HostBatchStruct)]
    lib.c3_write_group.argtypes = [C.POINTER(HostBatchStruct), vp, vp, vp, vp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.c_int]
    lib.c3_determine_consensus.argtypes = [vp, C.c_int, C.POINTER(cp), C.POINTER(cp), ip, cp, cp, C.c_int,
                                           cp, cp, C.c_int, vp, C.c_int, ip, vp, C.c_int, ip]
    _lib = lib
    return lib


class C3Error(RuntimeError):
    pass


def default_config(**kw):
    cfg = Config()
    load().c3_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


class Handle:
    """One per GPU (c3_create / c3_destroy)."""

    def __init__(self, **cfg):
        self.lib = load()
        self.cfg = default_config(**cfg)
        self.h = C.c_void_p()
        rc = self.lib.c3_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise C3Error("c3_create failed (%d): %s" % (rc, self.lib.c3_last_error(None).decode()))
        self.n = 0
        self.lens = None

    def close(self):
        if self.h:
            self.lib.c3_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise C3Error("c3 error %d: %s" % (rc, self.lib.c3_last_error(self.h).decode()))
        return rc

    def set_splints(self, splints):
        """splints: list of sequences (order defines splint_id)"""
        bs = [_b(s) for s in splints]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in bs], out=off[1:])
        self._chk(self.lib.c3_set_splints(self.h, len(bs), b"".join(bs), off.ctypes.data))
        self.n_splints = len(bs)

    def upload(self, seqs, quals, strands, splint_ids=None):
        """seqs/quals: lists of str/bytes (or pre-joined bytes with `lens`), strands: '+'/'-' per read"""
        bs = [_b(s) for s in seqs]
        n = len(bs)
        lens = np.array([len(b) for b in bs], dtype=np.int64)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        self.upload_flat(b"".join(bs), b"".join(_b(q) for q in quals), off,
                         "".join(strands) if not isinstance(strands, (bytes, bytearray)) else strands, splint_ids)

    def upload_flat(self, seq_cat, qual_cat, off, strands, splint_ids=None):
        n = len(off) - 1
        off = np.ascontiguousarray(off, dtype=np.int64)
        st = _b(strands)
        assert len(st) == n and len(seq_cat) == off[-1] == len(qual_cat)
        sid = None
        if splint_ids is not None:
            sid = np.ascontiguousarray(splint_ids, dtype=np.int16)
        sq = np.frombuffer(seq_cat, dtype=np.uint8)
        qq = np.frombuffer(qual_cat, dtype=np.uint8)
        self._chk(self.lib.c3_batch_upload(self.h, n, sq.ctypes.data, qq.ctypes.data, off.ctypes.data,
                                           sid.ctypes.data if sid is not None else None,
                                           np.frombuffer(st, dtype=np.uint8).ctypes.data))
        self.n = n
        self.off = off

    def upload_host(self, hb, strands, splint_ids):
        """upload a HostBatch of the native reader straight from its (page-locked) buffers: no Python copies"""
        st = np.frombuffer(_b(strands), dtype=np.uint8)
        sid = np.ascontiguousarray(splint_ids, dtype=np.int16)
        assert len(st) == hb.n == len(sid)
        self._chk(self.lib.c3_batch_upload(self.h, hb.n, hb.c.seqs, hb.c.quals, hb.c.off, sid.ctypes.data, st.ctypes.data))
        self.n = hb.n
        self.off = hb.off

    def stage_host(self, hb, strands, splint_ids):
        """c3_batch_stage: copy the NEXT batch on the second stream while the resident one is processed; the HostBatch
        must stay alive until commit()"""
        st = np.frombuffer(_b(strands), dtype=np.uint8)
        sid = np.ascontiguousarray(splint_ids, dtype=np.int16)
        assert len(st) == hb.n == len(sid)
        self._chk(self.lib.c3_batch_stage(self.h, hb.n, hb.c.seqs, hb.c.quals, hb.c.off, sid.ctypes.data, st.ctypes.data))
        self._staged = (hb.n, hb.off, hb)

    def commit(self):
        """c3_batch_commit: the staged batch becomes the resident one"""
        self._chk(self.lib.c3_batch_commit(self.h))
        self.n, self.off, _keep = self._staged
        self._staged = None

    def assign(self, splint_ids, strands):
        """c3_batch_assign: splint row / strand of the resident batch (e.g. from scan_splints)"""
        sid = np.ascontiguousarray(splint_ids, dtype=np.int16)
        st = np.frombuffer(_b(strands), dtype=np.uint8)
        assert len(sid) == len(st) == self.n
        self._chk(self.lib.c3_batch_assign(self.h, sid.ctypes.data, st.ctypes.data))

    def upload_pinned(self, pb, splint_ids=None):
        """c3_batch_upload straight from a PinnedBatch (page-locked buffers: DMA, no staging copy)"""
        sid = np.ascontiguousarray(splint_ids, dtype=np.int16) if splint_ids is not None else None
        self._chk(self.lib.c3_batch_upload(self.h, pb.n, pb.seqs, pb.quals, pb.off.ctypes.data,
                                           sid.ctypes.data if sid is not None else None, pb.strand.ctypes.data))
        self.n, self.off = pb.n, pb.off

    def stage_pinned(self, pb, splint_ids=None):
        """c3_batch_stage from a PinnedBatch; the batch must stay alive until commit()"""
        sid = np.ascontiguousarray(splint_ids, dtype=np.int16) if splint_ids is not None else None
        self._chk(self.lib.c3_batch_stage(self.h, pb.n, pb.seqs, pb.quals, pb.off.ctypes.data,
                                          sid.ctypes.data if sid is not None else None, pb.strand.ctypes.data))
        self._staged = (pb.n, pb.off, pb)

    def run(self, stages=STAGES_ALL):
        self._chk(self.lib.c3_batch_run(self.h, stages))
        self.last_timing = self.timing()         # c3_batch_commit clears the library's copy

    def results_raw(self, into=None):
        """(results structured array, consensus byte buffer, cons_off[n+1]) without building Python strings.
        `into`: a ResultBuffers object owned by the caller (pipelines that keep several results alive at once hand each one
        back to their own free list); without it ONE grow-only set per handle is reused, valid until the next call."""
        rb = into if into is not None else self.__dict__.setdefault("_own_rb", ResultBuffers())
        res, buf, coff = rb.fit(self.n, int(self.off[-1]) + 16)
        self._chk(self.lib.c3_batch_results(self.h, res.ctypes.data, buf.ctypes.data, len(buf), coff.ctypes.data))
        return res, buf, coff

    def results_snapshot(self):
        """first half of results_raw (c3_batch_results_snapshot): freezes the results of the resident batch on the device; the
        handle may then commit and run the next batch.  Returns what results_fetch needs to size its buffers."""
        self._chk(self.lib.c3_batch_results_snapshot(self.h))
        return self.n, int(self.off[-1]) + 16

    def results_fetch(self, into, shape):
        """second half (c3_batch_results_fetch): copies the snapshot into `into` (a ResultBuffers); may run on ANOTHER thread
        while this handle's owner works on the next batch.  `shape` = the value results_snapshot returned."""
        res, buf, coff = into.fit(*shape)
        rc = self.lib.c3_batch_results_fetch(self.h, res.ctypes.data, buf.ctypes.data, len(buf), coff.ctypes.data)
        if rc != 0:
            raise C3Error("c3_batch_results_fetch failed (%d)" % rc)
        return res, buf, coff

    def results(self, with_consensus=True):
        res = np.zeros(self.n, dtype=RESULT_DTYPE)
        coff = np.zeros(self.n + 1, dtype=np.int64)
        if not with_consensus:
            self._chk(self.lib.c3_batch_results(self.h, res.ctypes.data, None, 0, coff.ctypes.data))
            return res, None
        cap = int(self.off[-1]) + 16
        buf = np.zeros(cap, dtype=np.uint8)
        self._chk(self.lib.c3_batch_results(self.h, res.ctypes.data, buf.ctypes.data, cap, coff.ctypes.data))
        raw = buf.tobytes()
        cons = [raw[coff[i]:coff[i + 1]].decode() for i in range(self.n)]
        return res, cons

    def timing(self):
        t = Timing()
        self._chk(self.lib.c3_batch_timing(self.h, C.byref(t)))
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_}

    # ---- probes
    def track(self, i):
        L = int(self.off[i + 1] - self.off[i])
        out = np.zeros(L, dtype=np.int32)
        self._chk(self.lib.c3_fetch_track(self.h, i, out.ctypes.data, L))
        return out

    def smoothed(self, i):
        L = int(self.off[i + 1] - self.off[i])
        out = np.zeros(L, dtype=np.float64)
        self._chk(self.lib.c3_fetch_smoothed(self.h, i, out.ctypes.data, L))
        return out

    def raw_peaks(self, i):
        out = np.zeros(MAX_PEAKS, dtype=np.int32)
        n = self._chk(self.lib.c3_fetch_raw_peaks(self.h, i, out.ctypes.data, MAX_PEAKS))
        return out[:n].astype(np.int64)

    def draft(self, i):
        L = int(self.off[i + 1] - self.off[i]) + 8
        buf = C.create_string_buffer(L)
        n = self._chk(self.lib.c3_fetch_draft(self.h, i, buf, L))
        return buf.raw[:n].decode()

    def call_peaks(self, scores, min_dist, return_smoothed=False):
        s = np.ascontiguousarray(scores, dtype=np.int32)
        pk = np.zeros(MAX_PEAKS, dtype=np.int32)
        sm = np.zeros(len(s), dtype=np.float64) if return_smoothed else None
        n = self._chk(self.lib.c3_call_peaks(self.h, s.ctypes.data, len(s), int(min_dist), pk.ctypes.data, MAX_PEAKS,
                                             sm.ctypes.data if sm is not None else None))
        self.n = 0
        out = pk[:n].astype(np.int64)
        return (out, sm) if return_smoothed else out

    def poa_msa(self, seqs, out_cons=True, out_msa=True):
        n = len(seqs)
        if n == 0:
            return [], []
        bs = [_b(s) for s in seqs]
        arr = (C.c_char_p * n)(*bs)
        lens = (C.c_int * n)(*[len(b) for b in bs])
        tot = sum(len(b) for b in bs) + 16
        cons = C.create_string_buffer(tot) if out_cons else None
        msa = C.create_string_buffer(tot * n) if out_msa else None
        cl, ml = C.c_int(0), C.c_int(0)
        self._chk(self.lib.c3_poa_msa(self.h, n, arr, lens, cons, tot, C.byref(cl), msa, tot * n, C.byref(ml)))
        c = [cons.raw[:cl.value].decode()] if out_cons and cl.value else []
        m = [msa.raw[i * ml.value:(i + 1) * ml.value].decode() for i in range(n)] if out_msa and ml.value else []
        return c, m

    def scan_splints(self):
        """splint/strand finder over the resident batch (replaces blat, bin/preprocess.py:61-77).
        returns (table[n][n_splints][2][4] = max, argmax, mean, L; splint_id[n] (-1 = none); strand bytes)"""
        n = self.n
        tab = np.zeros((n, self.n_splints, 2, 4), dtype=np.int32)
        sid = np.zeros(n, dtype=np.int16)
        st = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.c3_scan_splints(self.h, tab.ctypes.data, sid.ctypes.data, st.ctypes.data))
        return tab, sid, st.tobytes()

    def scan_adapters(self):
        """adapter finder over the resident batch (replaces blat in C3POa_postprocessing.py:229-236): table
        [n][n_adapters][2 strands][12] = score, qStart, qEnd, tStart, tEnd, matches, misMatches, qBaseInsert,
        tBaseInsert, qNumInsert, tNumInsert, read length"""
        tab = np.zeros((self.n, self.n_splints, 2, 12), dtype=np.int32)
        self._chk(self.lib.c3_scan_adapters(self.h, tab.ctypes.data))
        return tab

    def match_index_batch(self, pieces, index_seqs):
        """c3_match_index_batch: winning index number (or -1) for every piece (str, <= 64 bases)"""
        n = len(pieces)
        if n == 0:
            return np.zeros(0, dtype=np.int32)
        buf = np.zeros((n, 64), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        for i, pc in enumerate(pieces):
            b = _b(pc)
            lens[i] = len(b)
            buf[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        bs = [_b(x) for x in index_seqs]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in bs], out=off[1:])
        out = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.c3_match_index_batch(self.h, n, buf.ctypes.data, lens.ctypes.data, len(bs), b"".join(bs), off.ctypes.data, out.ctypes.data))
        return out

    def pairwise_consensus(self, msa_rows, subreads, quals):
        """pairwise_consensus(msa_rows, subreads, quals) of bin/consensus.py:76"""
        ra, rb = _b(msa_rows[0]), _b(msa_rows[1])
        sa, sb, qa, qb = _b(subreads[0]), _b(subreads[1]), _b(quals[0]), _b(quals[1])
        out = C.create_string_buffer(len(ra) + 1)
        ol = C.c_int(0)
        self._chk(self.lib.c3_pairwise_consensus(self.h, ra, rb, len(ra), sa, len(sa), qa, sb, len(sb), qb, out, len(ra) + 1, C.byref(ol)))
        return out.raw[:ol.value].decode()

    def zero_repeats(self, d0, q0, d1, q1, min_len=0):
        b0, b1 = _b(d0), _b(d1)
        cap = len(b0) + len(b1) + 16
        out = C.create_string_buffer(cap)
        ol = C.c_int(0)
        self._chk(self.lib.c3_zero_repeats(self.h, b0, _b(q0), len(b0), b1, _b(q1), len(b1), int(min_len), out, cap, C.byref(ol)))
        return out.raw[:ol.value].decode()

    def determine_consensus(self, subs, quals, front=None, tail=None, return_draft=False):
        n = len(subs)
        bs, bq = [_b(s) for s in subs], [_b(q) for q in quals]
        arr, qarr = (C.c_char_p * n)(*bs), (C.c_char_p * n)(*bq)
        lens = (C.c_int * n)(*[len(b) for b in bs])
        tot = sum(len(b) for b in bs) + (len(front[0]) if front else 0) + (len(tail[0]) if tail else 0) + 64
        out, draft = C.create_string_buffer(tot), C.create_string_buffer(tot)
        ol, dl = C.c_int(0), C.c_int(0)
        f = (_b(front[0]), _b(front[1]), len(front[0])) if front else (None, None, 0)
        t = (_b(tail[0]), _b(tail[1]), len(tail[0])) if tail else (None, None, 0)
        self._chk(self.lib.c3_determine_consensus(self.h, n, arr, qarr, lens, f[0], f[1], f[2], t[0], t[1], t[2],
                                                  out, tot, C.byref(ol), draft, tot, C.byref(dl)))
        if return_draft:
            return out.raw[:ol.value].decode(), draft.raw[:dl.value].decode()
        return out.raw[:ol.value].decode()


def match_index(seq, index_seqs):
    """c3_match_index: number of the winning index (file order) or -1 (match_index, C3POa_postprocessing.py:266-285)"""
    bs = [_b(x) for x in index_seqs]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    sq = _b(seq)
    return int(load().c3_match_index(sq, len(sq), len(bs), b"".join(bs), off.ctypes.data))


def device_count():
    return int(load().c3_device_count())


def warm_device(dev):
    """create the HIP context of one device on the calling thread (nothing else); returns the c3_status"""
    return int(load().c3_warm_device(int(dev)))


class ResultBuffers:
    """grow-only host buffers for one batch of results (no page faults per batch); owned by whoever holds the object.
    pinned=True takes them from c3_host_alloc (page-locked): what c3_batch_results_begin needs to copy asynchronously"""

    def __init__(self, pinned=False):
        self.res = self.buf = self.coff = None
        self.pinned = pinned
        self._p = {}
        self._retired = []
        self.lib = load() if pinned else None

    def _alloc(self, name, count, dtype):
        if not self.pinned:
            return np.empty(count, dtype=dtype)
        nbytes = count * np.dtype(dtype).itemsize
        p = C.c_void_p()
        if self.lib.c3_host_alloc(nbytes + 64, C.byref(p)) != 0:
            raise MemoryError("c3_host_alloc(%d)" % nbytes)
        # a buffer that grows hands its predecessor back -- but not before the NEXT fit(): the arrays handed out by the previous
        # fit() are views of it (np.frombuffer owns nothing), and ResultFetcher's callers still read them while the next batch
        # is fetched into the other buffer
        old = self._p.pop(name, None)
        if old is not None:
            self._retired.append(old)
        self._p[name] = p
        return np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dtype, count=count)

    def _release_retired(self):
        for p in self._retired:
            self.lib.c3_host_free(p)
        self._retired = []

    def fit(self, n, cons_cap):
        if self._retired:
            self._release_retired()                   # blocks replaced by the previous fit(): their views are two batches old now
        if self.buf is None or len(self.buf) < cons_cap:
            self.buf = self._alloc("buf", cons_cap + cons_cap // 4, np.uint8)
        if self.res is None or len(self.res) < n:
            self.res = self._alloc("res", n + n // 4 + 1, RESULT_DTYPE)
        if not self.pinned:
            return self.res[:n], self.buf, np.zeros(n + 1, dtype=np.int64)
        if self.coff is None or len(self.coff) < n + 1:
            self.coff = self._alloc("coff", n + n // 4 + 2, np.int64)
        return self.res[:n], self.buf, self.coff[:n + 1]

    def close(self):
        self.res = self.buf = self.coff = None
        self._release_retired()
        for p in self._p.values():
            self.lib.c3_host_free(p)
        self._p = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResultFetcher:
    """software pipeline for the results of one handle: after_run() freezes the resident batch's results on the device
    (c3_batch_results_snapshot) and hands the copy to a helper thread (c3_batch_results_fetch: it runs beside the next batch's
    kernels); it returns the PREVIOUS batch's (res, cons bytes, cons_off), or None the first time.  drain() returns the last."""

    def __init__(self, handle, n_buffers=2, pinned=False):
        from concurrent.futures import ThreadPoolExecutor
        self.h = handle
        self.pool = ThreadPoolExecutor(1)
        self.rbs = [ResultBuffers(pinned=pinned) for _ in range(n_buffers)]
        self.k = 0
        self.fut = None

    def after_run(self):
        prev = self.drain()
        shape = self.h.results_snapshot()
        self.fut = self.pool.submit(self.h.results_fetch, self.rbs[self.k % len(self.rbs)], shape)
        self.k += 1
        return prev

    def drain(self):
        fut, self.fut = self.fut, None
        return fut.result() if fut is not None else None

    def close(self):
        self.drain()
        self.pool.shutdown()


class PinnedBatch:
    """one batch in flat host buffers from c3_host_alloc (page-locked when a GPU is present): what the boundary is handed"""

    def __init__(self, seq_cat, qual_cat, off, strands):
        self.lib = load()
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        self.n = len(self.off) - 1
        st = _b(strands)
        assert len(st) == self.n and len(seq_cat) == self.off[-1] == len(qual_cat)
        self.strand = np.frombuffer(st, dtype=np.uint8).copy()
        self._p = []
        for src in (seq_cat, qual_cat):
            p = C.c_void_p()
            if self.lib.c3_host_alloc(len(src) + 64, C.byref(p)) != 0:
                raise MemoryError("c3_host_alloc(%d)" % len(src))
            C.memmove(p, src, len(src))
            self._p.append(p)
        self.seqs, self.quals = self._p[0].value, self._p[1].value

    def close(self):
        for p in self._p:
            self.lib.c3_host_free(p)
        self._p = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatch:
    """one group of reads held by the native reader (valid until the reader reuses its buffer set)"""

    def __init__(self, c, owner=None):
        self.c = c
        self.owner = owner                       # keeps the reader (and its buffers) alive
        self.n = int(c.n)
        self.n_short = int(c.n_short)
        self.off = np.ctypeslib.as_array(C.cast(c.off, C.POINTER(C.c_int64)), shape=(self.n + 1,)).copy()
        self.name_off = np.ctypeslib.as_array(C.cast(c.name_off, C.POINTER(C.c_int64)), shape=(self.n + 1,)).copy()

    def names(self):
        raw = C.string_at(self.c.names, int(self.name_off[-1])) if self.n else b""
        no = self.name_off
        return [raw[no[i]:no[i + 1]].decode() for i in range(self.n)]

    def read(self, i):
        """(name, seq, qual) of read i as str -- for tests and small inputs"""
        no, o = self.name_off, self.off
        return (C.string_at(self.c.names + int(no[i]), int(no[i + 1] - no[i])).decode(),
                C.string_at(self.c.seqs + int(o[i]), int(o[i + 1] - o[i])).decode(),
                C.string_at(self.c.quals + int(o[i]), int(o[i + 1] - o[i])).decode())


class Reader:
    """native streaming FASTA/FASTQ(.gz) reader (c3_reader_*): mm.fastx_read replacement that yields SoA groups"""

    def __init__(self, path, n_sets=3, names_only=False, byte_range=None):
        """byte_range = (beg, end): only the records that START inside [beg, end) (plain files; c3_reader_open_range)"""
        self.lib = load()
        self.r = C.c_void_p()
        if byte_range is None:
            rc = self.lib.c3_reader_open(_b(str(path)), n_sets, C.byref(self.r))
        else:
            rc = self.lib.c3_reader_open_range(_b(str(path)), n_sets, int(byte_range[0]), int(byte_range[1]), C.byref(self.r))
        if rc != 0:
            raise OSError("cannot open %s" % path)
        if names_only:
            self.lib.c3_reader_names_only(self.r, 1)

    def next(self, max_reads, min_len=0, max_bases=0, set_index=None):
        """next group; set_index names the buffer set to fill (caller-managed free list), None = round-robin"""
        c = HostBatchStruct()
        if set_index is None:
            rc = self.lib.c3_reader_next(self.r, int(max_reads), int(max_bases), int(min_len), C.byref(c))
        else:
            rc = self.lib.c3_reader_next_set(self.r, int(set_index), int(max_reads), int(max_bases), int(min_len), C.byref(c))
        if rc != 0:
            raise ValueError("c3_reader_next: %s" % self.lib.c3_reader_error(self.r).decode())
        hb = HostBatch(c, self)
        hb.set_index = set_index
        return hb

    def noqual(self):
        """records without a quality line read so far (FASTA)"""
        return int(self.lib.c3_reader_noqual(self.r))

    def reserved_bytes(self):
        return int(self.lib.c3_reader_reserved_bytes(self.r))

    def range_lost(self):
        """a byte-range reader whose range holds bytes but no record start (multi-line FASTQ): read the file with one reader"""
        return bool(self.lib.c3_reader_range_lost(self.r))

    def close(self):
        if self.r:
            self.lib.c3_reader_close(self.r)
            self.r = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bgzf_size(path):
    """c3_bgzf_size: inflated size of a BGZF (bgzip) file, -1 when it is not BGZF from end to end -- the length that byte ranges of a
    Reader(..., byte_range=) over such a file are cut from"""
    return int(load().c3_bgzf_size(_b(str(path))))


def write_splint_psl(hb, table, splint_id, strand, splint_names, splint_lens, match, path):
    """c3_write_splint_psl: PSL rows of the assigned reads of one group, appended to `path`; returns the row count"""
    lib = load()
    names = (C.c_char_p * len(splint_names))(*[_b(n) for n in splint_names])
    lens = np.ascontiguousarray(splint_lens, dtype=np.int32)
    tab = np.ascontiguousarray(table, dtype=np.int32)
    sid = np.ascontiguousarray(splint_id, dtype=np.int16)
    st = np.frombuffer(_b(strand), dtype=np.uint8)
    rows = C.c_int64(0)
    rc = lib.c3_write_splint_psl(C.byref(hb.c), tab.ctypes.data, sid.ctypes.data, st.ctypes.data, len(splint_names), names,
                                 lens.ctypes.data, int(match), _b(str(path)), C.byref(rows))
    if rc != 0:
        raise OSError("c3_write_splint_psl failed (%d)" % rc)
    return int(rows.value)


class Assigner:
    """native PSL -> per-read splint / strand assignment (c3_assign_*): bin/preprocess.py:22-45 without Python objects"""

    def __init__(self, psl_path, splint_names):
        self.lib = load()
        self.names = list(splint_names)
        arr = (C.c_char_p * len(self.names))(*[_b(n) for n in self.names])
        self.a = C.c_void_p()
        if self.lib.c3_assign_open(_b(str(psl_path)), len(self.names), arr, C.byref(self.a)) != 0:
            raise OSError("cannot read %s" % psl_path)

    def batch(self, hb):
        """(splint_id int16[n] with -1 = none, strand bytes, number assigned)"""
        sid = np.zeros(hb.n, dtype=np.int16)
        st = np.zeros(hb.n, dtype=np.uint8)
        k = self.lib.c3_assign_batch(self.a, C.byref(hb.c), sid.ctypes.data, st.ctypes.data)
        if k < 0:
            raise RuntimeError("c3_assign_batch failed (%d)" % k)
        return sid, st.tobytes(), int(k)

    def seen(self):
        flags = np.zeros(len(self.names), dtype=np.uint8)
        rows = C.c_int64(0)
        self.lib.c3_assign_seen(self.a, flags.ctypes.data, C.byref(rows))
        return set(n for n, f in zip(self.names, flags) if f), int(rows.value)

    def close(self):
        if self.a:
            self.lib.c3_assign_close(self.a)
            self.a = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compress_file(src, dst=None, level=6, threads=0, remove=True):
    """-co (C3POa.py:86-99, C3POa_postprocessing.py -co): src -> dst (default src + '.gz') as a BGZF-layout gzip file, deflated by
    `threads` threads (0 = the usable cores); removes src afterwards, as the reference's gzip step leaves only the .gz"""
    dst = dst or src + ".gz"
    rc = load().c3_compress_file(_b(src), _b(dst), int(level), int(threads))
    if rc != 0:
        raise C3Error("c3_compress_file(%s -> %s): error %d" % (src, dst, rc))
    if remove:
        os.remove(src)
    return dst


def write_group(hb, res, cons_buf, cons_off, splint_ids, cons_paths, sub_paths, zero=True):
    """c3_write_group: append the consensus / subread records of one group to the per-splint files"""
    lib = load()
    n_spl = len(cons_paths)
    cp = (C.c_char_p * n_spl)(*[_b(p) for p in cons_paths])
    sp = (C.c_char_p * n_spl)(*[_b(p) for p in sub_paths])
    sid = np.ascontiguousarray(splint_ids, dtype=np.int16)
    res = np.ascontiguousarray(res)
    coff = np.ascontiguousarray(cons_off, dtype=np.int64)
    rc = lib.c3_write_group(C.byref(hb.c), res.ctypes.data, cons_buf.ctypes.data if cons_buf is not None else None,
                            coff.ctypes.data, sid.ctypes.data, n_spl, cp, sp, 1 if zero else 0)
    if rc != 0:
        raise OSError("c3_write_group failed (%d)" % rc)
