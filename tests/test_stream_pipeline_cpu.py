"""Host logic of the streamed CLI pipeline (c3poa_amd/stream.py) with SEVERAL workers, on CPU.

The product pipeline (native reader -> worker threads -> native writer) runs unchanged; only the GPU handle is replaced by
a stand-in that answers through the oracle (tests may use the oracle as the checker's engine) and sleeps for a random time,
so batches complete OUT OF ORDER.  What is under test is the ownership of the reader's buffer sets and of the result
buffers across worker threads (the `-n N` path of C3POa.py:236-256): every record of every group must come out exactly
once, with its own name, subreads and qualities.
"""
import os
import random
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from c3poa_amd import _lib, stream, synth  # noqa: E402
from c3poa_amd.seqio import fastx_read, revcomp  # noqa: E402


class FakeHandle:
    """stand-in for _lib.Handle: same methods the pipeline calls; results from the oracle; random latency"""
    delays = random.Random(7)

    def __init__(self, **cfg):
        self.cfg = types.SimpleNamespace(conk_match=5)
        self.md = cfg.get("mdistcutoff", 500)
        self.cur = self.staged = None
        self.last_timing = {"ms_pack": 0.0, "ms_total": 0.0}

    def set_splints(self, splints):
        self.splint = splints[0]

    def _snap(self, hb, strands):
        # copy out of the reader's buffers NOW, as the device upload does
        return [hb.read(i) for i in range(hb.n)], strands.decode() if isinstance(strands, bytes) else strands, hb.n, hb.off

    def upload_host(self, hb, strands, splint_ids):
        self.cur = self._snap(hb, strands)
        self.n, self.off = hb.n, hb.off

    def stage_host(self, hb, strands, splint_ids):
        self.staged = self._snap(hb, strands)

    def commit(self):
        self.cur, self.staged = self.staged, None
        self.n, self.off = self.cur[2], self.cur[3]

    def run(self):
        time.sleep(self.delays.random() * 0.05)

    def timing(self):
        return self.last_timing

    def results_snapshot(self):
        # (the real handle refuses a second snapshot before the first was fetched: the pipeline must respect that)
        assert getattr(self, "snap", None) is None, "snapshot taken before the previous one was fetched"
        self.snap = self.cur
        return self.cur[2], int(self.cur[3][-1]) + 16

    def results_fetch(self, into, shape):
        """runs on the pipeline's fetch thread while the device thread has moved on to the next batch"""
        from oracle import oracle_py as O
        reads, st, n, off = self.snap
        assert shape == (n, int(off[-1]) + 16)
        P = O.default_params(mdistcutoff=self.md)
        ores, ocons = O.process_batch(self.splint, [(r[1], r[2]) for r in reads], list(st), params=P, threads=4)
        res, buf, coff = into.fit(n, int(off[-1]) + 16)
        res[:] = np.zeros(n, dtype=_lib.RESULT_DTYPE)
        pos = 0
        for i, (r, c) in enumerate(zip(ores, ocons)):
            for f in ("status", "n_peaks", "n_sub", "has_front", "has_tail", "front_end", "tail_beg"):
                res[i][f] = getattr(r, f)
            res[i]["cons_len"] = len(c)
            res[i]["sub_beg"][:r.n_sub] = r.sub_beg[:r.n_sub]
            res[i]["sub_end"][:r.n_sub] = r.sub_end[:r.n_sub]
            coff[i] = pos
            buf[pos:pos + len(c)] = np.frombuffer(c.encode(), dtype=np.uint8)
            pos += len(c)
        coff[n] = pos
        time.sleep(self.delays.random() * 0.03)
        self.snap = None
        return res, buf, coff

    def close(self):
        pass


def _run(tmp_path, tag, recs, n_dev, batch, monkeypatch, readers_per_gpu=1, raw=False):
    out = str(tmp_path / tag) + "/"
    os.makedirs(out)
    fq = str(tmp_path / "reads.fastq")
    if not os.path.exists(fq):
        with open(fq, "w") as fh:
            for r in recs:
                fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
    monkeypatch.setattr(_lib, "Handle", FakeHandle)
    monkeypatch.setenv("C3_GPU_BATCH_READS", str(batch))
    monkeypatch.setenv("C3_READERS_PER_GPU", str(readers_per_gpu))
    monkeypatch.setenv("C3_MIN_RANGE_BYTES", "1")
    args = types.SimpleNamespace(out_path=out, reads=fq, groupSize=1000, lencutoff=1000, mdistcutoff=500, zero=True,
                                 compress_output=False)
    sd = {"Splint1": [synth.SPLINT1, revcomp(synth.SPLINT1)]}
    adapter = {r[0]: ["Splint1", r[3]] for r in recs}
    n = stream.run(args, sd, adapter, {"Splint1"}, n_dev)
    assert n == len(recs)
    assert sorted(os.listdir(out + "Splint1")) == ["R2C2_Consensus.fasta", "R2C2_Subreads.fastq"]       # no part file left
    if raw:
        return open(out + "Splint1/R2C2_Consensus.fasta", "rb").read(), open(out + "Splint1/R2C2_Subreads.fastq", "rb").read()
    return (sorted(fastx_read(out + "Splint1/R2C2_Consensus.fasta")), sorted(fastx_read(out + "Splint1/R2C2_Subreads.fastq")))


def test_three_workers_out_of_order_equal_one_worker(tmp_path, monkeypatch):
    recs = list(synth.generate("cfg1", n_reads=60))
    one = _run(tmp_path, "one", recs, 1, 1000, monkeypatch)
    many = _run(tmp_path, "many", recs, 3, 4, monkeypatch)           # 15 batches over 3 workers
    assert one == many
    assert len(one[0]) == 60 and len({r[0] for r in one[1]}) == len(one[1])
    # every subread record carries the bases of ITS read
    byname = {r[0]: r for r in recs}
    for name, seq, qual in many[1][::7]:
        src = byname[name.rsplit("_", 1)[0]]
        assert seq in src[1] and qual in src[2]


def test_byte_range_readers_cover_the_file_once(tmp_path, monkeypatch):
    """2 workers x 3 byte-range readers each (6 parser threads over 6 contiguous parts of the FASTQ), all appending to the
    same output files: every record exactly once; and with ONE worker the record order is the strict round-robin of its
    ranges, i.e. it does not depend on thread timing"""
    recs = list(synth.generate("cfg1", n_reads=50))
    one = _run(tmp_path, "one", recs, 1, 1000, monkeypatch)
    many = _run(tmp_path, "many", recs, 2, 3, monkeypatch, readers_per_gpu=3)
    assert one == many and len(one[0]) == 50
    a = _run(tmp_path, "a", recs, 1, 4, monkeypatch, readers_per_gpu=3, raw=True)
    b = _run(tmp_path, "b", recs, 1, 4, monkeypatch, readers_per_gpu=3, raw=True)
    assert a == b and a[0].count(b">") == 50


def test_result_fetcher_hands_back_the_previous_batch_in_order():
    """_lib.ResultFetcher (bench.py's results pipeline): after_run() snapshots batch k and returns batch k-1's buffers, drain()
    returns the last; the two buffer sets alternate, so what after_run returned stays intact while the next fetch runs"""
    import threading

    class H:
        def __init__(self):
            self.k, self.snap, self.lock = 0, None, threading.Lock()

        def results_snapshot(self):
            assert self.snap is None, "second snapshot before the first was fetched"
            self.snap = self.k
            self.k += 1
            return (4, 64)

        def results_fetch(self, into, shape):
            res, buf, coff = into.fit(*shape)
            time.sleep(0.02)
            buf[:4] = self.snap                       # the batch number, as payload
            coff[-1] = self.snap
            self.snap = None
            return res, buf, coff

    h = H()
    f = _lib.ResultFetcher(h)
    assert f.after_run() is None
    seen = []
    for _ in range(5):
        prev = f.after_run()                          # returns batch k-1 while batch k is being fetched
        seen.append((int(prev[1][0]), int(prev[2][-1])))
        time.sleep(0.03)
        assert int(prev[1][0]) == seen[-1][0]         # still intact after the concurrent fetch of the next batch
    last = f.drain()
    assert seen == [(k, k) for k in range(5)] and int(last[1][0]) == 5 and f.drain() is None
    f.close()


def test_fasta_input_is_refused_with_a_message_not_a_hang(tmp_path, monkeypatch):
    """records without qualities end the run with the reader's message (the reference dies on qual = None, C3POa.py:167); the
    reader thread's error must reach run() and the worker of that range must still see the end marker -- a SystemExit raised
    inside the thread used to vanish and leave the device thread waiting for ever"""
    import threading
    import pytest
    recs = list(synth.generate("cfg1", n_reads=6))
    fa = str(tmp_path / "reads.fasta")
    with open(fa, "w") as fh:
        for r in recs:
            fh.write(">%s\n%s\n" % (r[0], r[1]))
    out = str(tmp_path / "out") + "/"
    os.makedirs(out)
    monkeypatch.setattr(_lib, "Handle", FakeHandle)
    args = types.SimpleNamespace(out_path=out, reads=fa, groupSize=1000, lencutoff=1000, mdistcutoff=500, zero=True, compress_output=False)
    sd = {"Splint1": [synth.SPLINT1, revcomp(synth.SPLINT1)]}
    box = {}

    def go():
        try:
            stream.run(args, sd, {r[0]: ["Splint1", r[3]] for r in recs}, {"Splint1"}, 1)
        except BaseException as e:                   # noqa: BLE001
            box["err"] = e
    th = threading.Thread(target=go, daemon=True)
    th.start()
    th.join(60)
    assert not th.is_alive(), "stream.run hangs on FASTA input"
    assert isinstance(box.get("err"), _lib.C3Error) and "FASTQ" in str(box["err"])


def test_pinned_result_buffers_keep_a_replaced_block_until_the_next_fit(monkeypatch):
    """a page-locked ResultBuffers that grows must not free the block its previous fit() handed out while a caller may still
    read it (ResultFetcher alternates two buffers): the predecessor is released by the NEXT fit() or by close()"""
    import ctypes as C
    rb = _lib.ResultBuffers(pinned=False)
    freed, blocks = [], []

    class L:
        @staticmethod
        def c3_host_alloc(nbytes, pp):
            b = C.create_string_buffer(int(nbytes))
            blocks.append(b)
            C.cast(pp, C.POINTER(C.c_void_p))[0] = C.addressof(b)
            return 0

        @staticmethod
        def c3_host_free(p):
            freed.append(p.value if hasattr(p, "value") else p)
    rb.pinned, rb.lib = True, L
    res, buf, coff = rb.fit(4, 64)
    first = rb._p["buf"].value
    buf[:4] = 7
    rb.fit(4, 4096)                                  # grows: the first block must survive this call
    assert first not in freed and bytes(buf[:4]) == b"\x07" * 4
    rb.fit(4, 64)                                    # the next fit releases it
    assert first in freed
    rb.close()
    assert len(freed) == len(blocks)
