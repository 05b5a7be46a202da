"""Streaming driver of the CLI: native FASTQ reader -> GPU batches -> native writer.

Replaces the read loop / mp.Pool dispatch / per-group temp files of /root/reference/C3POa.py:236-271 with a software
pipeline that keeps every GPU busy: while batch i runs on the device (its worker thread blocks in c3_batch_run, which
releases the GIL), other threads parse batch i+1, copy it to the device (c3_batch_stage) and write the records of batch i-1.  No per-read Python objects are created for sequences or qualities: the
reader fills page-locked SoA buffers that c3_batch_upload copies by DMA and c3_write_group slices for the subread file.

The reference's -g (group size) only decides how reads are partitioned into <splint>/tmp<k>/ directories that are
concatenated and deleted at the end (C3POa.py:259-271); here a GPU batch is max(-g, GPU_BATCH_READS) reads (or
C3_GPU_BATCH_READS when set) and records are appended to the final files directly (same output tree, no double write).
"""
import os
import queue
import sys
import threading
import time

import numpy as np

from . import _lib

GPU_BATCH_READS = 131072         # enough reads to fill 256 CUs many times over (bench.py runs 100 000 per step)
HANDLES_PER_GPU = 1              # measured: two handles per GPU make the persistent kernels fight for CUs (0.55x)
GPU_BATCH_BASES = 1 << 30        # bound on host/device memory per batch
READERS_PER_GPU = 2              # byte-range readers (parser threads) per GPU worker
MIN_RANGE_BYTES = 256 << 20      # no point in cutting small inputs


def count_reads(path, lencutoff, assigner):
    """first pass of C3POa.py:200-207 + the bookkeeping of bin/preprocess.py:36-44: (reads passing the length cut-off,
    short reads, reads without a splint).  Names-only parse, native lookups."""
    rd = _lib.Reader(path, n_sets=1, names_only=True)
    total = short = assigned = 0
    while True:
        hb = rd.next(GPU_BATCH_READS, lencutoff, GPU_BATCH_BASES)
        short += hb.n_short
        if hb.n == 0:
            break
        total += hb.n
        assigned += assigner.batch(hb)[2]
    rd.close()
    return total, short, total - assigned


class DictAssigner:
    """adapter for callers that hold the reference's adapter_dict {name: [splint, strand]} (tests, stage seams)"""

    def __init__(self, adapter_dict, splint_names):
        self.d, self.sid_of = adapter_dict, {n: i for i, n in enumerate(splint_names)}

    def batch(self, hb):
        sid = np.full(hb.n, -1, dtype=np.int16)
        st = bytearray(b"?" * hb.n)
        k = 0
        for i, name in enumerate(hb.names()):
            ad = self.d.get(name)
            if ad:
                sid[i] = self.sid_of[ad[0]]; st[i] = 45 if ad[1] == "-" else 43; k += 1
        return sid, bytes(st), k


_KEPT = []


def run(args, splint_dict, assigner, adapter_set=None, n_dev=1, stats=None, finder_psl=None, keep_pinned=False):
    """the second pass of C3POa.py:236-271.  assigner: _lib.Assigner (native PSL table) or an adapter_dict.
    assigner=None is the fused mode: no PSL exists yet, so every batch is first scored against all splints on both
    strands (c3_scan_splints), assigned on the device (c3_batch_assign) and its PSL rows appended to finder_psl -- one pass
    over the reads instead of finder pass + counting pass + consensus pass.  returns the number of reads sent to the GPUs."""
    splint_names = sorted(splint_dict)
    if isinstance(assigner, dict):
        assigner = DictAssigner(assigner, splint_names)
    compress = bool(getattr(args, "compress_output", False))
    cons_paths = [args.out_path + n + "/R2C2_Consensus.fasta" for n in splint_names]
    sub_paths = [args.out_path + n + "/R2C2_Subreads.fastq" for n in splint_names]
    fused = assigner is None
    used = set(adapter_set or ())                                             # cat_files runs per adapter_set entry (C3POa.py:259)
    if fused:
        used = set(splint_names)                    # not known yet: directories of splints nobody used are removed at the end
        open(finder_psl + ".part", "w").close()
        sp_lens = [len(splint_dict[n][0]) for n in splint_names]
    if isinstance(assigner, DictAssigner):
        used |= set(v[0] for v in assigner.d.values())
    for n, cp, sp in zip(splint_names, cons_paths, sub_paths):
        if n in used:
            os.makedirs(args.out_path + n, exist_ok=True)
            for p in (cp, sp):                                   # "w+" semantics of cat_files (C3POa.py:88-92)
                open(p, "w").close()
                if os.path.exists(p + ".gz"):
                    os.remove(p + ".gz")
    # C3_GPU_BATCH_READS overrides the GPU batch size (tests drive the multi-batch pipeline with small inputs)
    gpu_batch = int(os.environ.get("C3_GPU_BATCH_READS", "0"))
    batch_reads = gpu_batch if gpu_batch > 0 else max(int(args.groupSize), GPU_BATCH_READS)
    n_work = n_dev * int(os.environ.get("C3_HANDLES_PER_GPU", HANDLES_PER_GPU))
    # ---- the input is cut into contiguous BYTE RANGES, one native reader (thread) per range, READERS_PER_GPU ranges per
    # worker: every GPU parses its own part of the file ("contiguous file ranges per GPU"), and one parser
    # thread (~2 GB/s of FASTQ) no longer caps the node.  All workers append to the same output files (c3_write_group
    # reserves each group's byte range under a lock): no per-group temp files, no merge pass (C3POa.py:259-271 concatenates
    # <splint>/tmp<k>/ directories in glob order -- the reference's record order is arbitrary as well; here it is the
    # round-robin order of the ranges, and the file order itself with a single range).
    size = os.path.getsize(args.reads)
    splittable = not str(args.reads).endswith(".gz") and size > 0
    if str(args.reads).endswith(".gz") and os.environ.get("C3_BGZF_RANGES") and not os.environ.get("C3_NO_BGZF"):
        # BGZF (bgzip) input CAN be cut into ranges of its inflated bytes like a plain file (c3_bgzf_size, c3_reader_open_range) -- opt-in:
        # on the 16-core quota of the GPU boxes it does not pay (one reader with eight inflating threads feeds 120-150 k reads/s, the
        # inflating itself is ~10 core-seconds per million reads; 2 / 4 / 8 ranges: 122 / 110 / 75 k, profiles/r05_host_ceiling_bgzf_ranges.txt)
        isz = _lib.bgzf_size(args.reads)
        if isz > 0:
            size, splittable = isz, True
    per_gpu = max(1, int(os.environ.get("C3_READERS_PER_GPU", READERS_PER_GPU)))
    min_range = int(os.environ.get("C3_MIN_RANGE_BYTES", MIN_RANGE_BYTES))
    n_ranges = max(1, min(n_work * per_gpu, size // max(min_range, 1))) if splittable else 1
    cuts = [size * k // n_ranges for k in range(n_ranges)] + [-1]
    if gpu_batch <= 0:
        batch_reads = max(batch_reads // max(1, n_ranges // n_work), 16384)       # same page-locked footprint per GPU as one reader
    _lib.load().c3_writer_reset()
    # Ownership of the host buffers is explicit: a reader fills buffer set j only after taking j from its free list, and the
    # writer puts j back once the group has been written (device threads finish out of order, so a counting semaphore over
    # round-robin sets would let a reader overwrite a set a lagging device still holds).  Result buffers follow the same
    # rule: one object per batch in flight, returned by the writer.
    N_SETS = 5                                      # (a sixth set would cover every stage at once; it costs 20 % more page-locked memory and start-up time for a stall that the timeline does not show)
    readers = [_lib.Reader(args.reads, n_sets=N_SETS, byte_range=(cuts[k], cuts[k + 1])) if n_ranges > 1
               else _lib.Reader(args.reads, n_sets=N_SETS) for k in range(n_ranges)]
    if n_ranges > 1 and any(r.range_lost() for r in readers):
        # a range with bytes but no 4-line record start (multi-line FASTQ, which mm.fastx_read accepts): such a file cannot be
        # entered in the middle -- its records would be dropped silently -- so ONE reader takes the whole file
        for r in readers:
            r.close()
        n_ranges, cuts = 1, [0, -1]
        readers = [_lib.Reader(args.reads, n_sets=N_SETS)]
    # (last in, first out: a set that has just come back is the one refilled next, so a reader only ever allocates -- and page-locks -- as many
    # sets as it really has in flight at once; first in, first out walked through all five even when two would do)
    free_sets = [queue.LifoQueue() for _ in range(n_ranges)]
    for fs in free_sets:
        for j in reversed(range(N_SETS)):
            fs.put(j)
    free_results = queue.Queue()
    for _k in range(4 * n_work + 2):                # run -> fetch -> (queue of 2) -> write: four batches per worker can hold one
        free_results.put(_lib.ResultBuffers())
    parsed = [queue.Queue(maxsize=1) for _ in range(n_ranges)]
    to_write = [queue.Queue(maxsize=2) for _ in range(n_work)]
    to_fetch = [queue.Queue(maxsize=1) for _ in range(n_work)]
    fetched = [threading.Event() for _ in range(n_work)]
    for ev_ in fetched:
        ev_.set()
    t = dict(parse=0.0, assign=0.0, upload=0.0, upload_dev=0.0, run=0.0, run_c=0.0, run_dev=0.0, alloc=0.0, fetch=0.0, snapshot=0.0, write=0.0, wait_in=0.0, wait_out=0.0,
             setup=0.0, close=0.0, scan=0.0, reads=0, batches=0, short=0, assigned=0, ranges=n_ranges)
    seen = set()
    errors, lock = [], threading.Lock()

    def reader_thread(k):                           # parse + splint/strand lookup of range k, ahead of its GPU
        rd = readers[k]
        ks = None                                   # the buffer set this thread holds and has not handed on
        try:
            while not errors:
                ks = free_sets[k].get()
                t0 = time.perf_counter()
                hb = rd.next(batch_reads, args.lencutoff, GPU_BATCH_BASES, set_index=ks)
                t1 = time.perf_counter()
                if rd.noqual():
                    # C3POa.py:167 averages ord(q) - 33 over the read's quality string and racon is run with -q 5: records
                    # without qualities cannot go through the reference either (it raises on qual = None).  An ordinary
                    # exception, so that it lands in `errors` and run() re-raises it (SystemExit would end this thread silently)
                    raise _lib.C3Error("C3POa: %s holds records without base qualities (FASTA); the consensus caller needs FASTQ" % args.reads)
                if hb.n == 0:
                    with lock:
                        t["short"] += hb.n_short
                    free_sets[k].put(ks); ks = None
                    break
                if fused:
                    sid, st, na = np.zeros(hb.n, dtype=np.int16), b"?" * hb.n, 0
                else:
                    sid, st, na = assigner.batch(hb)
                with lock:
                    t["parse"] += t1 - t0; t["assign"] += time.perf_counter() - t1
                    t["reads"] += hb.n; t["batches"] += 1; t["short"] += hb.n_short; t["assigned"] += na
                hb.range_index = k
                parsed[k].put((hb, sid, st)); ks = None
        except BaseException as e:                  # noqa: BLE001 -- re-raised by the caller's thread
            errors.append(e)
        finally:                                    # whatever happened: the worker of this range must see the end marker
            if ks is not None:
                free_sets[k].put(ks)
            parsed[k].put(None)

    def device_thread(w):                           # one per GPU: c3_batch_run sizes its stages on the host, so it blocks
        dev = w % n_dev
        if os.environ.get("C3_DEVICE_MAP"):             # test hook: worker -> physical device (two workers on one GPU)
            dmap = [int(x) for x in os.environ["C3_DEVICE_MAP"].split(",")]
            dev = dmap[dev % len(dmap)]
        mine = [k for k in range(n_ranges) if k % n_work == w]      # the ranges this worker serves, round-robin
        live = list(mine)
        rr = [0]

        def take(block=True):
            """next parsed batch of this worker's ranges in STRICT round-robin (the record order of a run does not depend on
            thread timing); None when all ranges are exhausted; block=False raises queue.Empty when the range whose turn it is
            has nothing ready"""
            tw = time.perf_counter()
            try:
                while live:
                    k = live[rr[0] % len(live)]
                    item = parsed[k].get() if block else parsed[k].get_nowait()
                    if item is None:
                        i = live.index(k)
                        live.remove(k)
                        rr[0] = i                                     # the next range moved into this position
                        continue
                    rr[0] = (live.index(k) + 1) % len(live)
                    return item
                return None
            finally:
                with lock:
                    t["wait_in"] += time.perf_counter() - tw

        try:
            ts = time.perf_counter()
            h = _lib.Handle(device=dev, mdistcutoff=args.mdistcutoff, zero=1 if getattr(args, "zero", True) else 0)
            h.set_splints([splint_dict[n][0] for n in splint_names])
            with lock:
                t["setup"] += time.perf_counter() - ts
                t["at_setup_done"] = time.perf_counter() - t_start
            # software pipeline on one handle: while batch i runs, batch i+1 (if a reader already has it) is copied
            # and 2-bit packed on the handle's second stream (c3_batch_stage); c3_batch_commit makes it resident once
            # the results of batch i have been fetched
            cur = take()
            t["at_first_batch"] = time.perf_counter() - t_start
            if cur is not None and not errors:
                t0 = time.perf_counter()
                h.upload_host(cur[0], cur[2], np.maximum(cur[1], 0))
                with lock:
                    t["upload"] += time.perf_counter() - t0
            done = cur is None
            while not done and not errors:
                hb, sid, st = cur
                if fused:                                         # splint / strand of the resident batch from the GPU finder
                    ts = time.perf_counter()
                    tab, sid, st = h.scan_splints()
                    h.assign(sid, st)
                    with lock:
                        na = _lib.write_splint_psl(hb, tab, sid, st, splint_names, sp_lens, h.cfg.conk_match, finder_psl + ".part")
                        t["assigned"] += na; t["scan"] += time.perf_counter() - ts
                        seen.update(splint_names[x] for x in np.unique(sid[sid >= 0]))
                nxt, staged = None, False
                try:
                    nxt = take(False)
                    if nxt is not None:
                        t0 = time.perf_counter()
                        h.stage_host(nxt[0], nxt[2], np.maximum(nxt[1], 0))
                        staged = True
                        with lock:
                            t["upload"] += time.perf_counter() - t0
                    else:
                        done = True
                except queue.Empty:
                    pass
                t1 = time.perf_counter()
                h.run()
                t2 = time.perf_counter()
                up_dev = h.last_timing["ms_pack"] * 1e-3
                run_dev = h.last_timing["ms_total"] * 1e-3
                run_c = h.last_timing.get("ms_wall", 0.0) * 1e-3; wl_c = h.last_timing.get("ms_alloc", 0.0) * 1e-3
                # results: frozen on the device here (c3_batch_results_snapshot, ~0.3 ms); the worker's fetch thread copies them to
                # the host while this thread commits and runs the next batch.  One snapshot per handle: wait for the previous fetch
                # (it started a whole batch ago)
                fetched[w].wait()
                if errors:
                    break
                fetched[w].clear()
                shape = h.results_snapshot()
                t3 = time.perf_counter()
                with lock:
                    t["upload_dev"] += up_dev; t["run_dev"] += run_dev; t["run_c"] += run_c; t["alloc"] += wl_c
                    t["run"] += t2 - t1; t["snapshot"] += t3 - t2
                tw = time.perf_counter()
                to_fetch[w].put((h, hb, sid, shape))
                with lock:
                    t["wait_out"] += time.perf_counter() - tw
                if done:
                    break
                if nxt is None:                                   # no reader was ahead: wait for one now
                    nxt = take()
                    if nxt is None:
                        break
                t0 = time.perf_counter()
                if staged:
                    h.commit()
                else:
                    h.upload_host(nxt[0], nxt[2], np.maximum(nxt[1], 0))
                with lock:
                    t["upload"] += time.perf_counter() - t0
                cur = nxt
            tc = time.perf_counter()
            t["at_last_run_done"] = tc - t_start
            fetched[w].wait()                       # the last snapshot has been copied out
            h.close()
            with lock:
                t["close"] += time.perf_counter() - tc
        except Exception as e:                      # noqa: BLE001
            errors.append(e)
        while live:                                 # keep draining so no reader blocks on a full queue
            try:
                take()
            except Exception:                       # noqa: BLE001
                break
        to_fetch[w].put(None)

    def fetch_thread(w):                            # c3_batch_results_fetch of batch i beside the kernels of batch i+1
        while True:
            item = to_fetch[w].get()
            if item is None:
                break
            h, hb, sid, shape = item
            rb = free_results.get()
            t0 = time.perf_counter()
            try:
                res, buf, coff = h.results_fetch(rb, shape)
            except Exception as e:                  # noqa: BLE001
                errors.append(e)
                fetched[w].set()
                free_results.put(rb)                # the buffers of a batch that is lost go back to their owners
                free_sets[hb.range_index].put(hb.set_index)
                continue
            fetched[w].set()
            with lock:
                t["fetch"] += time.perf_counter() - t0
            del h, item
            to_write[w].put((hb, sid, res, buf, coff, rb))
        to_write[w].put(None)

    def writer_thread(w):                           # c3_write_group releases the GIL: overlaps parsing and the GPUs
        while True:
            item = to_write[w].get()
            if item is None:
                break
            hb, sid, res, buf, coff, rb = item
            t0 = time.perf_counter()
            try:
                if not errors:
                    _lib.write_group(hb, res, buf, coff, sid, cons_paths, sub_paths, getattr(args, "zero", True))
            except Exception as e:                  # noqa: BLE001
                errors.append(e)
            with lock:
                t["write"] += time.perf_counter() - t0
            k, j = hb.range_index, hb.set_index
            del hb, item, res, buf, coff
            free_results.put(rb)
            free_sets[k].put(j)

    t_start = time.perf_counter()
    rthreads = [threading.Thread(target=reader_thread, args=(k,), daemon=True) for k in range(n_ranges)]
    others = [threading.Thread(target=writer_thread, args=(w,), daemon=True) for w in range(n_work)]
    others += [threading.Thread(target=fetch_thread, args=(w,), daemon=True) for w in range(n_work)]
    others += [threading.Thread(target=device_thread, args=(w,), daemon=True) for w in range(n_work)]
    for th in rthreads + others:
        th.start()
    for th in others:
        th.join()
    t["at_workers_done"] = time.perf_counter() - t_start
    if errors:
        raise errors[0]
    for th in rthreads:
        th.join()
    if keep_pinned:                                 # one-shot process (the CLI): process teardown releases the page-locked buffers
        _KEPT.extend(readers)
    else:
        for rd in readers:
            rd.close()
    t["at_readers_closed"] = time.perf_counter() - t_start
    if fused:
        os.replace(finder_psl + ".part", finder_psl)              # a rerun finds the PSL and takes the two-pass route
        for n, cp, sp in zip(splint_names, cons_paths, sub_paths):
            if n not in seen:                                       # adapter_set of bin/preprocess.py:34,43 = splints that were hit
                for p in (cp, sp):
                    if os.path.exists(p) and os.stat(p).st_size == 0:
                        os.remove(p)
                try:
                    os.rmdir(args.out_path + n)
                except OSError:
                    pass
        t["adapter_set"] = sorted(seen)
    if compress:                                                        # -co (C3POa.py:88-90)
        # every core deflates (c3_compress_file: independent gzip members in the BGZF layout -- any gzip reader takes the file, this
        # package's reader inflates it in parallel); the reference pushes the whole output through ONE gzip.open stream, line by line
        for p in cons_paths + sub_paths:
            if os.path.exists(p):
                _lib.compress_file(p, p + ".gz", level=6)
    if stats is not None:
        stats.update(t)
    if os.environ.get("C3_STREAM_STATS"):
        print("stream: " + " ".join("%s=%.3f" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v) for k, v in t.items()), file=sys.stderr)
    return t["reads"]
