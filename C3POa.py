#!/usr/bin/env python3
"""C3POa.py -- drop-in CLI of the MI355X-native R2C2 consensus caller.

Same flags, inputs and output tree as rvolden/C3POa v2.2.3 (/root/reference/C3POa.py:26-63, 175-272):
    python3 C3POa.py -r reads.fastq -s splint.fasta -o out -c config -l 1000 -d 500 -n N -g 1000 [-z] [-b] [-co]
writes <out>/c3poa.log, <out>/tmp/splint_to_read_alignments.psl (reused when present),
<out>/<splint>/R2C2_Consensus.fasta[.gz] and <out>/<splint>/R2C2_Subreads.fastq[.gz].

Differences, all deliberate: the per-read hot path runs on the GPU(s) through
libc3poa_hip.so instead of conk/pyabpoa/mappy/racon; -n selects how many GPUs share the groups;
the last, short group is flushed (the reference never dispatches it);
racon/blat config entries are accepted, racon is never executed.
"""
import argparse
import os
import sys

PATH = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, PATH)

from c3poa_amd import VERSION  # noqa: E402
from c3poa_amd.seqio import fastx_read, revcomp  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Makes consensus sequences from R2C2 reads.", add_help=True, prefix_chars="-")
    parser.add_argument("--reads", "-r", type=str, action="store", help="FASTQ file that contains the long R2C2 reads.")
    parser.add_argument("--splint_file", "-s", type=str, action="store", help="Path to the splint FASTA file.")
    parser.add_argument("--out_path", "-o", type=str, action="store", default=os.getcwd(),
                        help="Directory where all the files will end up. Defaults to your current directory.")
    parser.add_argument("--config", "-c", type=str, action="store", default="",
                        help="Config file with paths to racon and blat (tab separated).")
    parser.add_argument("--lencutoff", "-l", type=int, action="store", default=1000,
                        help="Sets the length cutoff for your raw sequences. Defaults to 1000.")
    parser.add_argument("--mdistcutoff", "-d", type=int, action="store", default=500,
                        help="Sets the median distance cutoff for consensus sequences. Defaults to 500.")
    parser.add_argument("--zero", "-z", action="store_false", default=True,
                        help="Use to exclude zero repeat reads. Defaults to True (includes zero repeats).")
    parser.add_argument("--numThreads", "-n", type=int, default=1, help="Number of GPUs (workers) to use. Defaults to 1.")
    parser.add_argument("--groupSize", "-g", type=int, default=1000,
                        help="Number of reads processed by each worker in each iteration. Defaults to 1000.")
    parser.add_argument("--splint-finder", dest="splint_finder", choices=["gpu", "blat"], default="gpu",
                        help="How reads are assigned to splints when the PSL does not exist yet: the GPU finder "
                             "(default) or the blat binary of the config file, as upstream.")
    parser.add_argument("--blatThreads", "-b", action="store_true", default=False, help="Accepted for compatibility.")
    parser.add_argument("--compress_output", "-co", action="store_true", default=False,
                        help="Use to compress (gzip) both the consensus fasta and subread fastq output files.")
    parser.add_argument("--version", "-v", action="version", version=VERSION, help="Prints the C3POa version.")
    if argv is None and len(sys.argv) == 1:
        parser.print_help()
        sys.exit(0)
    return parser.parse_args(argv)


def configReader(path, configIn):
    """C3POa.py:65-84: tab separated name<TAB>path, '#' comments, missing keys fall back to PATH"""
    progs = {}
    with open(configIn) as f:
        for line in f:
            if line.startswith("#") or not line.rstrip().split():
                continue
            line = line.rstrip().split("\t")
            progs[line[0]] = line[1]
    for missing in set(["racon", "blat"]) - set(progs):
        progs[missing] = missing
        sys.stderr.write("Using " + str(missing) + " from your path, not the config file.\n")
    return progs


_WARMERS = []            # context-warming threads of main(); joined before a one-shot process leaves


def early_warm(n_threads):
    """The HIP context of every GPU in use takes 1.1-1.3 s to create and everything else waits for it: start it on threads of
    their own BEFORE numpy and the package are imported (0.3-0.5 s), with nothing but ctypes -- the library is loaded here and
    found loaded by c3poa_amd._lib later.  Returns silently when there is no library or no GPU (main() reports that)."""
    import ctypes
    import threading
    lib_path = os.environ.get("C3POA_LIB", os.path.join(PATH, "c3poa_amd", "lib", "libc3poa_hip.so"))
    try:
        lib = ctypes.CDLL(lib_path)
        lib.c3_warm_device.argtypes = [ctypes.c_int]
    except (OSError, AttributeError):
        return
    if os.environ.get("C3_DEVICE_MAP"):
        devs = sorted(set(int(x) for x in os.environ["C3_DEVICE_MAP"].split(",")))
    else:
        devs = list(range(max(1, n_threads)))                            # (an ordinal beyond the visible GPUs returns an error code at once)
    for dev in devs:
        th = threading.Thread(target=lib.c3_warm_device, args=(dev,), daemon=True)
        th.start()
        _WARMERS.append(th)


def main(args, one_shot=False):
    """one_shot: the process ends right after this call (the command line): the page-locked reader buffers are then left to
    process teardown instead of being unpinned one by one (1.2 s per 5 GB)"""
    if not args.out_path.endswith("/"):
        args.out_path += "/"
    os.makedirs(args.out_path, exist_ok=True)
    log_file = open(args.out_path + "c3poa.log", "w+")
    if args.config:
        progs = configReader(args.out_path, args.config)
        racon, blat = progs["racon"], progs["blat"]
    else:
        racon, blat = "racon", "blat"
    tmp_dir = args.out_path + "tmp/"
    os.makedirs(tmp_dir, exist_ok=True)

    from c3poa_amd import stream, _lib
    from c3poa_amd.preprocess import ensure_psl
    import time
    t_main = [time.perf_counter()]
    splint_dict = {}
    for splint in fastx_read(args.splint_file):
        splint_dict[splint[0]] = [splint[1], revcomp(splint[1])]
    n_dev = max(1, min(max(1, args.numThreads), _lib.device_count()))        # -n = GPUs that share the groups
    if os.environ.get("C3_DEVICE_MAP"):                                      # test hook: worker w -> device map[w] (e.g. "0,0": the -n 2
        n_dev = max(1, min(max(1, args.numThreads), len(os.environ["C3_DEVICE_MAP"].split(","))))      # plumbing on a one-GPU box)
    # the HIP context of every device in use is created NOW, on threads of its own, beside the PSL load / the first parse
    # (1.2-1.9 s each; the workers' handles then find their context ready)
    import threading
    dmap_ = [int(x) for x in os.environ["C3_DEVICE_MAP"].split(",")] if os.environ.get("C3_DEVICE_MAP") else list(range(n_dev))

    if not _WARMERS:                                                         # (the command line has started them before its imports: early_warm)
        warmers = [threading.Thread(target=_lib.warm_device, args=(dev,), daemon=True) for dev in sorted(set(dmap_[:n_dev]))]
        for th in warmers:                                                   # (context only: a failure is reported by the worker's own c3_create)
            th.start()
        _WARMERS.extend(warmers)
    align_psl = tmp_dir + "splint_to_read_alignments.psl"
    have_psl = os.path.exists(align_psl) and os.stat(align_psl).st_size > 0
    if not have_psl and getattr(args, "splint_finder", "gpu") == "gpu":
        # no PSL yet: ONE pass over the reads -- every batch is scored against all splints on both strands on the GPU,
        # assigned, processed; the PSL (written on the way) makes a rerun take the route below
        print("Assigning splints to reads on the GPU", file=sys.stderr)
        st = {}
        stream.run(args, splint_dict, None, None, n_dev, stats=st, finder_psl=align_psl, keep_pinned=one_shot)
        total_reads, short_reads, no_splint = st["reads"], st["short"], st["reads"] - st["assigned"]
        t_main += [time.perf_counter()] * 3
    else:
        # splint / strand per read from the PSL (bin/preprocess.py:12-45).  The PSL is reused when it exists (or written by
        # blat); it is held in a native name -> (splint, strand) table.  The reference's first pass over the reads
        # (C3POa.py:200-207) only counts them for the log: those counts fall out of the one streaming pass below.
        assigner = _lib.Assigner(ensure_psl(blat, args, tmp_dir), sorted(splint_dict))
        t_main.append(time.perf_counter())
        adapter_set, _rows = assigner.seen()
        for adapter in adapter_set:
            os.makedirs(args.out_path + adapter, exist_ok=True)
        t_main.append(time.perf_counter())
        # streaming pipeline: native readers -> GPU batches -> native writers (c3poa_amd/stream.py); the tail group is
        # processed too (deliberate fix of SURVEY.md App. A.12)
        st = {}
        stream.run(args, splint_dict, assigner, adapter_set, n_dev, stats=st, keep_pinned=one_shot)
        total_reads, short_reads, no_splint = st["reads"], st["short"], st["reads"] - st["assigned"]
        if not one_shot:
            assigner.close()
        t_main.append(time.perf_counter())

    all_reads = total_reads + short_reads
    print("C3POa version:", VERSION, file=log_file)
    print("Total reads:", all_reads, file=log_file)
    if all_reads:
        print("No splint reads:", no_splint, "({:.2f}%)".format((no_splint / all_reads) * 100), file=log_file)
        print("Under len cutoff:", short_reads, "({:.2f}%)".format((short_reads / all_reads) * 100), file=log_file)
        print("Total thrown away reads:", short_reads + no_splint,
              "({:.2f}%)".format(((short_reads + no_splint) / all_reads) * 100), file=log_file)
    print("Reads after preprocessing:", all_reads - (short_reads + no_splint), file=log_file)
    log_file.close()
    if os.environ.get("C3_STREAM_STATS"):
        print("main: psl=%.3f count=%.3f consensus=%.3f" % tuple(b - a for a, b in zip(t_main, t_main[1:4])), file=sys.stderr)


if __name__ == "__main__":
    args = parse_args()
    if not args.reads or not args.splint_file:
        print("Reads (--reads/-r) and splint (--splint_file/-s) are required", file=sys.stderr)
        sys.exit(1)
    if not os.environ.get("C3_NO_EARLY_WARM"):          # (A/B hook of tools/cli_throughput.py)
        early_warm(args.numThreads)
    try:
        main(args, one_shot=True)
    finally:
        # never leave -- normally, by sys.exit or by an exception (a refused FASTA input, a bad path) -- while a thread is still inside
        # HIP initialisation: the interpreter's and the runtime's teardown beside it can hang or crash a run that should end with rc 1
        for th_ in _WARMERS:
            th_.join()
    # every output file has been written and closed; skip the interpreter's and the HIP runtime's teardown (unpinning the reader
    # buffers, freeing the device scratch: ~1.5 s that produce nothing)
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)
