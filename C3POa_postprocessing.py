#!/usr/bin/env python3
"""C3POa_postprocessing.py -- drop-in CLI of the post-processing step (trim / re-orient / demultiplex consensus reads).

Same flags, inputs and output files as rvolden/C3POa v2.2.3 (/root/reference/C3POa_postprocessing.py:17-62, 400-434):
    python3 C3POa_postprocessing.py -i R2C2_Consensus.fasta -a adapters.fasta -o out [-x indexes.fasta] [-u] [-t] [-b]
                                    [-n N] [-g 1000] [-bt] [-co] [-c config]
The adapter-to-read alignment that upstream delegates to blat runs on the GPU (c3_scan_adapters); the PSL file
<out>/adapter_to_consensus_alignment.psl is written and reused exactly as upstream reuses it.
"""
import argparse
import os
import sys

PATH = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, PATH)

from c3poa_amd import VERSION  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Reorients/demuxes/trims consensus reads.", add_help=True, prefix_chars="-")
    p.add_argument("--input_fasta_file", "-i", type=str, action="store", help="Fasta file with consensus called R2C2 reads")
    p.add_argument("--output_path", "-o", type=str, action="store", default=os.getcwd(),
                   help="Directory where all the files will end up. Defaults to your current directory.")
    p.add_argument("--adapter_file", "-a", type=str, action="store", help="Fasta file with adapter (3 and 5 prime) sequences")
    p.add_argument("--index_file", "-x", type=str, action="store", help="Fasta file with oligo dT indexes")
    p.add_argument("--config", "-c", type=str, action="store", default="", help="Accepted for compatibility (program paths).")
    p.add_argument("--undirectional", "-u", action="store_true",
                   help="cDNA molecules are undirectional: one sequence named 'Adapter' in the adapter file "
                        "(default: '3Prime_adapter' and '5Prime_adapter').")
    p.add_argument("--trim", "-t", action="store_true", help="Trim the adapters off the ends of the sequences.")
    p.add_argument("--barcoded", "-b", action="store_true", default=False,
                   help="10x reads: also writes a file with the 10x barcode sequences.")
    p.add_argument("--threads", "-n", type=int, default=1, help="Accepted for compatibility; > 1 selects upstream's multi-process output conventions.")
    p.add_argument("--groupSize", "-g", type=int, default=1000, help="Accepted for compatibility.")
    p.add_argument("--blatThreads", "-bt", action="store_true", default=False, help="Accepted for compatibility.")
    p.add_argument("--compress_output", "-co", action="store_true", default=False, help="gzip the output files (with -n > 1, as upstream).")
    p.add_argument("--adapter-finder", dest="adapter_finder", choices=["gpu"], default="gpu", help="How adapters are located (GPU local alignment).")
    p.add_argument("--version", "-v", action="version", version=VERSION, help="Prints the C3POa version.")
    if argv is None and len(sys.argv) == 1:
        p.print_help()
        sys.exit(0)
    return p.parse_args(argv)


def main(args):
    from c3poa_amd import postprocess
    return postprocess.run(args)


if __name__ == "__main__":
    args = parse_args()
    if not args.input_fasta_file or not args.adapter_file:
        print("Reads (--input_fasta_file/-i) and adapter (--adapter_file/-a) are required", file=sys.stderr)
        sys.exit(1)
    main(args)
