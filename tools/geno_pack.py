#!/usr/bin/env python
"""Tokenise a `.geno(.gz)` file once into a packed `.pgeno` file (genomics_general_amd.genoio: one byte per diploid call), which
popgenWindows.py / ABBABABAwindows.py / fourPopWindows.py / distMat.py / freq.py read in place of the text.

    python tools/geno_pack.py -g in.geno.gz -o in.pgeno -f phased [--haploid s1,s2] [--ploidyFile f]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import genoio                                             # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-g", "--genoFile", required=True)
    ap.add_argument("-o", "--outFile", required=True)
    ap.add_argument("-f", "--genoFormat", default="phased", choices=("phased", "pairs", "haplo", "diplo", "alleles"))
    ap.add_argument("--haploid", help="comma separated haploid samples")
    ap.add_argument("--ploidyFile", help="sample <tab> ploidy per line")
    ap.add_argument("--header", help="header line if the file has none")
    ap.add_argument("--blockMiB", type=int, default=256, help="text bytes per block")
    ap.add_argument("--codec", default="zlib", choices=("zlib", "none"), help="deflate the blocks (default) or store them raw")
    a = ap.parse_args(argv)
    pl = {}
    if a.ploidyFile:
        with open(a.ploidyFile) as f:
            for line in f:
                w = line.split()
                if len(w) >= 2:
                    pl[w[0]] = int(w[1])
    if a.haploid:
        for nm in a.haploid.split(","):
            pl[nm] = 1
    n = genoio.pack_geno(a.genoFile, a.outFile, a.genoFormat, pl, a.header, a.blockMiB << 20, a.codec)
    sys.stderr.write("%d sites -> %s (%d bytes)\n" % (n, a.outFile, os.path.getsize(a.outFile)))


if __name__ == "__main__":
    main()
