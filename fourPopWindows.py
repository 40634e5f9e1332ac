#!/usr/bin/env python
"""Drop-in for the reference's fourPopWindows.py: same command line, `.geno` in, CSV out; the twelve four-population
statistics (ABBA, BABA, ABAA, BAAA, D, fd, fd', fdm, fdm', fdh, fdh2, fh) per window computed on an MI355X by
libpopgen_hip.so.  See genomics_general_amd/cli.py."""
import sys

from genomics_general_amd.cli import fourpop_main

if __name__ == "__main__":
    sys.exit(fourpop_main())
