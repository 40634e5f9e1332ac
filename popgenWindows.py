#!/usr/bin/env python
"""Drop-in for the reference's popgenWindows.py: same command line, `.geno` in, CSV out; pi / dxy / Fst (and popFreq,
indPairDist) per window computed on an MI355X by libpopgen_hip.so.  See genomics_general_amd/cli.py."""
import sys

from genomics_general_amd.cli import popgen_main

if __name__ == "__main__":
    sys.exit(popgen_main())
