#!/usr/bin/env python
"""Drop-in for the reference's distMat.py: same command line, `.geno` in, distance matrices out (raw / phylip / nexus);
pairwise distances computed on an MI355X by libpopgen_hip.so.  See genomics_general_amd/cli.py."""
import sys

from genomics_general_amd.cli import distmat_main

if __name__ == "__main__":
    sys.exit(distmat_main())
