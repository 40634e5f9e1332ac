#!/usr/bin/env python
"""Drop-in for the reference's ABBABABAwindows.py: same command line, `.geno` in, CSV out; ABBA / BABA / D / fd / fdM per
window computed on an MI355X by libpopgen_hip.so.  See genomics_general_amd/cli.py."""
import sys

from genomics_general_amd.cli import abbababa_main

if __name__ == "__main__":
    sys.exit(abbababa_main())
