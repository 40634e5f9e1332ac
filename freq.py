#!/usr/bin/env python
"""Drop-in for the reference's freq.py: per-site per-population base counts / target-allele frequencies, counted on an MI355X by
libpopgen_hip.so (k_site_counts).  See genomics_general_amd/cli.py."""
import sys

from genomics_general_amd.cli import freq_main

if __name__ == "__main__":
    sys.exit(freq_main())
