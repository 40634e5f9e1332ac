#!/usr/bin/env python
"""Drop-in for simonhmartin/genomics_general VCF_processing/parseVCF.py (VCF -> .geno; `--packed` -> .pgeno)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomics_general_amd.vcf import parse_vcf_main  # noqa: E402

if __name__ == "__main__":
    sys.exit(parse_vcf_main())
