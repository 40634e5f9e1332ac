#!/usr/bin/env python
"""Drop-in for simonhmartin/genomics_general VCF_processing/parseVCF.py (VCF -> .geno; `--packed` -> .pgeno)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomics_general_amd.vcf import parse_vcf_main  # noqa: E402

if __name__ == "__main__":
    rc = parse_vcf_main()
    # the outputs are closed; what is left is releasing gigabytes of page-locked and device memory one buffer at a time (0.1 - 0.2 s
    # of a run of under a second): the process ends here and the driver takes it all back at once
    sys.stdout.flush()
    sys.stderr.flush()
    # (not under a profiler or a sanitizer, which write their reports when the process ends the long way; PG_FAST_EXIT=0: never)
    if os.environ.get("PG_FAST_EXIT", "1") != "0" and not any(k.startswith(("ROCP", "ROCTRACER", "LD_PRELOAD", "ASAN_")) for k in os.environ):
        os._exit(rc or 0)
    sys.exit(rc)
