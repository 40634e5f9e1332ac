#!/bin/bash
# round 3, call i: device tokenizer with per-column cell widths (mixed ploidy) and under freq.py -- whole GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
