#!/bin/bash
# round 3, call s: the whole GPU suite (cat and predefined windows on several ranks included)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5; grep -E "^E " $O/pytest.log | head -40
