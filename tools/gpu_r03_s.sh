#!/bin/bash
# round 3, call s: the cat window on several ranks, the counts-supplied finaliser, then the whole GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5; grep -E "^E " $O/pytest.log | head -20
