#!/bin/bash
# round 3, call z3: k_pack3 with its plane stores staged in LDS and written in bursts: tests, then same-process A/B against PG_PACK_BURST=0
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^E |FAILED" $O/pytest.log | head
python tools/ab_env.py PG_PACK_BURST=0 northstar 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python tools/ab_env.py PG_PACK_BURST=0 c2 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
