#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/pytest_chain.log
tail -6 $O/pytest_chain.log
python3 scratch/chain_ab.py 2>&1 | grep -v amdgpu
