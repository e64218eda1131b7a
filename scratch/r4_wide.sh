#!/bin/bash
# the 76-row form of the forward chain kernel: parity, kernel times against the 48-row form, the step it sits in
mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "row_chain" > gpurun_out/r04w/test_chain.log 2>&1; echo "chain tests rc=$?"
tail -5 gpurun_out/r04w/test_chain.log
timeout 300 python scratch/chain_bench.py nopf > gpurun_out/r04w/chain_wide.txt 2>&1; cat gpurun_out/r04w/chain_wide.txt
timeout 300 python scratch/chain_bench.py narrow > gpurun_out/r04w/chain_narrow.txt 2>&1; cat gpurun_out/r04w/chain_narrow.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04w/prof_wide -o cw -- python3 $GRAFT_REPO_ROOT/scratch/chain_bench.py nopf > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04w/prof_wide -name "*kernel_stats.csv" | head -1); grep -E "row_chain|Name" $f | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 8 --no-extra-workloads --no-cpu-baseline > gpurun_out/r04w/xe.json 2> gpurun_out/r04w/xe.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04w/xe.json").read().strip().splitlines()[-1]); print("xe ms", r["ms_per_step"])
PY
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r04w/test_model.log 2>&1; echo "model tests rc=$?"; tail -5 gpurun_out/r04w/test_model.log
