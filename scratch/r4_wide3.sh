#!/bin/bash
mkdir -p gpurun_out/r04w
date +%T
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "row_chain" > gpurun_out/r04w/test_chain.log 2>&1; echo "chain tests rc=$?"; tail -3 gpurun_out/r04w/test_chain.log
timeout 300 python scratch/chain_bench.py wide > gpurun_out/r04w/chain_wide8.txt 2>&1; cat gpurun_out/r04w/chain_wide8.txt
timeout 300 python scratch/chain_bench.py nopf > gpurun_out/r04w/chain_narrow8.txt 2>&1; cat gpurun_out/r04w/chain_narrow8.txt
date +%T
timeout 400 python bench.py --steps 30 --warmup 8 --no-extra-workloads --no-cpu-baseline > gpurun_out/r04w/xe.json 2> gpurun_out/r04w/xe.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04w/xe.json").read().strip().splitlines()[-1]); print("xe ms", r["ms_per_step"], json.dumps(r["roofline"].get("chain_kernel")))
PY
timeout 400 python bench.py --workload decode --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04w/dec.json 2> gpurun_out/r04w/dec.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04w/dec.json").read().strip().splitlines()[-1]); print("decode ms", r["ms_per_step"], json.dumps(r["roofline"].get("chain_kernel")))
PY
date +%T
