#!/bin/bash
mkdir -p gpurun_out/r04w
date +%T
timeout 300 python scratch/chain_bench.py nopf > gpurun_out/r04w/chain_wide4.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04w/chain_wide4.txt
date +%T
timeout 400 python bench.py --steps 30 --warmup 8 --no-extra-workloads --no-cpu-baseline > gpurun_out/r04w/xe.json 2> gpurun_out/r04w/xe.err; echo "bench rc=$?"; tail -3 gpurun_out/r04w/xe.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04w/xe.json").read().strip().splitlines()[-1]); print("xe ms", r["ms_per_step"])
PY
date +%T
