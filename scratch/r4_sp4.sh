#!/bin/bash
mkdir -p gpurun_out/r04s
for v in 988 988_kernels 988 988_kernels "" kernels; do
timeout 400 python bench.py --workload sparse_xe ${v:+--variant $v} --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04s/x$v.json 2> gpurun_out/r04s/x$v.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04s/x$v.json").read().strip().splitlines()[-1]); print("sparse_xe '$v' ms", r["ms_per_step"])
PY
done
timeout 1200 python -m pytest tests -x -q -m gpu -k "sparse or supermask or spmm or prune" > gpurun_out/r04s/t4.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04s/t4.log
