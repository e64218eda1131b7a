#!/bin/bash
mkdir -p gpurun_out/r04s
timeout 1200 python -m pytest tests -x -q -m gpu -k "sparse or supermask or spmm or prune" > gpurun_out/r04s/t2.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04s/t2.log
timeout 900 python scratch/spmm_crossover.py > gpurun_out/r04s/crossover2.txt 2>&1; echo "cross rc=$?"; grep -E "sp=0.9(75|88)" gpurun_out/r04s/crossover2.txt
for v in 988 988_kernels; do
timeout 400 python bench.py --workload sparse_xe --variant $v --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04s/$v.json 2> gpurun_out/r04s/$v.err; echo "bench $v rc=$?"
python - <<PY
import json
r=json.loads(open("gpurun_out/r04s/$v.json").read().strip().splitlines()[-1]); print("$v ms", r["ms_per_step"])
PY
done
