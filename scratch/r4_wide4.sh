#!/bin/bash
mkdir -p gpurun_out/r04w
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "row_chain or xe_step or golden or valid_position or scst or supermask or train_mode" > gpurun_out/r04w/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04w/t.log
for i in 1 2; do
timeout 400 python bench.py --steps 30 --warmup 8 --no-extra-workloads --no-cpu-baseline > gpurun_out/r04w/xe.json 2> gpurun_out/r04w/xe.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04w/xe.json").read().strip().splitlines()[-1]); print("xe ms", r["ms_per_step"], r["roofline"].get("chain_kernel",{}).get("avg_us"))
PY
done
for w in scst "scst --variant nodrop" decode; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04w/w.json 2> gpurun_out/r04w/w.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04w/w.json").read().strip().splitlines()[-1]); print("$w ms", r["ms_per_step"])
PY
done
