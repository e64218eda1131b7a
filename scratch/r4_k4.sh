#!/bin/bash
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_mode_sampling or scst" > gpurun_out/r04k/t.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04k/t.log
timeout 400 python bench.py --workload scst --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04k/scst.json 2> gpurun_out/r04k/scst.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04k/scst.json").read().strip().splitlines()[-1]); print("scst ms", r["ms_per_step"], json.dumps(r["roofline"].get("rollout_kernel"))[:300])
PY
