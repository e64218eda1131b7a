#!/bin/bash
mkdir -p gpurun_out/r04u
date +%T
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "unique_cache_rows or decoder_stack_kernel_vs" > gpurun_out/r04u/t.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r04u/t.log
date +%T
timeout 600 python bench.py --workload decode --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04u/dec.json 2> gpurun_out/r04u/dec.err; echo "bench rc=$?"; tail -3 gpurun_out/r04u/dec.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04u/dec.json").read().strip().splitlines()[-1]); print("ms", r["ms_per_step"]); print(json.dumps(r["roofline"])[:900])
PY
date +%T
