#!/bin/bash
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "sparse_weight_stream or sparse_decode or bf16_decode_executors" > gpurun_out/r04d/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04d/t.log
for v in "" 988; do
timeout 400 python bench.py --workload sparse_decode ${v:+--variant $v} --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04d/sd$v.json 2> gpurun_out/r04d/sd$v.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04d/sd$v.json").read().strip().splitlines()[-1]); print("sparse_decode '$v' ms", r["ms_per_step"])
PY
done
