#!/bin/bash
mkdir -p gpurun_out/r04g
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "gather or sparse_weight_stream or sparse_decode or bf16_decode_executors" > gpurun_out/r04g/t.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r04g/t.log
for v in 988 988_scatter "" ; do
timeout 400 python bench.py --workload sparse_decode ${v:+--variant $v} --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04g/sd$v.json 2> gpurun_out/r04g/sd$v.err
python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/r04g/sd$v.json").read().strip().splitlines()[-1]); print("sparse_decode '$v' ms", r["ms_per_step"], r["roofline"].get("dominant_kernel",{}).get("avg_us"))
except Exception as e: print("'$v' failed:", open("gpurun_out/r04g/sd$v.err").read()[-800:])
PY
done
