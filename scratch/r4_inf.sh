#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "decode or decoder_stack or sparse_weight or gather or beam or sample" > gpurun_out/r04i/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04i/t.log
for w in decode sparse_decode "sparse_decode --variant 988"; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04i/w.json 2> gpurun_out/r04i/w.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04i/w.json").read().strip().splitlines()[-1]); print("$w ms", r["ms_per_step"], r["roofline"].get("chain_kernel",{}).get("avg_us"))
PY
done
