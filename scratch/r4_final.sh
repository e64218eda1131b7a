#!/bin/bash
# end-of-round validation: the whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out/r04f
date +%T
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04f/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r04f/gpu_tests.log
date +%T
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
date +%T
time (timeout 900 python bench.py --steps 50 --warmup 10 > gpurun_out/r04f/r04_default_bench_line.json 2> gpurun_out/r04f/bench.err); echo "bench rc=$?"
wc -c gpurun_out/r04f/r04_default_bench_line.json
date +%T
