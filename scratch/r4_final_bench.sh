#!/bin/bash
mkdir -p gpurun_out/r04f
date +%T
time (timeout 900 python bench.py --steps 50 --warmup 10 > gpurun_out/r04f/r04_default_bench_line.json 2> gpurun_out/r04f/bench.err); echo "bench rc=$?"
wc -c gpurun_out/r04f/r04_default_bench_line.json; tail -3 gpurun_out/r04f/bench.err
date +%T
