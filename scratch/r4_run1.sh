#!/bin/bash
# round 4, GPU call 1: full parity suite, SCST lines (reference estimator / dropout-free), crossover microbench
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4/pytest_full.log
python bench.py --workload scst --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4/scst.json 2> gpurun_out/r4/scst.err
python bench.py --workload scst --variant nodrop --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4/scst_nodrop.json 2>> gpurun_out/r4/scst.err
python scratch/spmm_crossover.py > gpurun_out/r4/spmm_crossover.txt 2>&1
tail -5 gpurun_out/r4/pytest_full.log; cat gpurun_out/r4/scst.json gpurun_out/r4/scst_nodrop.json | cut -c1-600; tail -3 gpurun_out/r4/scst.err
