#!/bin/bash
mkdir -p gpurun_out/r04s
timeout 600 python -m pytest tests -x -q -m gpu -k "spmm or sparse_train or sparse_xe or supermask_step" > gpurun_out/r04s/t3.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04s/t3.log
timeout 900 python scratch/spmm_crossover.py > gpurun_out/r04s/crossover3.txt 2>&1; echo "cross rc=$?"; cat gpurun_out/r04s/crossover3.txt
