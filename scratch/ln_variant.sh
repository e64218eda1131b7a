#!/bin/bash
# ln_bench with the package library, then with each scratch/libortk_*.so variant
L=sparse-image-captioning_amd/libortk.so
cp $L /tmp/libortk_keep.so
echo "== current"; python scratch/ln_bench.py 2>&1 | grep rows
for v in scratch/libortk_*.so; do cp $v $L; echo "== $v"; python scratch/ln_bench.py 2>&1 | grep rows; python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-125; done
cp /tmp/libortk_keep.so $L
