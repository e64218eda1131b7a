#!/bin/bash
# time the decode with each prebuilt library variant scratch/libortk_<SKB>_<XKB>.so (copied over the package's library)
L=sparse-image-captioning_amd/libortk.so
cp $L /tmp/libortk_keep.so
for v in scratch/libortk_*_*.so; do
  cp $v $L
  timeout 120 python bench.py --workload decode --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('$v', d['ms_per_step'])"
done
cp /tmp/libortk_keep.so $L
