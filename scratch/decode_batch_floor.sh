#!/bin/bash
# ms per beam-5 decode against the batch size: the floor at 8 images is the cost of ~1 300 dependent launches
for b in 8 50 128 256 1024; do
  python bench.py --workload decode --batch $b --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('batch', $b, d['ms_per_step'], d['value'])"
done
