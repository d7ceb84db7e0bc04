#!/bin/bash
# GPU box: bench.py (no CPU baseline) against each given library; prints ms/step and roofline frac
for lib in "$@"; do
  RMX_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['parity']['lag_int_mismatches'])"
done
