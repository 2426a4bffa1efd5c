#!/bin/bash
# usage: tools/sweep.sh "CHAINS VARIANT [PROPOSALS_PER_LAUNCH]" ...   -- one bench.py line per config (GPU box)
for cfg in "$@"; do
  set -- $cfg
  M=${3:-1}; timeout 120 python bench.py --steps 100 --warmup 10 --chains $1 --variant $2 --proposals-per-launch ${3:-1} --rng ${4:-philox} --no-cpu-baseline --no-ess --no-sweep --no-config5 --no-config5 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chains %8s variant %s M ${3:-1} ${4:-philox}: %.3e steps/s  launch %8.1f us  mfma-frac %.3f' % ('$1','$2', d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
done
