#!/bin/bash
# Development aid (GPU box): the headline bench N times in a row; stops at the first run that fails; prints value and ms per step.
N=${1:-10}
for i in $(seq 1 $N); do
  timeout -k 10 120 python bench.py --no-extra --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', sys.argv[1], d['value'], d['ms_per_step'], d['exit_status_counts'])" $i || exit 1
done
