#!/bin/bash
# A/B timing of library variants (development aid): bash tools/ab.sh name1 name2 ...  -> bench lines for build_ab/lib_<name>.so
# run ON THE GPU BOX through gpurun; "base" = the in-tree library
for v in "$@"; do
  if [ "$v" = base ]; then unset ILQR_HIP_LIB; else export ILQR_HIP_LIB=$PWD/build_ab/lib_$v.so; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu --no-extra ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], {k:round(v['ms_per_pass'],3) for k,v in d['kernels'].items()})" || echo "$v failed"
done
