#!/bin/bash
# A/B of library variants over several configurations (development aid, run ON THE GPU BOX through gpurun):
#   bash tools/ab_configs.sh "c2 acrobot_big c4" base nomask   -> bench lines for the in-tree library and build_ab/lib_nomask.so
cfgs=$1; shift
for c in $cfgs; do
  AB_ARGS="--config $c" bash tools/ab.sh "$@" || exit 1
done
