#!/bin/bash
# Run ON THE GPU BOX (through gpurun): tools/collect_profiles.sh for every profiled configuration, one after the other, each
# step under its own time limit; stops at the first failure.  Summaries land in gpurun_out/final_profiles/ (copy them to
# profiles/ afterwards), the per-config traffic files in profiles/ directly.
# usage: bash tools/collect_all_profiles.sh <tag> [configs...]
TAG=${1:-r3}; shift
CFGS=${@:-c2 c3 c4 c5 pendulum acrobot_big}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final_profiles
for c in $CFGS; do
  timeout -k 10 900 bash $R/tools/collect_profiles.sh $TAG $c > $R/gpurun_out/final_profiles/collect_$c.log 2>&1 || { echo "collect $c failed"; tail -5 $R/gpurun_out/final_profiles/collect_$c.log; exit 1; }
  O=$R/gpurun_out/profiles_${TAG}_$c
  cp $O/${TAG}_${c}_pmc_summary.json $O/${TAG}_${c}_kernel_stats.csv $R/gpurun_out/final_profiles/
  cp $O/bench_under_rocprof.json $R/gpurun_out/final_profiles/${TAG}_${c}_bench_under_rocprof.json
  cp $R/profiles/pmc_traffic_$c.json $R/gpurun_out/final_profiles/
  echo "$c done"
done
