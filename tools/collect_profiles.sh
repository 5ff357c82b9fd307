#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of bench.py and the PMC passes of tools/pmc_probe.py.
# Counters go in separate passes (TCC: FETCH_SIZE 3 slots, WRITE_SIZE 2; SQ: 8), never together with tracing.
# usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/profiles_<tag>/
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$n -- python3 $R/tools/pmc_probe.py > $OUT/pmc_$n.log 2>&1
done
# the traffic figures first (bench.py reports roofline.traffic from profiles/pmc_traffic.json when its source hash is the tree's)
cd $R && python3 tools/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.json
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo done
