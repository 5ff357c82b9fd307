#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of bench.py and the PMC passes of tools/pmc_probe.py for ONE
# bench configuration.  Counters go in separate passes (TCC: FETCH_SIZE 3 slots, WRITE_SIZE 2; SQ: 8), never together with
# tracing; the profiled program itself comes after "--" (no wrapper).
# usage: bash tools/collect_profiles.sh <tag> [config]   -> gpurun_out/profiles_<tag>_<config>/ and profiles/<tag>_<config>_*
set -e
TAG=${1:-r3}
CFG=${2:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_${TAG}_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$n -- python3 $R/tools/pmc_probe.py $CFG > $OUT/pmc_$n.log 2>&1
  echo "pmc pass $n done"
done
cd $R && python3 tools/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.json
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic_$CFG.json
cp $OUT/pmc_summary.json $OUT/${TAG}_${CFG}_pmc_summary.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_${CFG}_kernel_stats.csv
echo done
