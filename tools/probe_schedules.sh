# Development aid (run on the GPU box): line-search round schedules (ILQR_LS_SCHEDULE) on the large-batch configurations.
for s in "4,4,8,16,16,16" "4,4,8,8,8,8,8,8,8" "4,4,8,16,8,8,8,8" "8,8,16,16,16" "4,4,4,4,8,8,8,8,8,8" "2,2,4,8,16,16,16" "4,4,8,8,8,16,16"; do
  for c in ${CONFIGS:-acrobot_big c4 pendulum}; do
    ILQR_LS_SCHEDULE=$s timeout -k 10 200 python bench.py --no-extra --no-cpu --steps 2 --warmup 1 --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d['value']), d['ms_per_step'], d['kernels']['linesearch']['ms_per_pass'])" "$s" $c || exit 1
  done
done
