for s in "8,8,16,16,16" "8,1,2,4,8,16,32" "8,2,2,4,8,16,32" "8,4,4,8,16,32" "8,1,1,2,4,8,16,32"; do
ILQR_LS_SCHEDULE=$s timeout -k 10 120 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$s', d['value'], d['ms_per_step'], {k:round(v['ms_per_pass'],3) for k,v in d['kernels'].items()})" || exit 1
done
