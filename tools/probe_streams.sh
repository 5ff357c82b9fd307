for cfg in acrobot_big c4 pendulum c5; do for s in 0 2 4; do
timeout -k 10 120 python bench.py --config $cfg --streams $s --steps 3 --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg streams $s', d['value'], d['ms_per_step'], {k:round(v['ms_per_pass'],3) for k,v in d['kernels'].items()})" || exit 1
done; done
