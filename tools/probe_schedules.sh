for S in "4,4,8,16" "8,8,16" "1,1,2,4,8,16,32" "2,2,4,8,16,32" "4,4,8,16,32" "8,8,16,32" "4,4,8,8,8,16"; do
  echo "== $S"
  ILQR_LS_SCHEDULE=$S timeout -k 10 200 python tools/probe_big.py cartpole 131072 2>&1 | grep -E "solves/s" | tail -1 || exit 1
  ILQR_LS_SCHEDULE=$S timeout -k 10 200 python tools/probe_big.py pendulum 131072 2>&1 | grep -E "solves/s" | tail -1 || exit 1
  ILQR_LS_SCHEDULE=$S timeout -k 10 200 python tools/probe_big.py acrobot 65536 2>&1 | grep -E "solves/s" | tail -1 || exit 1
done
