for rep in 1 2; do for lib in product alt; do
  if [ $lib = alt ]; then export HT_LIB_PATH=$PWD/hand_tracking_samples_amd/libht_alt.so; else unset HT_LIB_PATH; fi
  for args in "" "--frames-per-gpu 8192 --steps 5 --warmup 2" "--workload config5-e2e --steps 10"; do
    timeout -k 10 120 python3 bench.py --no-cpu-baseline $args 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$args', j['value'], j['ms_per_step'], j['verified'], j['phase_ms_per_step']['contacts'])"
  done; done; done
