#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : one short bench line per value of an env knob
VAR=$1; shift
for v in "$@"; do
  echo -n "$VAR=$v  "
  env $VAR=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], 'ms', j['roofline']['achieved'], 'GB/s', j['roofline']['frac'], 'parity', j['parity']['rel_l2_vs_cpu_oracle'])"
done
