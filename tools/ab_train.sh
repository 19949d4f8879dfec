#!/bin/bash
# A/B of library builds on the SAME device for the training iteration: runs bench.py --workload train alternately with
# each named variant (tools/variants/<name>/libnerf_hip.so; "new" = the product library) and prints ms per step.
# usage: bash tools/ab_train.sh <rounds> <variant> [<variant> ...] [-- bench.py args]
rounds=$1; shift
vs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
for i in $(seq $rounds); do for v in "${vs[@]}"; do
  if [ $v = new ]; then L=$PWD/nerf-loc_amd/lib/libnerf_hip.so; else L=$PWD/tools/variants/$v/libnerf_hip.so; fi
  echo -n "$v: "; NERF_HIP_LIB=$L timeout -k 10 150 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  %.0f rays/s  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" || exit 1
done; done
