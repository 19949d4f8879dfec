#!/bin/bash
# A/B of library builds on the SAME device (devices of the pool differ by several per cent): runs tools/mlp_bench.py
# alternately with each named variant (tools/variants/<name>/libnerf_hip.so; "new" = the product library).
# usage: bash tools/ab.sh <rounds> <variant> [<variant> ...] [-- mlp_bench args]
rounds=$1; shift
vs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
for i in $(seq $rounds); do for v in "${vs[@]}"; do
  if [ $v = new ]; then L=$PWD/nerf-loc_amd/lib/libnerf_hip.so; else L=$PWD/tools/variants/$v/libnerf_hip.so; fi
  echo -n "$v: "; NERF_HIP_LIB=$L timeout -k 10 100 python tools/mlp_bench.py --reps 7 "$@" 2>&1 | grep -E "^mlp|^stamps" | cut -c1-250 || exit 1
done; done
