#!/bin/bash
# Re-creates the files profiles/<tag>_* on a GPU box (run from the repository root; <tag> is the file-name prefix),
# in two parts so that each fits one 20-minute box call:
#   bash tools/reproduce.sh r02 a      tests + bench lines
#   bash tools/reproduce.sh r02 b      rocprofv3 traces, PMC passes, per-kernel bench
# Needs the library built beforehand (python -c "import __graft_entry__ as g; g.build()").  Every GPU step has its
# own time limit and the script stops at the first failure.
set -eu
tag=${1:-r02}
part=${2:-a}
out=gpurun_out/$tag
mkdir -p "$out" profiles
export TMPDIR=/tmp
if [ "$part" = a ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -s > "$out/gpu_tests.log" 2>&1
  tail -3 "$out/gpu_tests.log"
  timeout -k 10 300 python bench.py > "$out/bench_bf16.json"
  timeout -k 10 200 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/bench_fp32.json"
  for w in rays4m objects train; do
    timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$out/bench_$w.json"
  done
  timeout -k 10 200 python bench.py --workload train --train-precision bf16 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > "$out/bench_train_bf16.json"
  timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_sustained_200.json"
  cp "$out/bench_sustained_200.json" profiles/${tag}_bench_sustained_200.json
  NERF_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 8 --steps 2 --warmup 1 > "$out/eight_rank_rehearsal.txt"
  tail -1 "$out/eight_rank_rehearsal.txt" > profiles/${tag}_eight_rank_rehearsal.txt
  timeout -k 10 250 python3 tools/c_abi/bench_graph.py > profiles/${tag}_c_abi_graph.txt
  for p in bf16 fp32; do cp "$out/bench_$p.json" profiles/${tag}_bench_$p.json; done
  for w in rays4m objects train train_bf16; do cp "$out/bench_$w.json" profiles/${tag}_bench_$w.json; done
  cp "$out/gpu_tests.log" profiles/${tag}_gpu_tests.log
else
  bash tools/profile_bench.sh "$out/prof_bf16" bf16 > "$out/prof_bf16.log" 2>&1
  bash tools/profile_bench.sh "$out/prof_fp32" fp32 > "$out/prof_fp32.log" 2>&1
  timeout -k 10 200 python tools/kernels_bench.py > "$out/kernels_bench.json"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train_trace" -- python3 tools/train_bench.py --steps 5 --warmup 2 > "$out/train_trace.log" 2>&1
  python3 tools/kstats.py "$out/train_trace" 12 > "$out/train_kernel_stats.txt"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train16_trace" -- python3 tools/train_bench.py --precision bf16 --steps 5 --warmup 2 > "$out/train16_trace.log" 2>&1
  python3 tools/kstats.py "$out/train16_trace" 16 > "$out/train16_kernel_stats.txt"
  bash tools/pmc.sh "$out/pmc_bf16" > "$out/pmc_bf16.log" 2>&1
  bash tools/pmc_train.sh "$out/pmc_train" > "$out/pmc_train.log" 2>&1
  for p in bf16 fp32; do
    cp "$out/prof_$p/kernel_stats.txt" profiles/${tag}_kernel_stats_$p.txt
    cp "$out/prof_$p/traffic.txt" profiles/${tag}_pmc_traffic_$p.txt
    cp "$out/prof_$p/traffic.json" profiles/${tag}_traffic_$p.json
  done
  cp "$out/kernels_bench.json" profiles/${tag}_kernels_bench.json
  cp "$out/train_kernel_stats.txt" profiles/${tag}_kernel_stats_train.txt
  cp "$out/train16_kernel_stats.txt" profiles/${tag}_kernel_stats_train_bf16.txt
  cp "$out/pmc_bf16/summary.txt" profiles/${tag}_pmc_mlp_bf16_raw.txt
  cp "$out/pmc_train/summary.txt" profiles/${tag}_pmc_train_raw.txt
fi
echo "profiles/${tag}_* part $part written"
