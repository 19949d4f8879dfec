#!/bin/bash
# Re-creates every file under profiles/ for one round on a GPU box (run from the repository root):
#   bash tools/reproduce.sh r01
# Needs the library built beforehand (python -c "import __graft_entry__ as g; g.build()").
set -eu
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out" profiles
export TMPDIR=/tmp
python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -s > "$out/gpu_tests.log" 2>&1
python bench.py > "$out/bench_bf16.json"
python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_fp32.json"
for w in rays4m objects train; do
  python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > "$out/bench_$w.json"
done
bash tools/profile_bench.sh "$out/prof_bf16" bf16 > "$out/prof_bf16.log" 2>&1
bash tools/profile_bench.sh "$out/prof_fp32" fp32 > "$out/prof_fp32.log" 2>&1
python tools/kernels_bench.py > "$out/kernels_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train_trace" -- python3 tools/train_bench.py --steps 5 --warmup 2 > "$out/train_trace.log" 2>&1
python3 tools/kstats.py "$out/train_trace" 12 > "$out/train_kernel_stats.txt"
python bench.py --workload train --train-precision bf16 --no-cpu-baseline --steps 10 --warmup 3 > "$out/bench_train_bf16.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train16_trace" -- python3 tools/train_bench.py --precision bf16 --steps 5 --warmup 2 > "$out/train16_trace.log" 2>&1
python3 tools/kstats.py "$out/train16_trace" 16 > "$out/train16_kernel_stats.txt"
bash tools/pmc.sh "$out/pmc_bf16" > "$out/pmc_bf16.log" 2>&1
for p in bf16 fp32; do
  cp "$out/prof_$p/kernel_stats.txt" profiles/${tag}_kernel_stats_$p.txt
  cp "$out/prof_$p/traffic.txt" profiles/${tag}_pmc_traffic_$p.txt
  cp "$out/prof_$p/traffic.json" profiles/${tag}_traffic_$p.json
  cp "$out/bench_$p.json" profiles/${tag}_bench_$p.json
done
for w in rays4m objects train; do cp "$out/bench_$w.json" profiles/${tag}_bench_$w.json; done
cp "$out/gpu_tests.log" profiles/${tag}_gpu_tests.log
cp "$out/kernels_bench.json" profiles/${tag}_kernels_bench.json
cp "$out/train_kernel_stats.txt" profiles/${tag}_kernel_stats_train.txt
cp "$out/train16_kernel_stats.txt" profiles/${tag}_kernel_stats_train_bf16.txt
cp "$out/bench_train_bf16.json" profiles/${tag}_bench_train_bf16.json
cp "$out/pmc_bf16/summary.txt" profiles/${tag}_pmc_mlp_bf16_raw.txt
echo "profiles/${tag}_* written"
