set -u
mkdir -p gpurun_out/ev
timeout -k 10 400 python bench.py > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err || { echo bench failed; tail -5 gpurun_out/ev/bench_default.err; exit 1; }
for cfg in "512 fp32" "4096 fp32" "512 bf16" "4096 bf16"; do
  set -- $cfg
  timeout -k 10 200 python tools/train_bench.py --rays $1 --precision $2 --steps 30 --warmup 5 >> gpurun_out/ev/train_bench.txt 2>> gpurun_out/ev/train_bench.err || { echo train_bench $cfg failed; exit 1; }
done
bash tools/pmc.sh gpurun_out/ev/pmc_bf16x3 --precision bf16x3 > gpurun_out/ev/pmc_bf16x3.txt 2>&1 || { echo pmc failed; tail -5 gpurun_out/ev/pmc_bf16x3.txt; exit 1; }
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/ev/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ev/gpu_tests.log
