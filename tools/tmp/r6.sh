export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_oracle_golden.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python3 tools/train_bench.py --model SCAN --steps 20 2>&1 | tail -1; done
mkdir -p gpurun_out/t20
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t20/p -o t -- python3 tools/train_bench.py --model SCAN --steps 10 --warmup 3 > /dev/null 2>&1
grep -h "scan_train" gpurun_out/t20/p/*kernel_stats.csv | awk -F'","' '{print substr($1,1,70), $2, $4}'; rm -rf gpurun_out/t20/p
