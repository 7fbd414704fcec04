export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sgraf_batched_gpu.py tests/test_sgraf_train_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 300 python3 tools/train_bench.py --model SGRAF --module SGR --steps 20 2>&1 | tail -1
timeout 300 python3 tools/train_bench.py --model SGRAF --module SAF --steps 20 2>&1 | tail -1
done
mkdir -p gpurun_out/t16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t16/p -o t -- python3 tools/train_bench.py --model SGRAF --module SAF --steps 10 --warmup 3 > /dev/null 2>&1
grep -h "ctx_bwd\|ctx_fwd" gpurun_out/t16/p/*kernel_stats.csv | cut -d, -f2-4 ; rm -rf gpurun_out/t16/p
