timeout 900 python -m pytest tests/test_sgraf_batched_gpu.py tests/test_sgraf_train_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 300 python3 tools/train_bench.py --model SGRAF --module SGR --steps 20 2>&1 | tail -1; done
