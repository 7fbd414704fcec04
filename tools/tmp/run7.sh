timeout 900 python -m pytest tests/test_vsrn_train_gpu.py tests/test_vsrn_gpu.py tests/test_oracle_golden.py -x -q 2>&1 | tail -4
for i in 1 2; do timeout 300 python3 tools/train_bench.py --model VSRN --steps 10 2>&1 | tail -1; done
