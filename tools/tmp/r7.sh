timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python3 tools/train_bench.py --model SCAN --steps 20 2>&1 | tail -1; done
