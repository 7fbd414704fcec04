export TMPDIR=/tmp
mkdir -p gpurun_out/t9
timeout 600 python -m pytest tests/test_gemm_tn_gpu.py tests/test_gru_layers_gpu.py -x -q -m gpu 2>&1 | tail -2
for m in SCAN VSRN; do
timeout 300 python3 tools/train_bench.py --model $m --steps 20 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t9/$m -o t -- python3 tools/train_bench.py --model $m --steps 10 --warmup 3 > /dev/null 2>&1
python3 tools/trace_by_grid.py gpurun_out/t9/$m 13 60 | grep -i "skinny\|GPU time"
rm -rf gpurun_out/t9/$m
done
