timeout 900 python -m pytest tests/test_gemm_tn_gpu.py tests/test_gru_layers_gpu.py tests/test_train_gpu.py tests/test_vsrn_train_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/train_all.sh gpurun_out/t8 noprof
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t8/p -o t -- python3 tools/train_bench.py --model SCAN > /dev/null 2>&1
grep -h "skinny" gpurun_out/t8/p/*kernel_stats.csv | cut -c1-60,200-330; rm -rf gpurun_out/t8/p
