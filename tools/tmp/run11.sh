export TMPDIR=/tmp
mkdir -p gpurun_out/t11
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t11/V -o t -- python3 tools/train_bench.py --model VSRN --steps 10 --warmup 3 > /dev/null 2>&1
python3 tools/trace_by_grid.py gpurun_out/t11/V 13 45 > gpurun_out/t11/VSRN_by_grid.txt
rm -rf gpurun_out/t11/V
