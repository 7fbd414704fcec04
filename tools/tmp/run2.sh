mkdir -p gpurun_out/sgr gpurun_out/saem
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/sab sgr_attn_blocks.hip && timeout 300 /tmp/sab | tee ../../gpurun_out/sgr/ubench_attn_blocks_v2.txt)
export TMPDIR=/tmp
for m in "SAEM --batch 64" "VSRN"; do
  tag=$(echo $m | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/saem/$tag -o t -- python3 tools/train_bench.py --model $m --steps 10 --warmup 3 > gpurun_out/saem/$tag.log 2>&1
  python3 tools/trace_by_grid.py gpurun_out/saem/$tag 13 40 > gpurun_out/saem/${tag}_by_grid.txt
  rm -rf gpurun_out/saem/$tag
done
cat gpurun_out/saem/SAEM_by_grid.txt
