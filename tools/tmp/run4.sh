mkdir -p gpurun_out/sgr gpurun_out/t4
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/sab sgr_attn_blocks.hip && timeout 300 /tmp/sab > ../../gpurun_out/sgr/ubench_attn_blocks_v4.txt)
for i in 1 2; do
timeout 300 python3 tools/train_bench.py --model CAMERA --steps 20 2>&1 | tail -1
timeout 300 python3 tools/train_bench.py --model SAEM --batch 64 --steps 20 2>&1 | tail -1
done
timeout 300 python3 tools/train_bench.py --model VSRN --steps 10 2>&1 | tail -1
