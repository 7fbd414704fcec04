mkdir -p gpurun_out/sgr gpurun_out/t3
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/sab sgr_attn_blocks.hip && timeout 300 /tmp/sab > ../../gpurun_out/sgr/ubench_attn_blocks_v3.txt)
timeout 300 python -m pytest tests/test_gemm_tn_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/train_all.sh gpurun_out/t3 noprof
