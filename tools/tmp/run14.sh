timeout 1500 python -m pytest tests/test_gemm_tn_gpu.py tests/test_train_gpu.py tests/test_sgraf_train_gpu.py tests/test_sgraf_batched_gpu.py tests/test_camera_train_gpu.py tests/test_train_camera_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/train_all.sh gpurun_out/t14 noprof
