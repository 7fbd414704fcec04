timeout 1500 python -m pytest tests/test_gemm_tn_gpu.py tests/test_kernels_gpu.py tests/test_gemm_stream_gpu.py tests/test_train_gpu.py tests/test_camera_train_gpu.py tests/test_train_camera_gpu.py tests/test_saem_train_gpu.py tests/test_train_bert_gpu.py tests/test_vsrn_train_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/train_all.sh gpurun_out/t18 noprof
