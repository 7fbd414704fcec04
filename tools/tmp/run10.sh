timeout 900 python -m pytest tests/test_gemm_tn_gpu.py tests/test_gru_layers_gpu.py tests/test_vsrn_train_gpu.py tests/test_train_gpu.py tests/test_saem_train_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/train_all.sh gpurun_out/t10 noprof
