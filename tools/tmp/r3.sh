for i in 1 2; do
timeout 300 python3 tools/train_bench.py --model CAMERA --steps 20 2>&1 | tail -1
done
timeout 300 python3 tools/train_bench.py --model SAEM --batch 64 --steps 20 2>&1 | tail -1
timeout 300 python3 tools/train_bench.py --model VSRN --steps 20 2>&1 | tail -1
timeout 300 python3 tools/train_bench.py --model SGRAF --module SGR --steps 20 2>&1 | tail -1
