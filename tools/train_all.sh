#!/bin/bash
# Run ON THE GPU BOX (via gpurun): time model.train_emb for every model family (tools/train_bench.py), then collect
# rocprofv3 kernel statistics of the same commands.  Usage: tools/train_all.sh <outdir> [noprof]
set -u
OUT=${1:-gpurun_out/train_all}
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/times.txt
cfgs=("SCAN|--model SCAN" "VSE_PP|--model VSE_PP" "SAEM_batch_64|--model SAEM --batch 64" "CAMERA|--model CAMERA" "SGRAF_module_SAF|--model SGRAF --module SAF" "SGRAF_module_SGR|--model SGRAF --module SGR" "VSRN|--model VSRN")
for c in "${cfgs[@]}"; do
    tag=${c%%|*}; args=${c#*|}
    timeout 400 python3 tools/train_bench.py $args --steps 20 > $OUT/$tag.time.log 2>&1
    tail -1 $OUT/$tag.time.log | tee -a $OUT/times.txt
done
[ "${2:-}" = "noprof" ] && exit 0
for c in "${cfgs[@]}"; do
    tag=${c%%|*}; args=${c#*|}
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python3 tools/train_bench.py $args > $OUT/$tag.prof.log 2>&1
    f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/rocprofv3_kernel_stats_train_$tag.csv
    rm -rf $OUT/$tag
done
