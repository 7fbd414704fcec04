#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel statistics of tools/train_bench.py for each model family.
# Usage: tools/profile_train.sh  ->  gpurun_out/prof_train/rocprofv3_kernel_stats_train_<model>.csv
set -u
OUT=gpurun_out/prof_train
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag, train_bench arguments
    local tag=$1; shift
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python3 tools/train_bench.py "$@" > $OUT/$tag.log 2>&1
    local f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/rocprofv3_kernel_stats_train_$tag.csv
    tail -1 $OUT/$tag.log
    rm -rf $OUT/$tag
}
run SCAN --model SCAN
run SAEM_batch_64 --model SAEM --batch 64
run CAMERA --model CAMERA
run SGRAF_module_SAF --model SGRAF --module SAF
run SGRAF_module_SGR --model SGRAF --module SGR
run VSRN --model VSRN
