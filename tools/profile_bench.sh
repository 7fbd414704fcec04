#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + PMC passes for bench.py.  Usage: tools/profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
# PMC passes (separate runs, counters only -- no trace domains combined with --pmc)
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc4.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-400 >> $OUT/summary.txt
cat $OUT/summary.txt
# keep the merged output small: drop the raw per-dispatch csv/db, keep stats + summary
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
