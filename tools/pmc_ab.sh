#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace + two PMC passes of one bench workload under two environments (A/B of a kernel switch).
#   tools/pmc_ab.sh <tag> <workload> "<env A>" "<env B>"
set -u
TAG=$1; WL=$2; ENVA=${3:-}; ENVB=${4:-}
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
for V in A B; do
  if [ $V = A ]; then E="$ENVA"; else E="$ENVB"; fi
  OUT=gpurun_out/pmc_${TAG}_$V; mkdir -p $OUT
  ARGS="--workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-variants"
  for kv in $E; do export "$kv"; done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
  for kv in $E; do unset "${kv%%=*}"; done
  python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  echo "env: $E" >> $OUT/summary.txt
  tail -1 $OUT/bench_trace.log | cut -c1-300 >> $OUT/summary.txt
  find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
done
