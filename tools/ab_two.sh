#!/bin/bash
# Run ON THE GPU BOX: same-box, interleaved A/B of one bench workload: a side-by-side build under tools/ab/<name> (tools/ab_checkout.sh
# <commit> <name> or tools/ab_build.sh <name> <flags>) against this tree.   tools/ab_two.sh <name> <workload> <out dir> [rounds] [steps]
NAME=${1:-base}; WL=${2:-sgraf_sgr_f30k1k}; OUT=${3:-gpurun_out/ab}; N=${4:-2}; STEPS=${5:-5}
mkdir -p $OUT
ARGS="--workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-variants --no-other-configs"
for i in $(seq 1 $N); do
  python3 tools/ab/$NAME/bench.py $ARGS > $OUT/${NAME}_${WL}_$i.json 2>$OUT/err.log
  python3 bench.py $ARGS > $OUT/new_${WL}_$i.json 2>>$OUT/err.log
done
for f in $OUT/*_${WL}_*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"kernel_ms": [0-9.]*' $f) $(grep -o '"frac": [0-9.]*' $f | head -1)"; done
