#!/bin/bash
# Run ON THE GPU BOX: same-box, interleaved A/B of one bench workload: the round-3 tree (tools/ab/r3, built beforehand) against this tree
# under two environments (build the old tree first: tools/ab_checkout.sh 0807f13 r3).   tools/ab_sgr.sh <workload> <out dir> [rounds]
WL=${1:-sgraf_sgr_f30k1k}; OUT=${2:-gpurun_out/ab}; N=${3:-2}
mkdir -p $OUT
ARGS="--workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-other-configs"
for i in $(seq 1 $N); do
  python3 tools/ab/r3/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $OUT/r3_$i.json 2>$OUT/err.log
  ITR_SGR_GROUP_ROWS=32 python3 bench.py $ARGS > $OUT/new32_$i.json 2>>$OUT/err.log
  python3 bench.py $ARGS > $OUT/new64_$i.json 2>>$OUT/err.log
done
for f in $OUT/*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"kernel_ms": [0-9.]*' $f)"; done
