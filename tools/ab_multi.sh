#!/bin/bash
# Run ON THE GPU BOX: same-box, interleaved timing of ONE bench workload over several side-by-side builds (tools/ab/<name>; "." = this tree).
#   tools/ab_multi.sh <workload> <out dir> <rounds> <steps> <name> [<name> ...]
WL=$1; OUT=$2; N=$3; STEPS=$4; shift 4
mkdir -p $OUT
ARGS="--workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-variants --no-other-configs"
for i in $(seq 1 $N); do
  for NAME in "$@"; do
    if [ "$NAME" = "." ]; then B=bench.py; TAG=this; else B=tools/ab/$NAME/bench.py; TAG=$NAME; fi
    python3 $B $ARGS > $OUT/${TAG}_${WL}_$i.json 2>>$OUT/err.log
  done
done
for f in $OUT/*_${WL}_*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"kernel_ms": [0-9.]*' $f) $(grep -o '"frac": [0-9.]*' $f | head -1)"; done
