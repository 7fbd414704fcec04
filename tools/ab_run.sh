#!/bin/bash
# Run ON THE GPU BOX: same-box, interleaved runs of one bench workload for several trees:  "." = this tree, any other name = tools/ab/<name>
# (tools/ab_build.sh).   tools/ab_run.sh <workload> <out dir> <rounds> <tree> [<tree> ...]      env for a tree: name:VAR=value
WL=$1; OUT=$2; N=$3; shift 3
mkdir -p $OUT
for i in $(seq 1 $N); do
  for T in "$@"; do
    NAME=${T%%:*}; ENVS=""; [ "$T" != "$NAME" ] && ENVS=${T#*:}
    if [ "$NAME" = "." ]; then B=bench.py; TAG=main; else B=tools/ab/$NAME/bench.py; TAG=$NAME; fi
    [ -n "$ENVS" ] && TAG="${TAG}_$(echo $ENVS | tr '=' '_')"
    EXTRA=""; grep -q "no-other-configs" $B && EXTRA="--no-other-configs"
    env $ENVS python3 $B --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-variants $EXTRA > $OUT/${TAG}_$i.json 2>>$OUT/err.log
  done
done
for f in $OUT/*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"kernel_ms": [0-9.]*' $f)"; done
