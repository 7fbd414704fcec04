#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every bench workload once -> gpurun_out/bench_<tag>/<workload>.json, kernel stats for the pooled ones.
TAG=${1:-r04}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
python3 bench.py > $OUT/scan_t2i_coco5k.json 2> $OUT/scan_t2i_coco5k.err
for w in scan_i2t_coco5k vsepp_f30k1k vsrn_coco5k saem_coco5k camera_coco5k sgraf_saf_f30k1k sgraf_sgr_f30k1k sgraf_saf_coco5k sgraf_sgr_coco5k scan_t2i_f30k1k; do
  # a 10-50 ms step needs more than three of them: the first steps after the warm-up still run at the idle clock (VSE++ f30k: 11.1 ms
  # with --steps 3 --warmup 1, 10.0 ms with 20 / 5 on the same box)
  case $w in vsepp_f30k1k|scan_t2i_f30k1k) SW="--steps 20 --warmup 5";; *) SW="--steps 3 --warmup 1";; esac
  timeout 900 python3 bench.py --workload $w $SW --no-variants > $OUT/$w.json 2> $OUT/$w.err
done
python3 tools/make_synth_precomp.py /tmp/itr_synth --n-img 5000 > /dev/null && timeout 600 python3 bench.py --from-files /tmp/itr_synth --steps 3 --warmup 1 > $OUT/scan_t2i_coco5k_from_files.json 2> $OUT/from_files.err
for w in camera_coco5k saem_coco5k vsepp_f30k1k sgraf_saf_f30k1k sgraf_sgr_f30k1k vsrn_coco5k; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o t -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  find $OUT/prof_$w -name "*.db" -delete; find $OUT/prof_$w -name "*kernel_trace.csv" -delete
done
for f in $OUT/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    cb = d.get("cpu_baseline", {})
    print("%-32s %9.1f ms %8.2f Mpairs/s  frac %.3f  cpu %s" % (d["config"]["workload"], d["ms_per_step"], d["value"] / 1e6, d.get("roofline", {}).get("frac", float("nan")), cb.get("value")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
