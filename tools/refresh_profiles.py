#!/usr/bin/env python3
"""Copy the judged artefacts of one tools/profile_bench.sh run (gpurun_out/prof_<tag>/) into profiles/<round>/.

    python tools/refresh_profiles.py gpurun_out/prof_r01c profiles/r01 [bench.json]

Writes: rocprofv3_bench_<workload>_summary.txt, rocprofv3_kernel_stats.csv, scan_pmc.json (the PMC sums bench.py
reads for roofline.traffic: HBM bytes = (FETCH_SIZE * 2 + WRITE_SIZE) KiB, MI355X_MICROARCH.md gfx950 correction),
bench_n1.json."""
import json
import os
import re
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
summary = open(os.path.join(src, "summary.txt")).read()
m = re.search(r'"workload": "([a-z0-9_]+)"', summary)
workload = m.group(1) if m else "scan_t2i_coco5k"
name = "rocprofv3_bench_%s_summary.txt" % workload
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, name))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "rocprofv3_kernel_stats.csv"))


def counter(kernel, cname):
    """value 'per dispatch' of `cname` in the block of `kernel`"""
    blocks = re.split(r"\n   (?=\S)", summary)
    for b in blocks:
        if b.lstrip().startswith(kernel) or b.lstrip().startswith("void " + kernel):
            mm = re.search(r"%s\s+per dispatch ([0-9.e+]+)" % re.escape(cname), b)
            if mm:
                return float(mm.group(1))
    return None


kern = "itr::scan_xattn_kernel<0"          # <0, 0>: exact fp32, t2i (before round 3: <0>)
f, w = counter(kern, "FETCH_SIZE"), counter(kern, "WRITE_SIZE")
out = {"kernel": kern, "workload": workload, "n_gpus": 1,
       "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
       "hbm_bytes_per_launch": (f * 2 + w) * 1024 if f is not None and w is not None else None,
       "TCC_HIT_sum": counter(kern, "TCC_HIT_sum"), "TCC_MISS_sum": counter(kern, "TCC_MISS_sum"),
       "SQ_INSTS_MFMA": counter(kern, "SQ_INSTS_MFMA"), "SQ_VALU_MFMA_BUSY_CYCLES": counter(kern, "SQ_VALU_MFMA_BUSY_CYCLES"),
       "SQ_BUSY_CYCLES": counter(kern, "SQ_BUSY_CYCLES"), "GRBM_GUI_ACTIVE": counter(kern, "GRBM_GUI_ACTIVE"),
       "source": "%s/%s (separate --pmc passes)" % (dst.rstrip("/"), name),
       # the commit the profiled tree was built from (bench.py prints it next to the replayed traffic figure)
       "commit": __import__("subprocess").run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"}
json.dump(out, open(os.path.join(dst, "scan_pmc.json"), "w"), indent=1)
if len(sys.argv) > 3:
    shutil.copy(sys.argv[3], os.path.join(dst, "bench_n1.json"))
print(json.dumps(out, indent=1))
