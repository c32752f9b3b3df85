#!/bin/bash
# dev tool (GPU box): FETCH_SIZE / WRITE_SIZE per byte on 8-byte-per-lane coalesced accesses -> gpurun_out/hbm_calib/summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/hbm_calib
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/hbm_calib" tools/hbm_calib.hip || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/f" -o f -- "$OUT/hbm_calib" > "$OUT/f.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/w" -o w -- "$OUT/hbm_calib" > "$OUT/w.log" 2>&1
python - <<'PY' | tee gpurun_out/hbm_calib/summary.txt
import csv, glob, collections
n_bytes = (64 << 20) * 8
for tag, counter in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/hbm_calib/{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "calib" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = sum(v) / len(v)
        print(f"{counter:10s} {k:28s} {m:14.1f} per launch of {n_bytes} bytes -> {n_bytes / (m * 1024) if m else float('nan'):.4f} bytes per counted KiB-byte")
PY
