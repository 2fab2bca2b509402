#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/probes/fetch_calib_probe.hip); run on the GPU box:
#   bash tools/diag/fetch_calibration.sh > gpurun_out/fetch_calibration.txt
set -u
REPO=$(pwd)
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $REPO/tools/probes/fetch_calib_probe.hip 2>/dev/null || exit 1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/fc_$c -o fc -- /tmp/fetch_calib > /tmp/fc_$c.log 2>&1
  f=$(find /tmp/fc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
bytes_ = 40_000_000 * 12
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0]] = acc.get(r["Kernel_Name"].split("(")[0], 0.0) + float(r["Counter_Value"])
for k, v in acc.items():
    print(f"{sys.argv[2]:10s} {k:24s} counter {v:14.1f} KB   known bytes / (counter x 1024) = {bytes_ / (v * 1024) if v else float('nan'):.3f}")
PY
done
