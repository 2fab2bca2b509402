"""Condenses rocprofv3 output of tools/profile.sh into the small files committed under profiles/rNN:
kernel_stats.csv (copy of the --stats table) and pmc_summary.csv (per kernel, per counter: dispatches, mean)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def main(out: str) -> None:
    stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(out, "kernel_stats.csv"), "w") as dst:
            dst.write(fh.read())
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                cell = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                cell[0] += 1
                cell[1] += float(row["Counter_Value"])
    with open(os.path.join(out, "pmc_summary.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for k in sorted(acc):
            for c in sorted(acc[k]):
                n, s = acc[k][c]
                w.writerow([k, c, n, f"{s / n:.4f}"])
    print(json.dumps({"kernels": len(acc)}))


if __name__ == "__main__":
    main(sys.argv[1])
