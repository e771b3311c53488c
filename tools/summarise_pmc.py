"""Counter totals per kernel from rocprofv3 --pmc passes:  summarise_pmc.py DIR "pmc_*" OUT.txt "header line" [TRAFFIC.json]
TRAFFIC.json: HBM-side bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes - FETCH_SIZE (KiB) tallies
64 B per 128-B request on gfx950 (x 2), WRITE_SIZE (KiB) as reported."""
import collections
import csv
import glob
import json
import sys

out, pattern, dst, header = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
traffic_file = sys.argv[5] if len(sys.argv) > 5 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{out}/{pattern}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k][row["Counter_Name"]] += 1
with open(dst, "w") as g:
    g.write(header + "\n")
    for k, v in agg.items():
        if "bl_" not in k:
            continue
        g.write(k + "\n")
        for c, val in sorted(v.items()):
            g.write(f"    {c} {val:.6e} (launches {launches[k][c]}; per launch {val / launches[k][c]:.6e})\n")
        if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v and "SQ_WAVE_CYCLES" in v:
            pass
print(open(dst).read())
if traffic_file:
    traffic = {}
    for k, v in agg.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            n = launches[k]["FETCH_SIZE"]
            traffic[k] = {"launches": n, "fetch_bytes_per_launch_x2_corrected": v["FETCH_SIZE"] * 1024 * 2 / n,
                          "write_bytes_per_launch": v["WRITE_SIZE"] * 1024 / launches[k]["WRITE_SIZE"]}
    json.dump(traffic, open(traffic_file, "w"), indent=1)
    print(json.dumps(traffic, indent=1))
