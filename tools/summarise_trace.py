"""Per-kernel launch statistics from a rocprofv3 --kernel-trace directory:  summarise_trace.py DIR OUT.txt "header line" """
import collections
import csv
import glob
import sys

src, dst, header = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
regs = {}
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    regs[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"),
               row.get("Grid_Size"), row.get("Workgroup_Size"))
with open(dst, "w") as g:
    g.write(header + "\n")
    g.write("kernel, launches, avg_ms, min_ms, max_ms, total_ms, (VGPR, AGPR, SGPR, LDS, scratch, grid, wg)\n")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        g.write(f"{k}, {len(v)}, {sum(v) / len(v):.3f}, {min(v):.3f}, {max(v):.3f}, {sum(v):.2f}, {regs[k]}\n")
print(open(dst).read())
