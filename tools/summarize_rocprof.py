"""Summarise a rocprofv3 --kernel-trace --stats output directory into a small text file for profiles/.
usage: python tools/summarize_rocprof.py <rocprof_out_dir> <out.md> [title]"""
import collections
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    trace = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(trace)))
    g = collections.defaultdict(list)
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]),
               int(r["Grid_Size_Z"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        g[key].append(dur)
    total = sum(sum(v) for v in g.values())
    with open(out, "w") as f:
        f.write("# %s\n\nsource: `rocprofv3 --kernel-trace --stats --output-format csv` (kernel_trace.csv), grouped by kernel and grid\n\n" % title)
        f.write("| kernel | workgroups (x,y,z) | VGPR/AGPR | LDS B | calls | avg us | min us | max us | total ms | % |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
            if sum(v) / total < 0.0005:
                continue
            f.write("| `%s` | %d,%d,%d | %s/%s | %s | %d | %.2f | %.2f | %.2f | %.3f | %.2f |\n" % (
                k[0][:110], k[1], k[2], k[3], k[4], k[5], k[6], len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3, 100 * sum(v) / total))
    print("wrote", out)


if __name__ == "__main__":
    main()
