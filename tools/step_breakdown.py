"""Per-kernel breakdown of ONE decode step (the last complete one) from a rocprofv3 --kernel-trace csv directory.
usage: python tools/step_breakdown.py <rocprof_out_dir> [...]"""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "embedding_gather" in r["Kernel_Name"]]
    seg = rows[starts[-2]:starts[-1]]
    g = collections.OrderedDict()
    for r in seg:
        k = (r["Kernel_Name"].replace("void mila::", "").replace("mila::", "")[:52],
             int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("==", d, "step total %.1f us" % sum(sum(v) for v in g.values()))
    for k, v in g.items():
        print("  %-54s %5d,%s,%s n=%3d avg %7.2f tot %8.1f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), sum(v)))
