"""Gaps on the GPU timeline between consecutive decode steps, from a rocprofv3 kernel trace: the idle time after each step's last kernel (advance_position*).
usage: python tools/trace_gaps.py <rocprof_out_dir>"""
import csv
import glob
import os
import sys

import numpy as np

trace = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
gaps, steps = [], []
last_adv_end = None
for i, r in enumerate(rows):
    if last_adv_end is not None:
        gaps.append((int(r["Start_Timestamp"]) - last_adv_end[0]) / 1e3)
        steps.append((int(r["Start_Timestamp"]) - last_adv_end[1]) / 1e3)
        last_adv_end = None
    if "advance_position" in r["Kernel_Name"]:
        last_adv_end = (int(r["End_Timestamp"]), int(r["Start_Timestamp"]))
g = np.array(gaps)
print("steps: %d; idle after a step's last kernel until the next kernel starts: median %.1f us, mean %.1f us, p90 %.1f us, max %.1f us" % (len(g), np.median(g), g.mean(), np.percentile(g, 90), g.max()))
ends = [int(r["End_Timestamp"]) for r in rows if "advance_position" in r["Kernel_Name"]]
d = np.diff(ends) / 1e3
print("step period (end of step to end of next): median %.1f us, mean %.1f us" % (np.median(d), d.mean()))
