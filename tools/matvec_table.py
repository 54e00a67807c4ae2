"""Print the per-shape best (R, U, max_blocks) of a tools/bench_matvec.py sweep.  usage: matvec_table.py <sweep.jsonl> [<older.jsonl>]"""
import collections
import json
import sys


def load(f):
    g = collections.defaultdict(list)
    for l in open(f):
        if not l.startswith("{"):
            continue
        r = json.loads(l)
        if r["kernel"] == "matvec":
            g[(r["fmt"], r["shape"], r["K"], r["N"])].append((r["us"], r["R"], r["U"], r["max_blocks"]))
    return g


g = load(sys.argv[1])
old = load(sys.argv[2]) if len(sys.argv) > 2 else {}
for k, v in g.items():
    d = [x for x in v if x[1] == 0][0]
    o = [x for x in old.get(k, []) if x[1] == 0]
    print(k, ("old default %.2f ->" % o[0][0]) if o else "", "default %.2f" % d[0], " best:",
          " ".join("%.2f(R%dU%db%d)" % x for x in sorted(v)[:4]))
