"""Per-step kernel time of scripts/edge_cut_p8_probe.py ... solo from a rocprofv3 kernel trace: steps are delimited by
rank 0's resolve launches (one per step with MERGED=1); everything between two of them is one step's work -- rank 0's own
kernels and the owners' service for it.  usage: p8_solo_step.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "glx_dist_resolve_kernel" in r["Kernel_Name"]]
assert len(marks) >= 4, len(marks)
lo, hi, steps = marks[2], marks[-1], len(marks) - 3
acc, cnt = collections.Counter(), collections.Counter()
busy = 0
last_end = lo
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s < hi:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        acc[name] += e - s
        cnt[name] += 1
        if e > last_end:  # union of busy intervals (kernels of different ranks overlap)
            busy += e - max(s, last_end)
            last_end = e
print("steps %d: wall %.3f ms/step, GPU busy %.3f ms/step, kernel sum %.3f ms/step"
      % (steps, (hi - lo) / steps / 1e6, busy / steps / 1e6, sum(acc.values()) / steps / 1e6))
for n, v in acc.most_common(22):
    print("%-72s %6.1f launches/step %8.3f ms/step" % (n, cnt[n] / steps, v / steps / 1e6))
