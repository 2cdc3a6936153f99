"""Per-step kernel time of scripts/edge_cut_p8_probe.py ... solo | sym from a rocprofv3 kernel trace: steps are delimited
by rank 0's resolve launches (one per step with MERGED=1); everything between two of them is one step's work.
  solo: rank 0's own kernels and the eight owners' service for it (every owner its own launches).
  sym (--thread TID = rank 0's host thread, printed by the probe): only the kernels THAT thread enqueued -- one rank's
       step in the symmetric case: its own request + ONE sampling launch per hop and ONE row gather for its seven peers.
usage: p8_solo_step.py <kernel_trace.csv> [--thread TID]"""
import collections
import csv
import sys

tid = None
if "--thread" in sys.argv:
    tid = sys.argv[sys.argv.index("--thread") + 1]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if tid is not None:
    rows = [r for r in rows if r["Thread_Id"] == tid]
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
engine = sum(v for n, v in acc.items() if "copyBuffer" not in n)
print("steps %d%s: wall %.3f ms/step, GPU busy %.3f ms/step, kernel sum %.3f ms/step, engine kernels (transport copies "
      "excluded) %.3f ms/step in %.1f launches"
      % (steps, "" if tid is None else " (host thread %s only)" % tid, (hi - lo) / steps / 1e6, busy / steps / 1e6,
         sum(acc.values()) / steps / 1e6, engine / steps / 1e6,
         sum(c for n, c in cnt.items() if "copyBuffer" not in n) / steps))
for n, v in acc.most_common(26):
    print("%-72s %6.1f launches/step %8.3f ms/step %8.1f us/launch" % (n, cnt[n] / steps, v / steps / 1e6, v / cnt[n] / 1e3))
