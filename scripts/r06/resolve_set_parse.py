"""usage: resolve_set_parse.py <kernel_trace.csv>  -- per phase of scripts/r06/resolve_set_probe.py: average duration (us) of
rank 0's fill / resolve / assign / finalize launches (the first call of a phase, which may regrow the set, is dropped)."""
import csv, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
REPS = 6
PHASES = ["share=16,peek=1", "share=64,peek=1", "share=256,peek=1", "share=16,peek=0", "share=64,peek=0", "share=256,peek=0",
          "own_first=0", "own_first=1", "own_first=0", "own_first=1", "blocks=2048", "blocks=4096", "blocks=8192",
          "blocks=768,ids=4", "blocks=1024,ids=4", "blocks=1280,ids=4", "blocks=1536,ids=4", "blocks=1792,ids=4", "blocks=1024,ids=8",
          "blocks=1536,ids=8", "blocks=1280", "blocks=1536", "blocks=1024"]
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
want = ("glx_dist_fill_keys_kernel", "glx_dist_resolve_kernel", "glx_dist_assign_kernel", "glx_dist_finalize_list_kernel", "glx_lookup_kernel")
seq = collections.defaultdict(list)
for r in rows:
    for w in want:
        if w in r["Kernel_Name"]:
            seq[w].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n_res = len(seq["glx_dist_resolve_kernel"])
print("# resolve launches in the trace: %d (expected %d + regrows)" % (n_res, REPS * len(PHASES)))
print("%-28s %10s %10s %10s %10s %10s" % ("phase", "fill", "resolve", "assign", "finalize", "sum"))
for i, name in enumerate(PHASES):
    vals = []
    for w in want[:4]:
        s = seq[w]
        per = len(s) // len(PHASES) if len(s) >= len(PHASES) else 0
        part = s[i * per + 1:(i + 1) * per] if per > 1 else []
        vals.append(sum(part) / len(part) if part else float("nan"))
    print("%-28s %10.1f %10.1f %10.1f %10.1f %10.1f" % (name, *vals, sum(vals)))
