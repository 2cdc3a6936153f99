"""Diagnostic run of every staged reference test file (tests/refpy.py), without -x: one log per file under
gpurun_out/refpy/ and a summary line per file.  For the GPU box: python tests/scripts/refpy_run_all.py [substr ...]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refpy  # noqa: E402

out_dir = os.path.join(refpy.ROOT, "gpurun_out", "refpy")
os.makedirs(out_dir, exist_ok=True)
wanted = [a for a in sys.argv[1:] if not a.startswith('--')]
summary = []
for rel in refpy.test_files():
    if wanted and not any(w in rel for w in wanted):
        continue
    with tempfile.TemporaryDirectory() as cwd:
        t0 = time.time()
        try:
            run = refpy.KNOWN_BROKEN_IN_REFERENCE.get(rel, ((), None))[0]
            cmd_out = refpy.run_file(rel, cwd, timeout=300, tests=() if "--all" in sys.argv or run is None else run)
            rc, text = cmd_out.returncode, cmd_out.stdout
        except Exception as e:  # timeout
            rc, text = -1, "EXCEPTION %r" % (e,)
        dt = time.time() - t0
    tail = [l for l in text.splitlines() if l.strip()][-1:] or [""]
    summary.append("%-60s rc=%d %5.1fs  %s" % (rel, rc, dt, tail[0]))
    with open(os.path.join(out_dir, rel.replace("/", "__") + ".log"), "w") as f:
        f.write(text)
    print(summary[-1], flush=True)
with open(os.path.join(out_dir, "SUMMARY.txt"), "w") as f:
    f.write("\n".join(summary) + "\n")
