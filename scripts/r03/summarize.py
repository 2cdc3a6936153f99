"""gpurun_out/r03_prof/ (written by scripts/r03/profile.sh on the GPU box) -> profiles/r03/: per-workload kernel stats,
per-dispatch rows of the glx kernels, PMC traffic of the dominant launches, SUMMARY.md and profiles/pmc_traffic.json.

    python scripts/r03/summarize.py
"""
import collections
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "r03_prof")
DST = os.path.join(ROOT, "profiles", "r03")
os.makedirs(DST, exist_ok=True)
PMC_STEPS = 6  # the PMC passes run --steps 5 --warmup 1; later hop-2-sized aggregate launches are the cache-free probe


def short(n):
    m = re.search(r"(glx_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n


def rows_of(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def find(prefix, suffix):
    for f in sorted(os.listdir(SRC)):
        if f.startswith(prefix) and f.endswith(suffix):
            return os.path.join(SRC, f)
    return None


out = ["# r03 rocprofv3 summary (one MI355X)", "",
       "Commands: scripts/r03/profile.sh.  Per workload w in {c3, c2, c5}: `rocprofv3 --kernel-trace --stats --output-format csv -- "
       "python bench.py --workload w --steps 20 --warmup 5 --roofline-probes off` (lean: no CPU baseline / host boundary / "
       "probes / other configs), then `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in its own run with `--kernel-trace` only "
       "(5 steps + 1 warm-up, roofline probes ON: the launches after the workload's are bench.py's cache-free leg).  "
       "FETCH_SIZE x2 and WRITE_SIZE x1 as calibrated in profiles/r02/SUMMARY.md (MI355X_MICROARCH.md, HBM section; the "
       "calibration on `glx_place_rows_kernel` is repeated below).  The driver's full command (`python bench.py --steps 20 "
       "--warmup 5`) ran first: `bench_c3_n1_final.json`.", ""]
pmc_json = {}
for wl in ("c3", "c2", "c5", "c4"):
    bench_path = os.path.join(SRC, "%s_bench_trace.json" % wl)
    if not os.path.exists(bench_path):
        continue
    try:
        b = json.loads([ln for ln in open(bench_path).read().splitlines() if ln.startswith("{")][-1])
    except Exception:  # noqa: BLE001
        continue
    shutil.copy(bench_path, os.path.join(DST, "bench_%s_under_rocprof_trace.json" % wl))
    stats = find(wl + "_kernel_stats", ".csv")
    if stats:
        shutil.copy(stats, os.path.join(DST, "kernel_stats_%s.csv" % wl))
    tr_path = find(wl + "_trace_", "kernel_trace.csv")
    if tr_path:
        shutil.copy(tr_path, os.path.join(DST, "kernel_trace_%s_glx_only.csv" % wl))
    for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        p = find("%s_%s_" % (wl, d), "counter_collection.csv")
        if p:
            shutil.copy(p, os.path.join(DST, "pmc_%s_%s_glx_only.csv" % (name, wl)))
    out += ["## %s -- %s" % (wl, b["config"]["workload"][:160]), "",
            "bench.py under the trace: %.3f ms/step, %.3g edges/s; hop-2 aggregate launch %.3f ms by HIP events "
            "(`roofline.avg_launch_ms`)." % (b["ms_per_step"], b["value"], b["roofline"]["avg_launch_ms"]), "",
            "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
    for r in rows_of(os.path.join(DST, "kernel_stats_%s.csv" % wl)):
        if "glx" in r["Name"]:
            out.append("| `%s` | %s | %.3f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                             float(r["AverageNs"]) / 1e3, r["Percentage"]))
    # the dominant launches: largest grid of the aggregate kernel and of the sampler kernels
    tr = rows_of(os.path.join(DST, "kernel_trace_%s_glx_only.csv" % wl))

    def big(kind):
        rs = [r for r in tr if kind(r["Kernel_Name"])]
        if not rs:
            return None, []
        gs = lambda r: int(r.get("Grid_Size") or r["Grid_Size_X"])  # noqa: E731  (kernel trace: per-dimension columns)
        g = max(gs(r) for r in rs)
        sel = [r for r in rs if gs(r) == g]
        return short(sel[0]["Kernel_Name"]), [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sel]
    is_agg = lambda n: "glx_aggregate_kernel" in n  # noqa: E731
    is_smp = lambda n: "glx_sample_slots_kernel" in n or "glx_rwor" in n  # noqa: E731
    an, ad = big(is_agg)
    sn, sd = big(is_smp)
    if ad:
        out += ["", "Dominant aggregate launch `%s`: %d dispatches, average **%.3f ms** in the trace (bench.py's HIP events in the "
                "same run: %.3f ms)." % (an, len(ad), sum(ad) / len(ad), b["roofline"]["avg_launch_ms"])]
    if sd:
        rs_ms = (b.get("roofline_sampler") or {}).get("avg_launch_ms")
        out += ["Dominant sampler launch `%s`: %d dispatches, average **%.3f ms**%s." % (
            sn, len(sd), sum(sd) / len(sd), (" (HIP events: %.3f ms)" % rs_ms) if rs_ms else "")]

    def pmc(name):
        d = collections.OrderedDict()
        probe = []
        seen = collections.Counter()
        rows = rows_of(os.path.join(DST, "pmc_%s_%s_glx_only.csv" % (name, wl)))
        gmax = {}
        for r in rows:
            k = r["Kernel_Name"]
            gmax[k] = max(gmax.get(k, 0), int(r["Grid_Size"]))
        for r in rows:
            k = r["Kernel_Name"]
            if is_agg(k) and int(r["Grid_Size"]) == gmax[k]:
                seen[k] += 1
                if seen[k] > PMC_STEPS:
                    probe.append(float(r["Counter_Value"]))
                    continue
            if (is_agg(k) or is_smp(k)) and int(r["Grid_Size"]) != gmax[k]:
                continue  # hop-1 launches
            d.setdefault(k, []).append(float(r["Counter_Value"]))
        return d, probe
    (f, f_probe), (w, w_probe) = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
    if f:
        out += ["", "| kernel (largest launches only for the sampler / aggregate) | dispatches | FETCH_SIZE max KB | x2 -> read GB | "
                "WRITE_SIZE max KB | write GB |", "|---|---|---|---|---|---|"]
        for k in f:
            fm, wm = max(f[k]), max(w.get(k, [0]))
            out.append("| `%s` | %d | %.0f | %.3f | %.0f | %.3f |" % (short(k), len(f[k]), fm, fm * 2 * 1024 / 1e9, wm, wm * 1024 / 1e9))
    rec = {}
    ka = [k for k in f if is_agg(k)]
    if ka and ad:
        rd, wr = max(f[ka[0]]) * 2 * 1024, max(w.get(ka[0], [0])) * 1024
        alg = b["roofline"]["algorithmic_bytes_per_launch"]
        ms = sum(ad) / len(ad)
        rec.update(aggregate_hop2_bytes_per_launch=rd + wr, read_bytes_fetch_size_x2=rd, write_bytes=wr)
        out += ["", "Hop-2 aggregate launch: read %.2f GB + write %.2f GB = **%.2f GB of memory-side traffic per launch vs %.2f GB "
                "algorithmic**; at %.3f ms per launch that is %.2f TB/s of L2-side traffic (Infinity-Cache hits are inside "
                "FETCH_SIZE: MI355X_MICROARCH.md) and %.2f TB/s algorithmic." % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9, alg / 1e9, ms,
                                                                               (rd + wr) / ms / 1e9, alg / ms / 1e9)]
        if f_probe:
            prd, pwr = max(f_probe) * 2 * 1024, (max(w_probe) * 1024 if w_probe else 0.0)
            out += ["The same kernel on uniformly random rows (bench.py's cache-free leg, %d launches in the PMC pass): read %.2f GB "
                    "+ write %.2f GB = %.2f GB per launch = the algorithmic bytes: there, bytes moved are known exactly, which is "
                    "why `roofline.frac` is taken from that leg." % (len(f_probe), prd / 1e9, pwr / 1e9, (prd + pwr) / 1e9)]
    ks = [k for k in f if is_smp(k) and short(k) == sn]  # the launch the trace found dominant (hop 2), not hop 1's kernel
    if ks and sd and b.get("roofline_sampler"):
        rd, wr = max(f[ks[0]]) * 2 * 1024, max(w.get(ks[0], [0])) * 1024
        alg = b["roofline_sampler"]["algorithmic_bytes_per_launch"]
        rec.update(sample_hop2_bytes_per_launch=rd + wr, sample_read_bytes_fetch_size_x2=rd, sample_write_bytes=wr)
        out += ["Hop-2 sampler launch: read %.3f GB (FETCH_SIZE x2: an upper bound for 16 / 32-byte gathers, see r02) + write %.3f GB "
                "vs %.3f GB algorithmic: %.1fx -- every random record costs a whole line." % (rd / 1e9, wr / 1e9, alg / 1e9,
                                                                                            (rd + wr) / alg)]
    cal = [k for k in f if "glx_place_rows_kernel" in k]
    if cal and wl == "c3":
        known = 10_000_000 * 256 * 4
        out += ["", "Calibration in this run: `glx_place_rows_kernel` (the feature upload) reads and writes exactly %.3f GB; FETCH_SIZE "
                "reports %.3f GB (ratio %.3f -> x2), WRITE_SIZE %.3f GB (ratio %.3f -> x1)."
                % (known / 1e9, max(f[cal[0]]) * 1024 / 1e9, max(f[cal[0]]) * 1024 / known, max(w[cal[0]]) * 1024 / 1e9,
                   max(w[cal[0]]) * 1024 / known)]
    if rec:
        rec["source"] = "profiles/r03/pmc_FETCH_SIZE_%s_glx_only.csv + pmc_WRITE_SIZE_%s_glx_only.csv (FETCH_SIZE x2)" % (wl, wl)
        pmc_json["%s_b65536" % wl] = rec
    out.append("")

default = os.path.join(SRC, "bench_default.json")
if os.path.exists(default):
    shutil.copy(default, os.path.join(DST, "bench_c3_n1_final.json"))
    try:
        d = json.loads([ln for ln in open(default).read().splitlines() if ln.startswith("{")][-1])
        r = d["roofline"]
        out += ["## The driver's line (`python bench.py --steps 20 --warmup 5`: bench_c3_n1_final.json)", "",
                "- value %.4g edges/s, %.3f ms/step; verified_vs_oracle %s; gpu_over_cpu %.0f (reference C++ on %s threads)."
                % (d["value"], d["ms_per_step"], d.get("verified_vs_oracle"), d.get("gpu_over_cpu", float("nan")),
                   (d.get("cpu_baseline") or {}).get("cores")),
                "- roofline (hop-2 %s): achieved (algorithmic / timed launches) %.0f GB/s = %.2f of 8 TB/s, cache-assisted; "
                "**frac %.3f** (cache-free leg %.3f ms: %.0f GB/s); frac_compulsory %.3f; frac_traffic_offline %s; measured stream "
                "peak %s GB/s." % (d["config"]["aggregator"], r["achieved"], r["algorithmic_over_peak"], r["frac"],
                                   r.get("cache_free", {}).get("avg_launch_ms", float("nan")),
                                   r.get("cache_free", {}).get("achieved", float("nan")), r["frac_compulsory"],
                                   r.get("frac_traffic_offline"), {k: round(v) for k, v in r.get("peak_measured", {}).items()
                                                                  if isinstance(v, float)})]
        rs = d.get("roofline_sampler")
        if rs:
            out += ["- sampler (hop-2): %.3f ms, %.3g draws/s = %.2fx the uniform 32-byte gather rate of this box (%.3g records/s)."
                    % (rs["avg_launch_ms"], rs["draws_per_s"], rs.get("draws_over_uniform_gather_rate", float("nan")),
                       rs.get("gather32_uniform", {}).get("records_per_s", float("nan")))]
        for k, v in (d.get("other_configs") or {}).items():
            if "error" not in v:
                out += ["- other_configs.%s: %.3f ms/step, %.4g edges/s, roofline.frac %s, verified_vs_oracle %s."
                        % (k, v["ms_per_step"], v["value"], (v.get("roofline") or {}).get("frac"), v.get("verified_vs_oracle"))]
        e = d.get("edge_cut_world1") or {}
        for k, v in (e.get("placements") or {}).items():
            out += ["- edge_cut_world1.%s: %.3f ms/step%s." % (k, v["ms_per_step"], (
                "; %.1f count exchanges per step, host blocked in them %.3f ms per step (mostly waiting for the kernels queued "
                "ahead on the stream)" % (v.get("count_exchanges_per_step", v.get("host_syncs_per_step")),
                                          v.get("host_blocked_in_count_exchanges_ms_per_step", v.get("host_stall_ms_per_step")))
                if ("count_exchanges_per_step" in v or "host_syncs_per_step" in v) else ""))]
    except Exception as ex:  # noqa: BLE001
        out += ["(could not parse the default line: %r)" % ex]
open(os.path.join(DST, "SUMMARY.md"), "w").write("\n".join(out) + "\n")
if pmc_json:
    json.dump(pmc_json, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("\n".join(out)[:6000])
