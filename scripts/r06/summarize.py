"""gpurun_out/r06_prof/ (written by scripts/r06/profile.sh on the GPU box) -> profiles/r06/: per-workload kernel stats,
per-dispatch rows of the glx kernels, the CACHE-FREE launches of the roofline kernel in the same trace (what
`roofline.frac` is taken from: reproduced here from the trace alone, beside the run's own HIP-event figure), PMC
traffic + L2 hit rates of the dominant launches, SUMMARY_rocprof.md, and profiles/pmc_traffic.json (what bench.py
quotes as OFFLINE traffic).

    python scripts/r06/summarize.py
"""
import collections
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "r06_prof")
DST = os.path.join(ROOT, "profiles", "r06")
os.makedirs(DST, exist_ok=True)
PMC_STEPS = 6  # the PMC passes run --steps 5 --warmup 1; later hop-2-sized aggregate launches are the cache-free probe


def short(n):
    m = re.search(r"(glx_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n


def rows_of(path):
    return list(csv.DictReader(open(path))) if path and os.path.exists(path) else []


def src(name):
    p = os.path.join(SRC, name)
    return p if os.path.exists(p) else None


def keep(name, as_name):
    p = src(name)
    if p:
        shutil.copy(p, os.path.join(DST, as_name))
    return os.path.join(DST, as_name) if p else None


is_agg = lambda n: "glx_aggregate_grp_kernel" in n or "glx_aggregate_kernel" in n  # noqa: E731
is_smp = lambda n: "glx_sample_slots_kernel" in n or "glx_rwor" in n or "glx_topk" in n  # noqa: E731


def counters(path):
    """-> {kernel name: {counter: [values per dispatch, largest grid only for sampler / aggregate]}} and the probe launches
    (aggregate launches after the workload's own PMC_STEPS)."""
    rows = rows_of(path)
    gmax = {}
    for r in rows:
        gmax[r["Kernel_Name"]] = max(gmax.get(r["Kernel_Name"], 0), int(r["Grid_Size"]))
    # the hop-2 and hop-1 launches of one step may be different instantiations: keep the aggregate / sampler
    # instantiation with the largest grid, and only its largest launches
    agg_big = max([g for k, g in gmax.items() if is_agg(k)], default=0)
    smp_big = max([g for k, g in gmax.items() if is_smp(k)], default=0)
    d = collections.OrderedDict()
    probe = collections.defaultdict(list)
    seen = collections.Counter()
    for r in rows:
        k, c = r["Kernel_Name"], r["Counter_Name"]
        if is_agg(k) and int(r["Grid_Size"]) != agg_big:
            continue  # hop-1 launches
        if is_smp(k) and int(r["Grid_Size"]) != smp_big:
            continue
        if is_agg(k):
            seen[(k, c)] += 1
            if seen[(k, c)] > PMC_STEPS:
                probe[c].append(float(r["Counter_Value"]))
                continue
        d.setdefault(k, collections.defaultdict(list))[c].append(float(r["Counter_Value"]))
    return d, probe


TRACE_STEPS = 25  # the trace runs --steps 20 --warmup 5; later launches of the roofline kernel's shape are the probes
out = ["# r06 rocprofv3 summary (one MI355X)", "",
       "Commands: `scripts/r06/profile.sh` (gpurun).  Per workload w: `rocprofv3 --kernel-trace --stats "
       "--output-format csv -- python bench.py --workload w --steps 20 --warmup 5` (lean: no CPU baseline / host boundary / "
       "other configs; roofline probes ON, no counters: the launches of the roofline kernel's shape after the 25 of the "
       "workload's own steps are bench.py's cache-free leg -- warm-ups first, the timed ones last), then `--pmc FETCH_SIZE`, "
       "`--pmc WRITE_SIZE` and `--pmc TCC_HIT_sum TCC_MISS_sum`, "
       "each in a run of its own with `--kernel-trace` only (5 steps + 1 warm-up, roofline probes ON).  FETCH_SIZE x2 and WRITE_SIZE x1 as calibrated in profiles/r02 and r03 "
       "(MI355X_MICROARCH.md, HBM section).  c4's counter passes use `--kernel-include-regex` (counter collection over the "
       "1.6 B-edge build's dispatches crashes rocprofv3 itself).", ""]
pmc_json = {}
old_pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
for wl in ("c3", "c2", "c5", "c4"):  # whichever profile.sh was run for
    bench_path = src("%s_bench_trace.json" % wl)
    if not bench_path:
        continue
    try:
        b = json.load(open(src("%s_detail.json" % wl)))
        # the detail file is overwritten by every pass; the compact line of the trace run has the timings
        line = json.loads([ln for ln in open(bench_path).read().splitlines() if ln.startswith("{")][-1])
    except Exception:  # noqa: BLE001
        continue
    shutil.copy(bench_path, os.path.join(DST, "bench_%s_under_rocprof_trace.json" % wl))
    keep("%s_trace_t_kernel_stats.csv" % wl, "kernel_stats_%s.csv" % wl)
    keep("%s_trace_t_kernel_trace.csv" % wl, "kernel_trace_%s_glx_only.csv" % wl)
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
        keep("%s_%s_p_counter_collection.csv" % (wl, c), "pmc_%s_%s_glx_only.csv" % (c, wl))
    out += ["## %s -- %s" % (wl, line["config"]["workload"][:160]), "",
            "bench.py under the trace: %.3f ms/step, %.3g edges/s; roofline kernel launch %.3f ms by HIP events "
            "(`roofline.avg_launch_ms`)." % (line["ms_per_step"], line["value"], line["roofline"]["avg_launch_ms"]), "",
            "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
    for r in rows_of(os.path.join(DST, "kernel_stats_%s.csv" % wl)):
        if "glx" in r["Name"]:
            out.append("| `%s` | %s | %.3f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                             float(r["AverageNs"]) / 1e3, r["Percentage"]))
    tr = rows_of(os.path.join(DST, "kernel_trace_%s_glx_only.csv" % wl))

    def big(kind):
        rs = [r for r in tr if kind(r["Kernel_Name"])]
        if not rs:
            return None, []
        g = max(int(r["Grid_Size_X"]) for r in rs)
        sel = [r for r in rs if int(r["Grid_Size_X"]) == g]
        return short(sel[0]["Kernel_Name"]), [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sel]
    an, ad_all = big(is_agg)
    sn, sd = big(is_smp)
    ad, ad_probe = ad_all[:TRACE_STEPS], ad_all[TRACE_STEPS:]
    if ad:
        out += ["", "Longest aggregate launch `%s`: %d dispatches in the workload's own steps, average **%.3f ms** in the trace."
                % (an, len(ad), sum(ad) / len(ad))]
    cf = dict((line.get("roofline") or {}).get("cache_free") or {})
    try:  # the full record of the same run (the compact line drops launches_timed)
        cf.update((json.load(open(src("%s_detail_trace.json" % wl))).get("roofline") or {}).get("cache_free") or {})
    except Exception:  # noqa: BLE001
        pass
    if wl != "c5" and ad_probe and cf.get("avg_launch_ms"):
        # the cache-free leg: the same kernel, same request shape, ids uniform over the table -- its timed launches are
        # the LAST launches_timed dispatches of that shape in the process
        nt = int(cf.get("launches_timed", 5))
        timed = ad_probe[-nt:]
        alg_b = line["roofline"]["algorithmic_bytes_per_launch"]
        tr_ms = sum(timed) / len(timed)
        fr_trace = alg_b / (tr_ms * 1e-3) / 1e9 / 8000.0
        out += ["", "**`roofline.frac` from this trace alone.**  Cache-free launches (ids uniform over the table; the last %d "
                "dispatches of `%s` at the hop-2 grid): %s ms -> average **%.3f ms** in the trace; the same launches by HIP events "
                "inside bench.py (`roofline.cache_free.avg_launch_ms`): **%.3f ms**.  %.3f GB algorithmic / %.3f ms / 8 TB/s = "
                "**%.3f**; the line of this run says `roofline.frac` = **%.3f** (%+.1f %%)."
                % (nt, an, ", ".join("%.3f" % x for x in timed), tr_ms, cf["avg_launch_ms"], alg_b / 1e9, tr_ms, fr_trace,
                   line["roofline"]["frac"], (line["roofline"]["frac"] / fr_trace - 1) * 100)]
    if sd:
        out += ["Dominant sampler launch `%s`: %d dispatches, average **%.3f ms**." % (sn, len(sd), sum(sd) / len(sd))]
    f, f_probe = counters(os.path.join(DST, "pmc_FETCH_SIZE_%s_glx_only.csv" % wl))
    w, w_probe = counters(os.path.join(DST, "pmc_WRITE_SIZE_%s_glx_only.csv" % wl))
    t, t_probe = counters(os.path.join(DST, "pmc_TCC_HIT_sum_%s_glx_only.csv" % wl))
    rec = {}
    ka = [k for k in f if is_agg(k)]
    if ka and ad:
        k = ka[0]
        rd, wr = max(f[k]["FETCH_SIZE"]) * 2 * 1024, max(w.get(k, {}).get("WRITE_SIZE", [0])) * 1024
        n_ids = {"c3": 16_384_000, "c2": 9_830_400, "c4": 19_660_800, "c5": 6_553_600}[wl]
        n_seg = {"c3": 1_638_400, "c2": 983_040, "c4": 1_310_720, "c5": 655_360}[wl]
        D = line["config"].get("dim", 256)
        alg = n_ids * (4 * D + 12) + n_seg * (4 * D + 4)
        ms = sum(ad) / len(ad)
        rec.update(aggregate_hop2_bytes_per_launch=rd + wr, read_bytes_fetch_size_x2=rd, write_bytes=wr)
        out += ["", "Longest aggregate launch: read %.2f GB (FETCH_SIZE x2) + write %.2f GB = **%.2f GB of memory-side traffic per "
                "launch vs %.2f GB algorithmic** (%.2fx); at %.3f ms per launch that is %.2f TB/s memory-side (Infinity-Cache hits are "
                "inside FETCH_SIZE) and %.2f TB/s algorithmic." % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9, alg / 1e9, (rd + wr) / alg, ms,
                                                                  (rd + wr) / ms / 1e9, alg / ms / 1e9)]
        if k in t and t[k].get("TCC_HIT_sum"):
            hit, miss = sum(t[k]["TCC_HIT_sum"]), sum(t[k]["TCC_MISS_sum"])
            rec["aggregate_hop2_l2_hit_rate"] = hit / (hit + miss)
            out += ["L2: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) = **%.3f** over %d launches." % (hit / (hit + miss),
                                                                                                   len(t[k]["TCC_HIT_sum"]))]
        if f_probe.get("FETCH_SIZE"):
            prd = max(f_probe["FETCH_SIZE"]) * 2 * 1024
            pwr = max(w_probe["WRITE_SIZE"]) * 1024 if w_probe.get("WRITE_SIZE") else 0.0
            msg = ("The same kernel on uniformly random rows (bench.py's cache-free leg, %d launches in the PMC pass): read %.2f GB + "
                   "write %.2f GB = %.2f GB per launch" % (len(f_probe["FETCH_SIZE"]), prd / 1e9, pwr / 1e9, (prd + pwr) / 1e9))
            if t_probe.get("TCC_HIT_sum"):
                ph, pm = sum(t_probe["TCC_HIT_sum"]), sum(t_probe["TCC_MISS_sum"])
                msg += "; L2 hit rate there %.3f" % (ph / (ph + pm))
            out += [msg + "."]
        if wl == "c5":  # the roofline kernel of c5 is the u-i hop's launch: the middle-sized aggregate grid
            grids = sorted({int(r["Grid_Size"]) for r in rows_of(os.path.join(DST, "pmc_FETCH_SIZE_c5_glx_only.csv"))
                            if is_agg(r["Kernel_Name"])})
            rows_f = [r for r in rows_of(os.path.join(DST, "pmc_FETCH_SIZE_c5_glx_only.csv")) if is_agg(r["Kernel_Name"])]
            rows_w = [r for r in rows_of(os.path.join(DST, "pmc_WRITE_SIZE_c5_glx_only.csv")) if is_agg(r["Kernel_Name"])]
            # per step: i-s (largest grid), u-i, u-s (the two smaller launches have the same grid: take them in order)
            small_f = [float(r["Counter_Value"]) for r in rows_f if int(r["Grid_Size"]) != max(grids)]
            small_w = [float(r["Counter_Value"]) for r in rows_w if int(r["Grid_Size"]) != max(grids)]
            if small_f and small_w:
                ui_f, ui_w = small_f[0::2][:PMC_STEPS], small_w[0::2][:PMC_STEPS]
                rec["aggregate_item_bytes_per_launch"] = max(ui_f) * 2 * 1024 + max(ui_w) * 1024
                out += ["u-i hop launch (the roofline kernel: 655,360 ids over the 9.2 GB item table): read %.3f GB + write %.3f GB "
                        "= %.3f GB vs %.3f GB algorithmic." % (max(ui_f) * 2 * 1024 / 1e9, max(ui_w) * 1024 / 1e9,
                                                               rec["aggregate_item_bytes_per_launch"] / 1e9,
                                                               (655_360 * (4 * D + 12) + 65_536 * (4 * D + 4)) / 1e9)]
    ks = [k for k in f if is_smp(k) and short(k) == sn]
    if ks and sd:
        k = ks[0]
        rd, wr = max(f[k]["FETCH_SIZE"]) * 2 * 1024, max(w.get(k, {}).get("WRITE_SIZE", [0])) * 1024
        rec.update(sample_hop2_bytes_per_launch=rd + wr, sample_read_bytes_fetch_size_x2=rd, sample_write_bytes=wr)
        out += ["Hop-2 sampler launch: read %.3f GB (FETCH_SIZE x2: an upper bound for 16 / 32-byte gathers) + write %.3f GB."
                % (rd / 1e9, wr / 1e9)]
    if rec:
        rec["source"] = "profiles/r06/pmc_{FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum}_%s_glx_only.csv (FETCH_SIZE x2)" % wl
        pmc_json["%s_b65536" % wl] = rec
    out.append("")

open(os.path.join(DST, "SUMMARY_rocprof.md"), "w").write("\n".join(out) + "\n")
if pmc_json:
    try:
        merged = json.load(open(old_pmc))
    except Exception:  # noqa: BLE001
        merged = {}
    merged.update(pmc_json)
    json.dump(merged, open(old_pmc, "w"), indent=1)
print("\n".join(out)[:9000])
