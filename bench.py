#!/usr/bin/env python
"""bench.py -- headline benchmark of the glx sampling + aggregation hot path.

One "step" = one pass of the hot path over one batch of B0 seed vertices:
  hop-1 sample (B0 rows x k1) -> hop-2 sample (B0*k1 rows x k2)
  -> aggregate the hop-2 neighbours into B0*k1 segments
  -> aggregate the hop-1 neighbours into B0 segments,
all through the C-ABI (include/glx.h) with inputs resident in HBM.  Every
sampled neighbour is aggregated exactly once, so sampled-edges/s ==
aggregated-vertices/s for the whole step; the per-phase rates are reported too.

Default workload = BASELINE.json configs[2], the config its metric ("100M-edge
graph") is quoted on: RMAT (0.57,0.19,0.19,0.05) 10M nodes / 100M edges,
EdgeWeightSampler (alias) fanout [25,10], MaxAggregator, dim=256.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the graph is edge-cut partitioned (llabs(v) % N), every
rank drives its own batch of B0 seeds (weak scaling) and each hop's requests are
routed to the owning shard and back with RCCL all-to-all (graph-learn_amd/dist.py).
Features are replicated by one load-time RCCL all-gather when the table fits
(--features auto: V*D*4 <= 96 GiB of the 288 GB), else kept sharded with a
per-request halo exchange (--features sharded).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import glx  # noqa: E402
import synth  # noqa: E402

WORKLOADS = {
    # name: (V, E, sampler, fanout, aggregator, dim, graph seed, description)
    "c3": (10_000_000, 100_000_000, "EdgeWeightSampler", (25, 10), "MaxAggregator", 256, 4,
           "BASELINE configs[2]: RMAT power-law 10M nodes / 100M edges, WeightedSampler(alias) "
           "fanout [25,10], MaxAggregator, dim=256"),
    "c2": (2_400_000, 62_000_000, "RandomWithoutReplacementSampler", (15, 10), "MeanAggregator", 128, 2,
           "BASELINE configs[1] shape: RMAT 2.4M nodes / 62M edges, RandomWithoutReplacement "
           "fanout [15,10], MeanAggregator, dim=128"),
    "c4": (111_000_000, 1_600_000_000, "RandomSampler", (20, 15), "MeanAggregator", 128, 6,
           "BASELINE configs[3] shape on ONE GPU: RMAT 111M nodes / 1.6B edges (ogbn-papers100M-sized), "
           "RandomSampler fanout [20,15], MeanAggregator, dim=128"),
    "tiny": (200_000, 2_000_000, "EdgeWeightSampler", (25, 10), "MaxAggregator", 256, 4,
             "tiny smoke workload (not a benchmark)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)

LINE_LIMIT = 4096  # bytes of the ONE stdout line: the driver keeps ~9 KB of stdout tail and parses the line from it


def _short(text, n):
    text = " ".join(str(text).split())
    return text if len(text) <= n else text[:n - 3] + "..."


def _num(x, digits=6):
    """Floats rounded to `digits` significant digits (the line is for a parser, the detail file keeps everything)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    return x


def _pick(d, keys):
    return {k: _num(d[k]) for k in keys if d is not None and k in d and d[k] is not None}


def compact_roofline(r):
    if not r:
        return None
    out = _pick(r, ("bound", "peak", "unit", "achieved", "frac", "algorithmic_gbs", "memory_side_gbs", "algorithmic_over_peak",
                    "frac_compulsory", "traffic", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch",
                    "compulsory_bytes_per_launch", "frac_of_gather_ceiling", "l2_hit_rate",
                    "share_of_step", "offline_tcp_busy_frac", "offline_l1_hit_rate"))
    out["kernel"] = _short(r.get("kernel", ""), 72)
    # (achieved_basis stays in the detail record: frac_basis below says which launches achieved / frac come from)
    if r.get("frac_basis"):
        # whole sentences only: the basis of the fraction must not be cut mid-word
        out["frac_basis"] = r["frac_basis"].split(" (")[0].split(";")[0]
    if r.get("traffic_source"):
        out["traffic_source"] = _short(r["traffic_source"], 60)
    # (hbm_bytes_bracket = [compulsory_bytes_per_launch, traffic] and traffic_over_algorithmic stay in the detail record:
    # both follow from keys that are in the line)
    cf = r.get("cache_free")
    if cf:
        out["cache_free"] = _pick(cf, ("avg_launch_ms", "achieved", "frac", "frac_of_measured_stream_peak",
                                        "frac_of_measured_row_gather"))
    pk = r.get("peak_measured")
    if pk:
        out["peak_measured_gbs"] = {k: _num(v, 4) for k, v in pk.items() if isinstance(v, float)}
    return out


def compact_cpu(c):
    if not c:
        return None
    out = _pick(c, ("value", "unit", "cores", "threads", "nproc", "kind", "sampling_edges_per_s", "aggregation_vertices_per_s",
                    "storage_mode", "edges_built", "feature_rows"))
    out["sample"] = _short(c.get("sample", ""), 160)
    if c.get("thread_sweep"):
        out["thread_sweep"] = [_pick(t, ("cores", "value")) for t in c["thread_sweep"]]
    if c.get("modes"):
        out["modes"] = {m: {t: _num(v, 4) for t, v in tv.items()} for m, tv in c["modes"].items()}
    return out


def compact_line(res, detail_path=None, limit=LINE_LIMIT):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` + the verification flags, at most `limit`
    bytes.  Everything else this run measured (probes, per-leg statistics, notes) goes to the detail file and stderr."""
    cfg = res.get("config") or {}
    config = {"workload": _short(cfg.get("workload", ""), 200)}
    for k in ("seeds_per_step_per_gpu", "fanout", "sampler", "aggregator", "dim", "nodes", "edges", "hipgraph_step"):
        if k in cfg and cfg[k] is not None:
            config[k] = cfg[k]
    if cfg.get("parallelism"):
        config["parallelism"] = _short(cfg["parallelism"], 120)
    for k in ("seeds", "segment_ids"):
        if cfg.get(k):
            config[k] = cfg[k].split(" (")[0]  # the first clause, whole
    line = {"metric": _short(res.get("metric", ""), 90)}
    for k in ("value", "unit", "n_gpus", "ranks", "rccl_ranks", "transport", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"):
        if k in res:
            line[k] = _num(res[k], 9)
    line["config"] = config
    if "error" in res:
        line["error"] = _short(res["error"], 200)
    line["roofline"] = compact_roofline(res.get("roofline"))
    line["cpu_baseline"] = compact_cpu(res.get("cpu_baseline"))
    for k in ("gpu_over_cpu", "verified_vs_oracle", "verified_sharded_equals_unpartitioned"):
        if res.get(k) is not None:
            line[k] = _num(res[k], 5)
    if res.get("request_shapes"):
        line["request_shapes"] = {k: _num(v, 5) for k, v in res["request_shapes"].items()}
    optional = []  # dropped from the back if the line would outgrow the limit
    if res.get("phases"):
        line["phases"] = _pick(res["phases"], ("sampling_kernels_ms_per_step", "aggregation_kernels_ms_per_step"))
        optional.append("phases")
    if res.get("roofline_sampler"):
        rs = res["roofline_sampler"]
        line["roofline_sampler"] = dict(_pick(rs, ("achieved", "frac", "traffic", "avg_launch_ms", "frac_of_gather_ceiling")),
                                        kernel=_short(rs.get("kernel", ""), 60))
        optional.append("roofline_sampler")
    if res.get("roofline_u_i"):
        line["roofline_u_i"] = compact_roofline(res["roofline_u_i"])
        optional.append("roofline_u_i")
    if res.get("placements"):
        line["placements"] = {name: _pick(leg, ("ms_per_step", "value")) for name, leg in res["placements"].items()}
        optional.append("placements")
    if res.get("verified_legs"):
        line["verified_legs"] = res["verified_legs"]
        optional.append("verified_legs")
    if res.get("other_configs"):
        oc = {}
        for name, rec in res["other_configs"].items():
            if "error" in rec:
                oc[name] = {"error": _short(rec["error"], 60)}
                continue
            roof = rec.get("roofline") or {}
            oc[name] = dict(_pick(rec, ("ms_per_step", "value")), bound=roof.get("bound"), frac=_num(roof.get("frac"), 4),
                            frac_basis=(roof.get("frac_basis") or "").split(":")[0].replace(" of this run", ""),
                            traffic=_num(roof.get("traffic"), 4),
                            verified=rec.get("verified_vs_oracle"))
        line["other_configs"] = oc
        optional.append("other_configs")
    p8 = res.get("edge_cut_p8_one_gpu")
    if isinstance(p8, dict) and p8.get("ms_per_rank_step"):
        line["edge_cut_p8_one_gpu"] = dict(_pick(p8, ("ms_per_step_all_ranks", "ms_per_rank_step")),
                                           verified=p8.get("answers_equal_unpartitioned"))
        optional.append("edge_cut_p8_one_gpu")
    for k in ("host_boundary", "small_batches"):
        v = res.get(k)
        if isinstance(v, dict):
            if k == "host_boundary" and v.get("edges_per_s"):
                line["host_boundary_edges_per_s"] = _num(v["edges_per_s"], 4)
                optional.append("host_boundary_edges_per_s")
            if k == "small_batches":
                line["small_batches"] = {b: _pick(r, ("ms_per_step", "value")) for b, r in v.items() if isinstance(r, dict)}
                optional.append("small_batches")
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, separators=(", ", ": "))
    trimmed = []
    for k in ("small_batches", "host_boundary_edges_per_s", "edge_cut_p8_one_gpu", "verified_legs", "phases", "roofline_sampler", "roofline_u_i",
              "placements", "other_configs"):
        if len(text) <= limit:
            break
        if k in optional and k in line:
            del line[k]
            trimmed.append(k)
            line["trimmed"] = trimmed
            text = json.dumps(line, separators=(", ", ": "))
    if len(text) > limit:  # last resort: the contract's keys alone
        for k in list(line):
            if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail", "error"):
                del line[k]
        line["config"] = {"workload": _short(config["workload"], 100)}
        text = json.dumps(line, separators=(", ", ": "))
    return text


def emit_result(result_out, res, args):
    """Rank 0: the full record to the detail file (+ stderr), the compact line to the real stdout."""
    path = getattr(args, "detail_out", None) or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as fh:
            json.dump(res, fh, indent=1)
            fh.write("\n")
    except OSError as ex:
        log("could not write %s: %r" % (path, ex))
        path = None
    sys.stderr.write("[bench-detail] " + json.dumps(res) + "\n")
    sys.stderr.flush()
    result_out.write(compact_line(res, os.path.relpath(path, ROOT) if path and path.startswith(ROOT) else path) + "\n")
    result_out.flush()


def n1_extras(args, world, sharded, workload=None, batch=None):
    """Which side measurements of the N = 1 line run.  They are N = 1 ONLY: with more than one rank (or the sharded
    code path) none of them runs, so that an 8-rank job spends its lease on the placements (VERDICT r04 item 5;
    tests/test_bench_helpers.py pins it)."""
    workload = args.workload if workload is None else workload
    batch = args.batch if batch is None else batch
    one = world == 1 and not sharded
    headline_c3 = one and workload == "c3" and batch == 65536
    return {"cpu_baseline": one and args.cpu_baseline == "on",
            "host_boundary": one and args.host_boundary == "on",
            "roofline_probes": one and args.roofline_probes == "on",
            "verify_oracle": one and args.verify_oracle == "on",
            "request_shape_legs": one and args.request_shape_legs == "on",
            "small_batches": one and args.small_batches == "on" and batch == 65536,
            "edge_cut_probe": headline_c3 and args.edge_cut_probe == "on",
            "other_configs": headline_c3 and bool(args.other_configs)}


def comm_facts(comm, share_device):
    """n_gpus / ranks / rccl_ranks of the line come from the communicator the run actually used (glx_comm_info), not from
    --gpus or the launcher's environment."""
    names = {glx.COMM_RCCL: "rccl (xGMI)", glx.COMM_LOCAL: "in-process threads", glx.COMM_CALLBACKS: "host-staged (gloo test rig)"}
    return {"n_gpus": comm.world, "ranks": comm.world, "rccl_ranks": comm.world if comm.transport == glx.COMM_RCCL else 0,
            "transport": names.get(comm.transport, str(comm.transport)) + (", every rank on ONE device (--share-device)"
                                                                          if share_device else "")}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` (one process per GPU, rendezvous on 127.0.0.1) -- the reference's fan-out needs no external
    launcher either (core/runner/op_runner.h:49-90).  Too few devices: ONE JSON error line, non-zero exit."""
    import socket
    have = 0
    try:
        have = glx.device_count()
    except glx.GlxError:
        have = 0
    if not args.share_device and have < args.gpus:
        print(json.dumps({"metric": "sampled-edges/sec + aggregated-vertices/sec", "value": None, "unit": "edges/s",
                          "n_gpus": have, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": args.workload},
                          "error": "--gpus %d asked for, %d visible (glx_device_count): nothing measured" % (args.gpus, have)}),
              flush=True)
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no launcher in the environment: re-executing as %s" % " ".join(cmd))
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def cpu_baseline(wl, src, dst, weight, args, seed_pool=None):
    """Times the reference's own CPU path (oracle/_ref, built from the reference's sources) on this box's host cores:
    the WHOLE graph of the workload and its whole feature table, under both of the reference's storage modes (2 = its
    default: vector-of-vectors adjacency + per-node attribute objects; 3 = compressed: CSR + flat attributes), at
    T = 1, 32 (InterThreadNum's default, config.cc:90) and nproc request threads -- SURVEY 8(d).  One worker process
    per storage mode: the two graphs are built side by side (the build is single-threaded), the timed sections take
    turns (a file lock), so the threads of one never compete with the other's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_bindings import have_ref
    V, E, sampler, (k1, k2), agg, D = wl[:6]
    if not have_ref():
        return cpu_baseline_port(wl, src, dst, weight, args, seed_pool)
    import shutil
    import subprocess
    import tempfile
    t_all = time.time()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    work = tempfile.mkdtemp(prefix="glx_cpu_", dir=base)
    try:
        np.save(os.path.join(work, "src.npy"), src.cpu().numpy())
        np.save(os.path.join(work, "dst.npy"), dst.cpu().numpy())
        if weight is not None:
            np.save(os.path.join(work, "w.npy"), weight.cpu().numpy())
        np.save(os.path.join(work, "pool.npy"), seed_pool if seed_pool is not None else np.arange(V, dtype=np.int64))
        open(os.path.join(work, "lock"), "w").close()
        modes = [int(m) for m in str(args.cpu_storage_modes).split(",") if m.strip()]
        procs = []
        for m in modes:
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(m), "--cpu-worker-dir", work,
                   "--workload", args.workload, "--cpu-time-budget", str(args.cpu_time_budget),
                   "--cpu-seeds-per-request", str(args.cpu_seeds_per_request), "--cpu-thread-sweep", args.cpu_thread_sweep,
                   "--cpu-edge-limit", str(args.cpu_edge_limit)]
            procs.append((m, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        recs = {}
        for m, pr in procs:
            try:
                so, se = "", ""
                so, se = pr.communicate(timeout=args.cpu_wall_limit)
                recs[m] = json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1])
            except Exception as ex:  # noqa: BLE001 -- a mode that fails is reported, the other still counts
                pr.kill()
                recs[m] = {"error": "%r (exit code %s): %s" % (ex, pr.returncode, (se or "")[-300:])}
            log("cpu baseline, StorageMode %d: %s" % (m, {k: v for k, v in recs[m].items() if k != "legs"}))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    good = {m: r for m, r in recs.items() if "legs" in r}
    if not good:
        return {"error": "no storage mode finished: %s" % recs, "value": None}
    # the baseline the GPU is compared with: the reference's best configuration of the six
    best = max(((leg["value"], m, leg) for m, r in good.items() for leg in r["legs"]), key=lambda x: x[0])
    value, mode, leg = best
    r0 = good[mode]
    return {
        "value": value, "unit": "edges/s", "cores": leg["threads"], "threads": leg["threads"], "nproc": os.cpu_count(),
        "kind": "reference", "storage_mode": mode,
        "sampling_edges_per_s": leg["sampling_edges_per_s"], "aggregation_vertices_per_s": leg["aggregation_vertices_per_s"],
        "modes": {str(m): {str(l["threads"]): l["value"] for l in r["legs"]} for m, r in good.items()},
        "thread_sweep": [{"cores": l["threads"], "value": l["value"], "sampling_edges_per_s": l["sampling_edges_per_s"],
                          "aggregation_vertices_per_s": l["aggregation_vertices_per_s"]} for l in r0["legs"]],
        "edges_built": r0["edges_built"], "feature_rows": r0["feature_rows"],
        "feature_rows_by_mode": {str(m): r["feature_rows"] for m, r in good.items()},
        "build_s": {str(m): r["build_s"] for m, r in good.items()},
        "sample": ("reference C++ (oracle/_ref) %s [%d,%d] + %s, whole graph (%d of %d edges) + whole feature table (%d x %d), "
                   "StorageMode 2 and 3 (3: the %d rows its int32 attribute offsets reach) x T = %s request threads (one "
                   "request per thread, %d seeds/request); value = best of those = mode %d at T = %d; per leg ~%.0fs sampling + ~%.0fs aggregation timed; value = 1/(1/sampling + "
                   "1/aggregation); aggregation ids = destinations of random edges (the in-degree-biased mix a sampler "
                   "returns)" % (sampler, k1, k2, agg, r0["edges_built"], E, r0["feature_rows"], D, min(V, (2 ** 31 - 1) // D),
                                 "/".join(str(l["threads"]) for l in r0["legs"]), args.cpu_seeds_per_request, mode,
                                 leg["threads"], args.cpu_time_budget, args.cpu_time_budget)),
        "errors": {str(m): r["error"] for m, r in recs.items() if "error" in r} or None,
        "wall_s": time.time() - t_all,
    }


def cpu_worker(args):
    """One storage mode of cpu_baseline(), in a process of its own (StorageMode is a process-global flag of the
    reference)."""
    import fcntl
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_bindings import RefLib, _p
    V, E, sampler, (k1, k2), agg, D = WORKLOADS[args.workload][:6]
    mode, work = args.cpu_worker, args.cpu_worker_dir
    src = np.load(os.path.join(work, "src.npy"), mmap_mode="r")
    dst = np.load(os.path.join(work, "dst.npy"), mmap_mode="r")
    w = np.load(os.path.join(work, "w.npy"), mmap_mode="r") if os.path.exists(os.path.join(work, "w.npy")) else None
    pool = np.load(os.path.join(work, "pool.npy"))
    ref = RefLib(storage_mode=mode, padding_mode=1)
    t0 = time.time()
    n_edges = min(E, args.cpu_edge_limit) if args.cpu_edge_limit > 0 else E
    for lo in range(0, n_edges, 5_000_000):  # insertion (= edge id) order, as the reference's loader adds them
        hi = min(lo + 5_000_000, n_edges)
        ref.L.glref_add_edges(ref.h, b"e", _p(np.ascontiguousarray(src[lo:hi])), _p(np.ascontiguousarray(dst[lo:hi])),
                              _p(np.ascontiguousarray(w[lo:hi])) if w is not None else None, hi - lo)
    ref.L.glref_build_graph(ref.h, b"e")
    block = np.random.default_rng(5).random((1 << 20, D), dtype=np.float32) * 2 - 1
    # StorageMode 3 keeps the float attributes of a node type in ONE flat array and computes a row's offset as an
    # int32 product (compressed_memory_node_storage.cc:160-163: `it->second * side_info_.f_num`, both int32): rows past
    # (2^31 - 1) / f_num are out of its reach (the whole 10 M x 256 table of C3 crashes it).  That mode therefore gets
    # the largest table it can address, and its aggregation ids are folded into it.
    rows = V if mode != 3 else min(V, (2 ** 31 - 1) // D)
    for lo in range(0, rows, 1 << 20):  # the whole table; every 1 M-row block gets the same values (timing only)
        hi = min(lo + (1 << 20), rows)
        ref.L.glref_add_nodes(ref.h, b"n", _p(np.arange(lo, hi, dtype=np.int64)), _p(block), hi - lo, D)
    ref.L.glref_build_nodes(ref.h, b"n")
    build_s = time.time() - t0
    B = args.cpu_seeds_per_request
    rng = np.random.default_rng(123 + mode)
    out = ctypes.c_int64()
    threads = []
    for tc in [int(x) for x in args.cpu_thread_sweep.split(",") if x.strip()]:
        tc = tc if tc > 0 else (os.cpu_count() or 1)
        if tc not in threads:
            threads.append(tc)
    legs = []
    n_ids = B * k1 * k2
    with open(os.path.join(work, "lock")) as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # timed sections take turns across the workers
        for tc in threads:
            seeds = pool[rng.integers(0, pool.shape[0], B * tc)]
            d1 = ref.L.glref_time_sample_2hop(ref.h, b"e", sampler.encode(), _p(seeds), B, k1, k2, 1, tc, ctypes.byref(out))
            reps = int(max(1, min(64, args.cpu_time_budget / max(d1, 1e-3))))
            seeds = pool[rng.integers(0, pool.shape[0], B * tc * reps)]
            ds = ref.L.glref_time_sample_2hop(ref.h, b"e", sampler.encode(), _p(seeds), B, k1, k2, reps, tc, ctypes.byref(out))
            r_s = out.value / ds
            ids = np.ascontiguousarray(dst[np.sort(rng.integers(0, n_edges, n_ids * tc))]) % rows
            rng.shuffle(ids)
            d2 = ref.L.glref_time_aggregate(ref.h, b"n", agg.encode(), _p(ids), n_ids, k2, 1, tc, ctypes.byref(out))
            areps = int(max(1, min(64, args.cpu_time_budget / max(d2, 1e-3))))
            ids = np.ascontiguousarray(dst[np.sort(rng.integers(0, n_edges, n_ids * tc * areps))]) % rows
            rng.shuffle(ids)
            da = ref.L.glref_time_aggregate(ref.h, b"n", agg.encode(), _p(ids), n_ids, k2, areps, tc, ctypes.byref(out))
            r_a = out.value / da
            legs.append({"threads": tc, "value": 1.0 / (1.0 / r_s + 1.0 / r_a), "sampling_edges_per_s": r_s,
                         "aggregation_vertices_per_s": r_a, "requests_per_thread": [reps, areps], "timed_s": [ds, da]})
        fcntl.flock(lock, fcntl.LOCK_UN)
    ref.close()
    print(json.dumps({"storage_mode": mode, "build_s": build_s, "edges_built": n_edges, "feature_rows": rows, "legs": legs}), flush=True)


def cpu_baseline_port(wl, src, dst, weight, args, seed_pool=None):
    """Fallback when oracle/_ref (the reference's own code) is not on this box: the
    C restatement (oracle/glx_oracle.c) timed single-threaded with the reference's cost
    model switched on (per-row, per-request alias rebuild: edge_weight_sampler.cc:78-92)."""
    from oracle_bindings import Oracle
    import synth as _synth
    V, E, sampler, (k1, k2), agg, D = wl[:6]
    t_all = time.time()
    orc = Oracle()
    Ec = min(E, 20_000_000)  # bounded sample: a prefix of the edge stream
    s_h = src[:Ec].cpu().numpy()
    d_h = dst[:Ec].cpu().numpy()
    w_h = weight[:Ec].cpu().numpy() if weight is not None else None
    rp, col, eid, ws = _synth.csr_numpy(s_h, d_h, w_h, V)
    g = dict(row_ptr=rp, col=col, eid=eid, weight=ws)
    orc.L.glxo_set_reference_cost_model(1)
    rng = np.random.default_rng(123)
    B = max(8, args.cpu_seeds_per_request // 8)
    edges, t0 = 0, time.time()
    while time.time() - t0 < args.cpu_time_budget:
        pool = seed_pool if seed_pool is not None else np.arange(V, dtype=np.int64)
        seeds = pool[rng.integers(0, pool.shape[0], B)]
        n1, _ = orc.sample(g, sampler, seeds, k1, seed=1, call_counter=edges)
        n2, _ = orc.sample(g, sampler, n1.reshape(-1), k2, seed=1, call_counter=edges + 1)
        edges += n1.size + n2.size
    dts = time.time() - t0
    orc.L.glxo_set_reference_cost_model(0)
    Vc = min(V, 1_000_000)
    X = (np.random.default_rng(5).random((Vc, D), dtype=np.float32) * 2 - 1)
    ids = (n2.reshape(-1) % Vc).astype(np.int64)
    seg = (np.arange(ids.shape[0]) // k2).astype(np.int32)
    verts, t0 = 0, time.time()
    while time.time() - t0 < args.cpu_time_budget / 2:
        orc.aggregate(X, agg, ids, seg, ids.shape[0] // k2)
        verts += ids.shape[0]
    dta = time.time() - t0
    r_s, r_a = edges / dts, verts / dta
    return {"value": 1.0 / (1.0 / r_s + 1.0 / r_a), "unit": "edges/s", "cores": 1, "threads": 1,
            "nproc": os.cpu_count(), "kind": "port",
            "sampling_edges_per_s": r_s, "aggregation_vertices_per_s": r_a,
            "sample": ("oracle/glx_oracle.c (C restatement, reference cost model: alias table rebuilt per row per "
                       "request) single-threaded; %s [%d,%d] + %s; graph = first %d of %d edges; %d seeds/request; "
                       "%.1fs sampling + %.1fs aggregation timed; features = %d rows x %d"
                       % (sampler, k1, k2, agg, Ec, E, B, dts, dta, Vc, D)),
            "wall_s": time.time() - t_all}


def host_boundary_rate(args):
    """What a graph-learn caller gets THROUGH the drop-in boundary: host_path_bench issues the same 2-hop
    EdgeWeightSampler [25,10] + MaxAggregator (dim 256) requests from T host threads through the C++ operator
    API (OpFactory::Create(name)->Process(req, res)) with host buffers -- one request per pool thread, the
    reference's concurrency model (in_memory_service.cc:64-71) -- so every id and every embedding crosses PCIe.
    `value` above is the device-resident rate; this is the PCIe-inclusive one."""
    import subprocess
    exe = os.path.join(ROOT, "graph-learn_amd", "lib", "host_path_bench")
    if not os.path.exists(exe):
        return None
    t_hb = time.time()
    threads, B, reps = args.host_boundary_threads, 1024, 40  # 10 per thread: start / tail skew of the pool costs 20 %
    try:
        # the headline graph's shape: RMAT scale 24 folded onto 10 M nodes, 100 M edges, dim 256
        r = subprocess.run([exe, str(threads), str(B), str(reps), "24", "100000000", "256", "10000000"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        rec = json.loads(line)
    except Exception as ex:  # noqa: BLE001 -- never lose the headline line
        return {"error": repr(ex)}
    return {"edges_per_s": rec["sampled_edges_per_s_host_pointer_path"], "threads": threads, "seeds_per_request": B,
            "requests_per_thread": reps, "wall_s_incl_build": time.time() - t_hb,
            "graph": "RMAT %d nodes / %d edges, dim %d: the headline workload's shape, built through the C++ store API" % (rec["nodes"], rec["edges"], rec["dim"]),
            "response_bytes_per_request": B * 25 * 16 + B * 250 * 16 + (B * 25 + B) * (256 * 4 + 4),
            "note": "PCIe-inclusive: requests and responses are host tensors of the C++ operator API; response blocks "
                    "are pinned (glx_host_register) so the device->host copies are single DMA transfers"}


def edge_cut_world1(args):
    """The N > 1 code path with one rank, in a process of its own (this one keeps its store resident): same workload,
    same step count, both feature placements, answers verified bit for bit against the unpartitioned operators."""
    import subprocess
    env = dict(os.environ, GLX_DIST_NO_SHORTCUT="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-sharded", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--cpu-baseline", "off", "--host-boundary", "off", "--roofline-probes", "off",
           "--edge-cut-probe", "off", "--small-batches", "off", "--other-configs", "", "--verify"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as ex:  # noqa: BLE001 -- never lose the headline line
        return {"error": repr(ex)}
    return {"placements": rec.get("placements"), "verified_equals_unpartitioned": rec.get("verified_sharded_equals_unpartitioned"),
            "verified_legs": rec.get("verified_legs"), "replicated_per_gpu": rec.get("config", {}).get("replicated_per_gpu"),
            "hot_fraction": args.hot_fraction, "halo_exchange_hop2": rec.get("halo_exchange_hop2"),
            "sampling_exchange_hop2": rec.get("sampling_exchange_hop2"),
            "note": "world size 1 over RCCL, generic path (GLX_DIST_NO_SHORTCUT=1): partition, self-exchange, resolve + "
                    "dedup, halo slots, 3-source reduce, 3-stage pipeline -- no link time; compare with ms_per_step above"}


def edge_cut_p8_one_gpu(args):
    """One GPU playing eight ranks (scripts/edge_cut_p8_probe.py in a process of its own): eight threads, each a rank with its
    own shard of the headline graph and features, its own 65,536-seed step per iteration, the in-process transport (device
    copies where xGMI would be), hot-row + graph replicas, merged aggregation, speculation ledger; every rank's answers
    compared with the unpartitioned operators.  NOT a multi-GPU measurement: the eight ranks' kernels share one GPU and the
    'links' are copies on it -- wall time of a step of ALL ranks / 8 bounds one rank's GPU work per step from above
    (DESIGN 12 has the kernel-level figure from the same rig under rocprofv3)."""
    import subprocess
    exe = os.path.join(ROOT, "scripts", "edge_cut_p8_probe.py")
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, GRAPH_REPLICA="1", MERGED="1", LEDGER="1")
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, exe, "8", str(args.hot_fraction), "12"], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=420)
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{\"p8_probe\"")][-1])["p8_probe"]
    except Exception as ex:  # noqa: BLE001 -- never lose the headline line
        return {"error": repr(ex)}
    rec["wall_s_incl_build"] = time.time() - t0
    rec["note"] = ("eight ranks as threads sharing ONE GPU, in-process transport: not a multi-GPU number; ms_per_rank_step = "
                   "wall time of a step of all ranks / 8, transport copies and the eight ranks' contention included")
    return rec


def bench_c5(args, dev, result_out, world=1, rank=0, sharded=False):
    """BASELINE configs[4] shape: heterogeneous user-item-shop graph, three
    weighted edge types (u-i 300M, i-s 100M, u-s 100M edges over 40M / 9M / 1M nodes),
    per-edge-type TopkSampler (k = 10, 10, 5) + type-wise SumAggregator, dim=256.
    One storage handle per type replaces the reference's HeterDispatcher
    (core/graph/heter_dispatcher.h:44-56).  With N > 1 ranks every edge type is edge-cut by
    llabs(src) % N (one ShardedStore per type: RCCL all-to-all per hop) and the two feature
    tables (9.2 GB + 1 GB) are replicated."""
    D, B0 = 256, args.batch
    extras = n1_extras(args, world, sharded)
    sc = max(1, args.c5_scale)
    n_user, n_item, n_shop = 40_000_000 // sc, 9_000_000 // sc, 1_000_000 // sc
    spec = {"u-i": (n_user, n_item, 300_000_000 // sc, 10), "i-s": (n_item, n_shop, 100_000_000 // sc, 10),
            "u-s": (n_user, n_shop, 100_000_000 // sc, 5)}
    t0 = time.time()
    graphs, whole = {}, {}
    seed_pool = None
    if sharded:
        import dist as gdist
    for i, (t, (ns, nd, ne, k)) in enumerate(spec.items()):
        src, dst, w = synth.rmat_edges_torch(1 << 26, ne, 20 + i, dev, weighted=True)
        src %= ns
        dst %= nd
        if t == "u-i":
            seed_pool = torch.unique(src)  # users that have at least one u-i edge
        if sharded:
            if args.verify:
                whole[t] = glx.Graph.from_edges(src, dst, w, device=dev.index)
            own = (src % world) == rank
            shard = glx.Graph.from_edges(src[own].contiguous(), dst[own].contiguous(), w[own].contiguous(),
                                         edge_ids=torch.nonzero(own).view(-1), device=dev.index)
            graphs[t] = gdist.ShardedStore(gdist.DeviceOps(), shard)
            del own
        else:
            graphs[t] = glx.Graph.from_edges(src, dst, w, device=dev.index)
        del src, dst, w
    x_item = glx.Features(synth.features_torch(n_item, D, 31, dev), device=dev.index)
    x_shop = glx.Features(synth.features_torch(n_shop, D, 32, dev), device=dev.index)
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log("c5 stores built in %.1fs: %s" % (time.time() - t0, {t: (g.graph if sharded else g).num_edges
                                                               for t, g in graphs.items()}))
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    n_steps = args.warmup + args.steps
    seeds = seed_pool[torch.randint(0, seed_pool.shape[0], (n_steps, B0), generator=gen, device=dev)]
    k1, k2, k3 = 10, 10, 5
    i64 = dict(dtype=torch.int64, device=dev)
    s1, e1 = torch.empty((B0, k1), **i64), torch.empty((B0, k1), **i64)
    s2, e2 = torch.empty((B0 * k1, k2), **i64), torch.empty((B0 * k1, k2), **i64)
    s3, e3 = torch.empty((B0, k3), **i64), torch.empty((B0, k3), **i64)
    seg = lambda n, f: (torch.arange(n, device=dev) // f).to(torch.int32)  # noqa: E731
    g2, g1, g3 = seg(B0 * k1 * k2, k2), seg(B0 * k1, k1), seg(B0 * k3, k3)
    f32 = dict(dtype=torch.float32, device=dev)
    o2 = (torch.empty((B0 * k1, D), **f32), torch.empty(B0 * k1, dtype=torch.int32, device=dev))
    o1 = (torch.empty((B0, D), **f32), torch.empty(B0, dtype=torch.int32, device=dev))
    o3 = (torch.empty((B0, D), **f32), torch.empty(B0, dtype=torch.int32, device=dev))

    last = {}

    def step(i, check=False):
        if sharded:
            a1, b1 = graphs["u-i"].sample("TopkSampler", seeds[i], k1)
            a2, b2 = graphs["i-s"].sample("TopkSampler", a1.view(-1), k2)
            a3, b3 = graphs["u-s"].sample("TopkSampler", seeds[i], k3)
        else:
            graphs["u-i"].sample("TopkSampler", seeds[i], k1, out=(s1, e1))
            graphs["i-s"].sample("TopkSampler", s1.view(-1), k2, out=(s2, e2))
            graphs["u-s"].sample("TopkSampler", seeds[i], k3, out=(s3, e3))
            a1, a2, a3 = s1, s2, s3
        last["a2"] = a2
        last["a1"] = a1
        # dense sampler responses imply their segments (segment i = the neighbours of request row i): no segment tensor
        x_shop.aggregate("SumAggregator", a2.view(-1), None, B0 * k1, out=o2)
        x_item.aggregate("SumAggregator", a1.view(-1), None, B0, out=o1)
        x_shop.aggregate("SumAggregator", a3.view(-1), None, B0, out=o3)
        if check:  # the partitioned result against unpartitioned copies of the three graphs
            w1, we1 = whole["u-i"].sample("TopkSampler", seeds[i], k1)
            w2, we2 = whole["i-s"].sample("TopkSampler", w1.view(-1), k2)
            w3, we3 = whole["u-s"].sample("TopkSampler", seeds[i], k3)
            return bool(torch.equal(a1, w1) and torch.equal(b1, we1) and torch.equal(a2, w2) and torch.equal(b2, we2)
                        and torch.equal(a3, w3) and torch.equal(b3, we3))
        return None

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    glx.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    glx.profile_enable(False)
    verified = None
    if sharded:
        t = torch.tensor([elapsed], device=(dev if args.backend == "nccl" else "cpu"), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.verify:
            flag = torch.tensor([1 if step(n_steps - 1, check=True) else 0], dtype=torch.int64,
                                device=(dev if args.backend == "nccl" else "cpu"))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified = bool(flag.item())
            log("verify: sharded == unpartitioned on every rank: %s" % verified)
    t_agg = glx.profile_collect(glx.KERNEL_AGGREGATE)
    t_smp = glx.profile_collect(glx.KERNEL_SAMPLE)
    slots = B0 * (k1 + k1 * k2 + k3)
    n2, sg2 = B0 * k1 * k2, B0 * k1
    n1 = B0 * k1
    ms2 = float(np.mean(t_agg[0::3]))  # i-s hop over the 1 M-row shop table: the longest launch of a step
    ms1 = float(np.mean(t_agg[1::3]))  # u-i hop over the 9 M-row item table: the roofline kernel (see below)
    oracle_check = None
    if extras["verify_oracle"]:
        # the last timed step's three sampler responses and three aggregates against the oracle, on row subsets cut from
        # the regenerated raw edge lists (tests/headline_check.py); outside the timed region
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from headline_check import check_aggregate, check_sample, _pick
            from oracle_bindings import Oracle
            orc, pick = Oracle(), np.random.default_rng(5)
            t0v = time.time()
            bad, sub_edges = [], 0
            for i, (t, (ns, nd, ne, k)) in enumerate(spec.items()):
                src, dst, w = synth.rmat_edges_torch(1 << 26, ne, 20 + i, dev, weighted=True)
                src %= ns
                dst %= nd
                req, nbr, eid = {"u-i": (seeds[n_steps - 1], s1, e1), "i-s": (s1.view(-1), s2, e2),
                                 "u-s": (seeds[n_steps - 1], s3, e3)}[t]
                ok, ne_sub = check_sample(orc, (src, dst, w), "TopkSampler", k, req, nbr, eid, 0, 0,
                                          _pick(req.shape[0], 4096, pick))
                sub_edges += ne_sub
                if not ok:
                    bad.append(t + " sample")
                del src, dst, w
            for name, n_rows, fseed, ids2d, (emb, cnt), want in (("i-s", n_shop, 32, s2, o2, 16384), ("u-i", n_item, 31, s1, o1, 2048),
                                                                 ("u-s", n_shop, 32, s3, o3, 2048)):
                X = synth.features_torch(n_rows, D, fseed, dev)
                if not check_aggregate(orc, lambda ids: X[ids], "SumAggregator", ids2d, emb, cnt, _pick(ids2d.shape[0], want, pick)):
                    bad.append(name + " aggregate")
                del X
            torch.cuda.empty_cache()
            oracle_check = {"ok": not bad, "mismatches": bad, "rows_per_edge_type": 4096, "segments": [16384, 2048, 2048],
                            "edges_in_subgraphs": sub_edges, "wall_s": time.time() - t0v}
        except Exception as ex:  # noqa: BLE001
            oracle_check = {"ok": None, "error": repr(ex)}
        log("c5 verify vs oracle: %s" % oracle_check)
    # c5's `roofline` is quoted on its DOMINANT launch: the i-s hop's reduce over the 1 M-row shop
    # table -- two thirds of the step.  It is not an HBM-bound launch: the request touches ~0.4 GB of distinct rows, ~75 % of
    # its row reads hit the CU's L1 (Topk's deterministic answers repeat; circular padding repeats rows inside a segment),
    # and the per-CU vector-memory pipeline (TA -> TCP -> TD) is busy ~95 % of the launch (profiles/r06/c5_is_reduce.md).
    # Its bound is that pipeline ("issue"), its ceiling the SAME launch with no cache miss at all -- ids uniform over
    # 2,048 rows, resident in every XCD's L2 -- timed in this run; frac = floor time / live time.  The u-i hop's launch
    # (ids over the 9.2 GB item table: HBM-bound, 9 % of the step) keeps its HBM roofline beside it as `roofline_u_i`.
    roof_ui = roofline_aggregate("SumAggregator", D, B0, n1, ms1, int(len(t_agg[1::3])), s1 if not sharded else last["a1"],
                                 "c5", B0, offline_ok=False)
    roof_ui["kernel"] = "glx_aggregate_grp_kernel (u-i hop SumAggregator over the 9 M-row item table, dim=%d)" % D
    bytes_is = n2 * (4 * D + 12) + sg2 * (4 * D + 4)
    step_ms = elapsed / args.steps * 1e3
    roof = {"kernel": "glx_aggregate_grp_kernel (i-s hop SumAggregator: %d ids over the 1 M-row / 1 GB shop table, dim=%d)" % (n2, D),
            "bound": "issue", "unit": "GB/s", "avg_launch_ms": ms2, "launches_timed": int(len(t_agg[0::3])),
            "share_of_step": ms2 / step_ms, "algorithmic_bytes_per_launch": bytes_is,
            "algorithmic_gbs": bytes_is / (ms2 * 1e-3) / 1e9, "algorithmic_over_peak": bytes_is / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "achieved": bytes_is / (ms2 * 1e-3) / 1e9, "cache_assisted": True,
            "bound_detail": "per-CU vector-memory pipeline (TA/TCP/TD): busy ~95 % of the launch, about three quarters of the row reads hit L1, "
                            "waves parked on memory 55 % of their cycles, 1 M-row table mostly on-die (rocprofv3 counter "
                            "passes of scripts/r06/c5_is_probe.py: profiles/r06/c5_is_reduce.md); not HBM (memory-side traffic "
                            "is a quarter of the algorithmic bytes) and not L2 bandwidth (half of 34.5 TB/s at the floor)"}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not sharded and os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get("c5_b%d" % B0, {})
        except Exception:  # noqa: BLE001
            rec = {}
        if rec.get("aggregate_hop2_bytes_per_launch"):
            roof["traffic"] = rec["aggregate_hop2_bytes_per_launch"]
            roof["traffic_source"] = "OFFLINE: profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per i-s launch); not of this run"
            roof["memory_side_gbs"] = rec["aggregate_hop2_bytes_per_launch"] / (ms2 * 1e-3) / 1e9
        for k in ("is_tcp_busy_frac", "is_l1_hit_rate", "is_l2_hit_rate", "is_wave_wait_frac"):
            if rec.get(k) is not None:
                roof[k.replace("is_", "offline_")] = rec[k]
        if rec.get("aggregate_item_bytes_per_launch"):
            roof_ui["traffic"] = rec["aggregate_item_bytes_per_launch"]
            roof_ui["traffic_source"] = "OFFLINE: profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per u-i launch); not of this run"
    roof_smp = None
    if not sharded and len(t_smp) >= 3:
        roof_smp = roofline_sampler("TopkSampler", k2, sg2, n2, float(np.mean(t_smp[1::3])), int(len(t_smp[1::3])), "c5", B0)
    if extras["roofline_probes"]:
        def timed(f_tab, ids_t, n_seg, out, reps=10):
            for _ in range(2):
                f_tab.aggregate("SumAggregator", ids_t, None, n_seg, out=out)
            torch.cuda.synchronize()
            glx.profile_enable(True)
            for _ in range(reps):
                f_tab.aggregate("SumAggregator", ids_t, None, n_seg, out=out)
            torch.cuda.synchronize()
            glx.profile_enable(False)
            return float(np.mean(glx.profile_collect(glx.KERNEL_AGGREGATE)))
        # the i-s launch's floor: the same kernel, same request shape, every row read an L2 hit (2,048 rows = 2 MB)
        fake = torch.randint(0, 2048, (n2,), generator=gen, device=dev)
        ms_floor = timed(x_shop, fake, sg2, o2)
        # ... and its HBM case: ids uniform over the whole 1 GB table (beyond the 256 MB Infinity Cache)
        fake = torch.randint(0, n_shop, (n2,), generator=gen, device=dev)
        ms_uni = timed(x_shop, fake, sg2, o2)
        roof["peak"] = bytes_is / (ms_floor * 1e-3) / 1e9
        roof["frac"] = ms_floor / ms2
        roof["l2_resident_floor"] = {"rows": "ids uniform over 2,048 rows (2 MB: resident in every XCD's L2)", "avg_launch_ms": ms_floor,
                                     "launches_timed": 10, "algorithmic_gbs": roof["peak"],
                                     "frac_of_l2_peak_34500": roof["peak"] / 34500.0}
        roof["uniform_over_table"] = {"rows": "ids uniform over all %d rows (1 GB)" % n_shop, "avg_launch_ms": ms_uni,
                                      "launches_timed": 10, "algorithmic_gbs": bytes_is / (ms_uni * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": bytes_is / (ms_uni * 1e-3) / 1e9 / HBM_PEAK_GBS}
        roof["achieved_basis"] = "algorithmic bytes of the i-s launch / its average live launch time (cache-assisted: not a bandwidth)"
        roof["frac_basis"] = ("L2-resident floor of this run: the same kernel on the same request shape with every row read an "
                              "L2 hit / the live launch -- how close the launch is to the rate its own instruction stream allows")
        # cache-free leg of the u-i launch: the same kernel on the same request shape with ids uniform over the 9 M item rows
        fake = torch.randint(0, n_item, (n1,), generator=gen, device=dev)
        ms = timed(x_item, fake, B0, o1)
        cf = roof_ui["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9
        roof_ui["cache_free"] = {"rows": "uniform over the %d item rows (%.1f GB)" % (n_item, n_item * D * 4 / 1e9),
                                 "avg_launch_ms": ms, "launches_timed": 10, "achieved": cf, "frac": cf / HBM_PEAK_GBS,
                                 "note": "a short launch (%d ids, %.2f GB): ramp-up and tail are a visible share of it" %
                                         (n1, roof_ui["algorithmic_bytes_per_launch"] / 1e9)}
        roof_ui["frac"] = cf / HBM_PEAK_GBS
        roof_ui["achieved"] = cf
        roof_ui["achieved_basis"] = "cache-free leg: algorithmic bytes (== memory-side traffic there) / its average launch time"
        roof_ui["frac_basis"] = "cache-free leg of this run: u-i hop shape, ids uniform over the 9.2 GB item table / 8 TB/s"
        del fake
    res = {
        "metric": "sampled-edges/sec + aggregated-vertices/sec (per-edge-type Topk + type-wise Sum per step)",
        "value": world * slots * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c5: BASELINE configs[4] shape on %s: user-item-shop graph, 3 weighted edge types (u-i 300M, i-s 100M, "
                               "u-s 100M edges over 40M / 9M / 1M nodes), per-edge-type TopkSampler (k = 10, 10, 5) + type-wise "
                               "SumAggregator, dim=256" % ("ONE GPU" if world == 1 else "%d GPUs" % world),
                   "seeds_per_step_per_gpu": B0,
                   "edge_types": {t: {"edges": v[2], "k": v[3]} for t, v in spec.items()}, "dim": D,
                   "parallelism": ("1 GPU" if not sharded else
                                   "3 edge types edge-cut llabs(src)%%%d + RCCL all-to-all per hop; item / shop "
                                   "features replicated" % world)},
        "phases": {"sampling_kernels_ms_per_step": float(np.sum(t_smp)) / args.steps,
                   "aggregation_kernels_ms_per_step": float(np.sum(t_agg)) / args.steps},
        "roofline": roof,
        "roofline_u_i": roof_ui,
        "cpu_baseline": None,
    }
    if roof_smp is not None:
        res["roofline_sampler"] = roof_smp
    if oracle_check is not None:
        res["verified_vs_oracle"] = oracle_check["ok"]
        res["oracle_check"] = oracle_check
    if verified is not None:
        res["verified_sharded_equals_unpartitioned"] = verified
    if sharded:
        res.update(comm_facts(graphs["u-i"].comm, args.share_device))
    if rank == 0:
        emit_result(result_out, res, args)
    if sharded:
        dist.destroy_process_group()
    if verified is False:
        sys.exit(3)


def roofline_aggregate(agg, D, n_segments, n_ids, avg_ms, launches, ids_last, workload, B0, sources=1, offline_ok=True):
    """Roofline entry of the hop-2 segmented reduce.  Everything named frac* is computed from THIS run:
      algorithmic bytes: n_ids * (4D + 12) + n_segments * (4D + 4) -- every row occurrence counted;
      compulsory bytes: the DISTINCT rows of the launch's ids once + ids + outputs -- what HBM must deliver even
        with perfect caches; achieved / frac use these (<= the real traffic <= peak * t, so frac <= 1);
      offline PMC traffic (profiles/pmc_traffic.json) is reported beside, labelled as not of this run."""
    bytes_alg = n_ids * (4 * D + 12) + n_segments * (4 * D + 4)
    distinct = int(torch.unique(ids_last.reshape(-1)).numel())
    bytes_comp = distinct * 4 * D + n_ids * 12 + n_segments * (4 * D + 4)
    t = avg_ms * 1e-3
    roof = {"kernel": "glx_aggregate_grp_kernel (hop-2 %s, dim=%d%s)" % (agg, D, ", 3 row sources" if sources == 3 else ""),
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": avg_ms, "launches_timed": launches,
            "algorithmic_bytes_per_launch": bytes_alg,
            # algorithmic bytes / the average duration of the launches inside the timed region.  NOT a
            # bandwidth: every row occurrence is counted, so re-reads of hub rows served by L2 / Infinity Cache are in
            # it -- the rate the kernel delivers to its consumer, which on a power-law request exceeds what HBM supplies
            "algorithmic_gbs": bytes_alg / t / 1e9,
            "algorithmic_over_peak": bytes_alg / t / 1e9 / HBM_PEAK_GBS, "cache_assisted": True,
            # `achieved` is always a rate of bytes that provably crossed the memory system, so achieved / peak = frac <= 1:
            # here the compulsory bytes of the live launch (a lower bound); the cache-free leg replaces it below
            "achieved": bytes_comp / t / 1e9,
            "achieved_basis": "compulsory bytes of the last timed launch / live average launch time",
            "distinct_rows_last_launch": distinct, "compulsory_bytes_per_launch": bytes_comp,
            "frac_compulsory": bytes_comp / t / 1e9 / HBM_PEAK_GBS,
            "note_compulsory": "lower bound for the timed launches: the distinct rows of the launch once + ids + outputs "
                               "(what HBM must deliver with perfect caches) / live launch time / 8 TB/s",
            # until the cache-free leg below has run, the only fraction this run can vouch for is the lower bound
            "frac": bytes_comp / t / 1e9 / HBM_PEAK_GBS,
            "frac_basis": "compulsory bytes of this run's last timed hop-2 launch / live average launch time / 8 TB/s (lower bound)",
            "traffic": None}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if offline_ok and os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get("%s_b%d" % (workload, B0), {})
        except Exception:  # noqa: BLE001
            rec = {}
        traffic = rec.get("aggregate_hop2_bytes_per_launch")
        if traffic:
            roof["traffic"] = traffic
            roof["traffic_source"] = ("OFFLINE: profiles/pmc_traffic.json, rocprofv3 --pmc passes of this command on another "
                                      "box (FETCH_SIZE x2 + WRITE_SIZE per launch); not measured in this run")
            roof["frac_traffic_offline"] = min(1.0, traffic / t / 1e9 / HBM_PEAK_GBS)
            # memory-side counter bytes / live launch time: what the fabric (HBM + Infinity Cache) moved per second
            roof["memory_side_gbs"] = traffic / t / 1e9
            # what HBM itself delivered lies between the two: Infinity-Cache hits are inside the memory-side count
            roof["hbm_bytes_bracket"] = [bytes_comp, traffic]
            roof["traffic_over_algorithmic"] = traffic / bytes_alg
        if rec.get("aggregate_hop2_l2_hit_rate") is not None:
            roof["l2_hit_rate"] = rec["aggregate_hop2_l2_hit_rate"]
            roof["l2_hit_rate_source"] = "OFFLINE: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) of the same launches (profiles/pmc_traffic.json)"
    return roof


def roofline_sampler(sampler, k, rows, slots, avg_ms, launches, workload, B0):
    """Roofline entry of the hop-2 sampler launch.  SURVEY 8(d): 32 B per output slot (8 col + 8 edge id read, 16
    written), EdgeWeight + 8 B per draw (alias prob + index), + 24 B per request row (src id + two row_ptr)."""
    per_slot = 40 if sampler in ("EdgeWeightSampler", "InDegreeSampler") else 32
    bytes_alg = slots * per_slot + rows * 24
    t = avg_ms * 1e-3
    r = {"kernel": "hop-2 %s launch (k=%d, %d request rows)" % (sampler, k, rows), "bound": "hbm", "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "avg_launch_ms": avg_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": bytes_alg,
         "achieved": bytes_alg / t / 1e9, "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS, "draws_per_s": slots / t, "traffic": None}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get("%s_b%d" % (workload, B0), {})
        except Exception:  # noqa: BLE001
            rec = {}
        if rec.get("sample_hop2_bytes_per_launch"):
            r["traffic"] = rec["sample_hop2_bytes_per_launch"]
            r["traffic_source"] = "OFFLINE: profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per launch); not of this run"
            r["traffic_over_algorithmic_offline"] = rec["sample_hop2_bytes_per_launch"] / bytes_alg
    return r


def verify_vs_oracle(args, wl, dev, seeds_last, out, call_counters):
    """--verify-oracle: the outputs of the last TIMED step against the oracle, on row subsets cut from the raw edge
    list (tests/headline_check.py).  Outside the timed region; the oracle is the checker, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from headline_check import check_step
    V, E, sampler, fan, agg, D, gseed, _ = wl
    weighted = sampler in ("EdgeWeightSampler", "TopkSampler")
    t0 = time.time()
    src, dst, weight = synth.rmat_edges_torch(V, E, gseed, dev, weighted=weighted, scramble=not args.no_scramble)
    X = synth.features_torch(V, D, gseed + 1, dev)
    hubs = torch.topk(torch.bincount(src, minlength=V), 100).indices.cpu().numpy()
    r = check_step((src, dst, weight), lambda ids: X[ids], sampler, fan, agg, seeds_last, out, seed=42,
                   call_counters=call_counters, rows_hop1=4096, rows_hop2=8192, segments=16384, hub_ids=hubs)
    del src, dst, weight, X
    torch.cuda.empty_cache()
    r["wall_s"] = time.time() - t0
    r["what"] = ("last timed step: sampled (neighbour, edge id) rows and aggregated segments re-computed by oracle/glx_oracle.c "
                 "on sub-graphs cut from the raw edge list; bit-for-bit equality")
    log("verify vs oracle: %s" % r)
    return r


def other_configs(args):
    """BASELINE configs[1] (c2), configs[4] (c5) and configs[3] (c4: 111 M nodes / 1.6 B edges, here on ONE GPU) on this
    GPU, each in a process of its own (the headline's store is released first), same --steps / --warmup: driver-timed
    numbers for the configs that are not the headline."""
    import subprocess
    out = {}
    for name in [x for x in args.other_configs.split(",") if x.strip()]:
        wl_name, _, variant = name.partition("-")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", wl_name,
               "--seed-dist", "degree" if variant == "degree-seeds" else "uniform", "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--cpu-baseline", "off", "--host-boundary", "off", "--edge-cut-probe", "off",
               "--small-batches", "off", "--other-configs", ""]
        t0 = time.time()
        r = None
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out[name] = {"workload": rec["config"]["workload"], "seeds": rec["config"].get("seeds"), "ms_per_step": rec["ms_per_step"], "value": rec["value"],
                         "unit": rec["unit"], "steps": rec["steps"], "phases": rec.get("phases"), "roofline": rec.get("roofline"),
                         "roofline_u_i": rec.get("roofline_u_i"), "roofline_sampler": rec.get("roofline_sampler"), "verified_vs_oracle": rec.get("verified_vs_oracle"),
                         "wall_s": time.time() - t0}
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line
            out[name] = {"error": repr(ex), "returncode": getattr(r, "returncode", None),
                         "stderr_tail": (r.stderr[-1500:] if r is not None and r.stderr else None)}
        log("other config %s: %s" % (name, {k: out[name].get(k) for k in ("ms_per_step", "value", "error", "stderr_tail")}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS) + ["c5"])
    ap.add_argument("--batch", type=int, default=65536, help="seed vertices per step per GPU (B0)")
    ap.add_argument("--features", default="auto", choices=["auto", "replicated", "sharded"],
                    help="N>1: which feature placement `value` reports.  auto / sharded = north_star's: the table "
                         "stays edge-cut, aggregation fetches halo rows per request (auto also times the "
                         "replicated placement and reports it beside); replicated = the whole table on every "
                         "GPU (one load-time all-gather) is the headline")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="overlap step i+1's sampling (+ xGMI exchange) with step i's aggregation on two "
                         "HIP streams; auto = on for N>1")
    ap.add_argument("--no-scramble", action="store_true",
                    help="keep raw RMAT vertex ids (bit-skewed: 44%% of the edges land on shard 0 of 8 under "
                         "llabs(id)%%P, and hub rows alias onto few HBM channels) instead of the Graph500-style "
                         "random relabeling")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="auto = nccl (RCCL over xGMI), or gloo with --share-device; gloo = test rig only: collectives "
                         "staged through host memory")
    ap.add_argument("--detail-out", default=None,
                    help="where rank 0 writes the full record (default: bench_detail.json beside this file); the stdout "
                         "line is the compact one (<= 4 KB)")
    ap.add_argument("--share-device", action="store_true",
                    help="test rig: every rank uses cuda:0 (several ranks on a 1-GPU box, with --backend gloo)")
    ap.add_argument("--verify", action="store_true",
                    help="after timing, recompute one step on an unpartitioned copy of the graph held by this "
                         "rank and require bit-identical outputs from the sharded path")
    ap.add_argument("--verify-sharded", default="auto", choices=["auto", "off"],
                    help="N>1: auto = --verify whenever an unpartitioned copy of the graph and the feature table fits "
                         "beside the shards (<= 64 GiB): every partitioned leg of the line is then checked, bit for bit, "
                         "against the unpartitioned operators on every rank")
    ap.add_argument("--hot-fraction", type=float, default=0.5,
                    help="N>1: every GPU keeps a replica of this fraction of the feature rows (the top vertices by "
                         "global in-degree); the rest is fetched per request (halo exchange of the cold tail).  0.5 since "
                         "round 6 (5 GB of rows + 6 GB of adjacency per GPU on the headline graph, 4 %% of the HBM): one "
                         "rank's step at P = 8 costs 3.12 / 2.72 / 2.50 / 2.51 ms of kernels at 0.10 / 0.25 / 0.50 / 1.0 "
                         "and moves 8x fewer halo rows at 0.5 than at 0.25 (profiles/r06/p8_sym_hot_sweep.txt)")
    ap.add_argument("--host-boundary", default="on", choices=["on", "off"],
                    help="N=1: also run graph-learn_amd/lib/host_path_bench (requests through the C++ operator API "
                         "with host buffers) and report its PCIe-inclusive rate under \"host_boundary\"")
    ap.add_argument("--host-boundary-threads", type=int, default=32)
    ap.add_argument("--small-batches", default="on", choices=["on", "off"],
                    help="N=1 at the default batch: also time B0 = 1024 and 8192 on the same store (\"small_batches\")")
    ap.add_argument("--edge-cut-probe", default="on", choices=["on", "off"],
                    help="N=1, C3: after the headline, time the same steps once more through the multi-GPU machinery with "
                         "ONE rank over RCCL (partition, exchanges, resolve, halo slots, 3-source reduce: the generic "
                         "path, GLX_DIST_NO_SHORTCUT=1), verified against the plain operators, and report it under "
                         "\"edge_cut_world1\" -- what the edge-cut path costs before any link time")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="N=1: replay the step as one captured hipGraph (glx_plan) instead of 4 kernel launches; "
                         "auto = on for launch-bound batches (B0 <= 8192)")
    ap.add_argument("--graph-streams", type=int, default=3, help="--graph: plans / streams the steps alternate over")
    ap.add_argument("--watchdog", type=float, default=300.0,
                    help="N>1: seconds the timed legs may take before rank 0 prints a result without a value and all "
                         "ranks exit (a hung collective must not hang the node)")
    ap.add_argument("--setup-watchdog", type=float, default=900.0,
                    help="N>1: seconds everything before the timed legs may take (process group, communicators, "
                         "store build, replica fetch)")
    ap.add_argument("--stage-priority", default="normal", choices=["high", "normal"],
                    help="pipelined legs: HIP stream priority of the sampling / halo-prefetch stages")
    ap.add_argument("--graph-replica", default="auto", choices=["auto", "on", "off"],
                    help="N>1: also replicate the hot vertices' adjacency rows (their sampling requests stay local).  auto = "
                         "when the replica fits the free HBM, and config.workload says which it was; on = refuse to run when "
                         "it does not fit")
    ap.add_argument("--pure-leg", default="on", choices=["on", "off"],
                    help="N>1: after the placements with replicas, time the same steps once more with NO replica at all "
                         "(hot fraction 0, no graph replica): every remote request row and every remote feature row "
                         "crosses the links -- north_star's exchange itself; reported as placements.edge_cut_pure")
    ap.add_argument("--speculate", default="on", choices=["on", "off"],
                    help="N>1: time the partitioned placements a second time with a speculation ledger (glx_dist_ledger): "
                         "sampling requests travel in fixed-capacity messages without a count exchange, the aggregation's "
                         "count exchange confirms them -- one blocking host wait per step instead of three; reported as "
                         "placements.*_speculated, verified like the others")
    ap.add_argument("--design-r", default="on", choices=["on", "off"],
                    help="N>1, with --pure-leg on: also time the pure placement with the reference's partial-reduce-and-stitch "
                         "aggregation (glx_dist_aggregate_partial) instead of the halo-row exchange; placements.edge_cut_pure_design_r")
    ap.add_argument("--graph-hot-fraction", type=float, default=None,
                    help="N>1: fraction of the vertices (hottest first) whose adjacency rows are replicated; "
                         "default: --hot-fraction")
    ap.add_argument("--hot-by", default="indegree", choices=["access", "indegree"],
                    help="N>1: how the replicated rows are chosen: by access count over a few profiling requests, or by "
                         "global in-degree (glx_dist_hot_ids: needs no request profile)")
    ap.add_argument("--hot-profile-steps", type=int, default=4)
    ap.add_argument("--roofline-probes", default="on", choices=["on", "off"],
                    help="N=1: also time the aggregation kernel on cache-free (uniform) rows and a device copy")
    ap.add_argument("--verify-oracle", default="on", choices=["on", "off"],
                    help="N=1: after timing, re-check the last timed step's outputs on row subsets against the oracle "
                         "(tests/headline_check.py) and emit \"verified_vs_oracle\"")
    ap.add_argument("--seed-dist", default="uniform", choices=["uniform", "degree"],
                    help="seed vertices of a batch: uniform over the vertices that have out-edges, or degree-biased (the source "
                         "of a uniformly drawn edge: hubs are asked for in proportion to their out-degree) -- SURVEY 8(d)'s two "
                         "request mixes")
    ap.add_argument("--other-configs", default="c2,c5,c4,c3-degree-seeds",
                    help="N=1, headline workload at the default batch: also time these workloads (comma separated), each in a "
                         "process of its own, and report them under \"other_configs\"")
    ap.add_argument("--request-shape-legs", default="on", choices=["on", "off"],
                    help="N=1: time the same steps twice more -- with an explicit segment_ids tensor per aggregation (the "
                         "reference's AggregatingRequest always carries one) and with seeds uniform over ALL vertices "
                         "(SURVEY 8(d)) -- and make the slowest of the three the headline when it is more than 2%% slower")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded (RCCL) code path even with one process (testing)")
    ap.add_argument("--c5-scale", type=int, default=1, help="divide the c5 node / edge counts (test rig)")
    ap.add_argument("--cpu-baseline", default="on", choices=["on", "off"])
    ap.add_argument("--cpu-time-budget", type=float, default=3.0,
                    help="s of sampling and s of aggregation timed per (storage mode, thread count) leg of the CPU baseline")
    ap.add_argument("--cpu-seeds-per-request", type=int, default=128)
    ap.add_argument("--cpu-storage-modes", default="3,2",
                    help="reference StorageModes the CPU baseline runs: 2 = its default (vector-of-vectors adjacency, per-node "
                         "attribute objects), 3 = compressed (CSR + flat attributes); one worker process each")
    ap.add_argument("--cpu-thread-sweep", default="1,32,0",
                    help="request threads of the CPU baseline, comma separated (0 = nproc; 32 = InterThreadNum's default)")
    ap.add_argument("--cpu-edge-limit", type=int, default=0, help="CPU baseline: build only this prefix of the edge stream (0 = all)")
    ap.add_argument("--cpu-wall-limit", type=float, default=420.0, help="s a CPU baseline worker may take before it is abandoned")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-dir", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.backend == "auto":
        args.backend = "gloo" if args.share_device else "nccl"
    if args.cpu_worker:
        cpu_worker(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    # Contract: rank 0 prints exactly ONE JSON line on stdout.  Libraries (RCCL's
    # version banner, rocm warnings) also write to fd 1, so keep a private copy of
    # the real stdout for the result and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if torch.cuda.device_count() <= local_rank:  # a launcher that started more ranks than the node has GPUs
        raise SystemExit("rank %d: local rank %d has no GPU of its own (%d visible); one process per GPU -- --share-device "
                         "is the one-GPU test rig" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    setup_dog = None
    if world > 1:
        # a rank that never arrives (or a communicator that never forms) must not hang the node: the timed legs
        # have their own watchdog, this one covers everything before them
        import threading

        def setup_give_up():
            if rank == 0:
                emit_result(result_out, {
                    "metric": "sampled-edges/sec + aggregated-vertices/sec", "value": None, "unit": "edges/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic", "config": {"workload": args.workload},
                    "error": "setup watchdog: communicator / store set-up did not finish in %.0f s" % args.setup_watchdog,
                }, args)
            os._exit(5)
        setup_dog = threading.Timer(args.setup_watchdog, setup_give_up)
        setup_dog.daemon = True
        setup_dog.start()
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.workload == "c5":
        if setup_dog is not None:
            setup_dog.cancel()
        bench_c5(args, dev, result_out, world=world, rank=rank, sharded=sharded)
        return
    wl = WORKLOADS[args.workload]
    V, E, sampler, (k1, k2), agg, D, gseed, desc = wl
    B0 = args.batch
    t0 = time.time()
    weighted = sampler in ("EdgeWeightSampler", "TopkSampler")
    src, dst, weight = synth.rmat_edges_torch(V, E, gseed, dev, weighted=weighted, scramble=not args.no_scramble)
    torch.cuda.synchronize()
    log("edge list generated in %.1fs" % (time.time() - t0))

    # Seed vertices = vertices that have out-edges (RMAT leaves about half of the ids
    # without any; a training set is made of vertices that have neighbours).  Frontier
    # vertices reached by sampling may still have no out-edges: those rows are
    # default-filled exactly as the reference does (random_sampler.cc:58-59).
    seed_pool = torch.unique(src)
    if args.seed_dist == "degree":
        seed_pool = src.clone()  # one entry per edge: a uniform draw from it picks a vertex in proportion to its out-degree

    # The CPU baseline runs AFTER the timed GPU region (an idle GPU clocks down during
    # 1-2 minutes of host work); keep host copies of the edge list for it.
    extras = n1_extras(args, world, sharded)
    host_edges = None
    if extras["cpu_baseline"]:
        host_edges = (src.cpu(), dst.cpu(), weight.cpu() if weight is not None else None,
                      seed_pool.cpu().numpy())

    # Storage build on the device (glx_graph_build: radix sorts + RLE + scan + alias
    # tables + id map); rows end up weight-descending like the reference's Build().
    t1 = time.time()
    st_smp = st_agg = replica = whole = None
    hot = None
    if not sharded:
        # the graph first, its edge list released, THEN the feature matrix: C4's build (1.6 B edges: sort temporaries) and
        # its 57 GB of features + their staging copy would otherwise peak together
        graph = glx.Graph.from_edges(src, dst, weight, device=local_rank)
        del src, dst, weight
        src = dst = weight = None
        torch.cuda.empty_cache()
        X = synth.features_torch(V, D, gseed + 1, dev)
        feats = glx.Features(X, device=local_rank)
        placement = "1 GPU"
    else:
        X = synth.features_torch(V, D, gseed + 1, dev)
        import dist as gdist
        if not args.verify and args.verify_sharded == "auto" and V * D * 4 + E * 56 <= 64 * (1 << 30):
            args.verify = True
        if args.verify:
            whole = (glx.Graph.from_edges(src, dst, weight, device=local_rank), glx.Features(X, device=local_rank))
        own = (src % world) == rank  # edge-cut: out-edges of v live on shard llabs(v) % P
        eids = torch.nonzero(own).view(-1)
        graph = glx.Graph.from_edges(src[own].contiguous(), dst[own].contiguous(),
                                     weight[own].contiguous() if weight is not None else None,
                                     edge_ids=eids, device=local_rank)
        del own, eids
        x_shard = X[rank::world].contiguous()
        ids = torch.arange(rank, V, world, dtype=torch.int64, device=dev)
        feats = glx.Features(x_shard, ids=ids, device=local_rank)  # this rank's rows of the edge-cut table
        del ids
        # one communicator per stream: sampling (+ its exchanges) of step i+1 overlaps aggregation of step i
        comm_s = gdist.comm_for_group(None, local_rank)
        comm_a = gdist.comm_for_group(None, local_rank)  # the halo prefetch stage's collectives
        st_smp = glx.DistStore(comm_s, graph=graph)
        st_agg = glx.DistStore(comm_a, features=feats)
        # hot-row replica: the top vertices by GLOBAL in-degree (computed from the shards), fetched once
        n_hot = int(V * args.hot_fraction)
        t_hot = time.time()
        if n_hot <= 0:
            hot = np.empty(0, np.int64)
        elif args.hot_by == "indegree":
            hot = st_smp.hot_ids(n_hot)  # no request profile needed: global in-degree, computed from the shards
        else:
            # rank the rows by how often requests actually touch them: a few profiling requests through the
            # edge-cut sampler, access counts summed over the ranks (a weighted sampler does not visit vertices
            # in proportion to their in-degree: the in-degree top 10 % serve 87.5 % of C3's hop-2 accesses,
            # the access-count top 10 % serve ~95 %)
            acc = torch.zeros(V, dtype=torch.int32, device=dev)
            pgen = torch.Generator(device=dev)
            pgen.manual_seed(77 + rank)
            for j in range(args.hot_profile_steps):
                ps = seed_pool[torch.randint(0, seed_pool.shape[0], (B0,), generator=pgen, device=dev)]
                p1, _ = st_smp.sample(sampler, ps, k1, seed=4242, call_counter=2 * j)
                p2, _ = st_smp.sample(sampler, p1.view(-1), k2, seed=4242, call_counter=2 * j + 1)
                acc += torch.bincount(p2.view(-1).clamp(0, V - 1), minlength=V).to(torch.int32)
                acc += torch.bincount(p1.view(-1).clamp(0, V - 1), minlength=V).to(torch.int32)
            if world > 1:
                if args.backend == "nccl":
                    dist.all_reduce(acc)
                else:
                    h = acc.cpu()
                    dist.all_reduce(h)
                    acc = h.to(dev)
            hot = torch.topk(acc, n_hot).indices.to(torch.int64).cpu().numpy()
            del acc
        st_agg.set_cache(hot)
        torch.cuda.synchronize()
        log("hot-row replica (%s): %d rows (%.2f GB per GPU) selected + fetched in %.1fs"
            % (args.hot_by, hot.shape[0], hot.shape[0] * D * 4 / 1e9, time.time() - t_hot))
        graph_replica = None
        # room: the replica can hold nearly every edge (RMAT hubs own most of them) at ~56 B per edge, plus the build's
        # temporaries; skipped (with a note) when that is not there
        room = torch.cuda.mem_get_info(dev)[0] > 3 * E * 56
        if args.graph_replica == "on" and hot.shape[0] > 0 and not room:
            raise SystemExit("--graph-replica on: not enough free HBM for a replica of up to %d edges (use auto / off)" % E)
        if args.graph_replica == "auto" and hot.shape[0] > 0 and not room:
            log("graph replica: off (auto: not enough free HBM for a replica of up to %d edges)" % E)
        if args.graph_replica in ("on", "auto") and hot.shape[0] > 0 and room:
            # the same vertices' adjacency rows on every GPU: hop-2 request rows are hop-1 samples, i.e. mostly hubs,
            # and those are then sampled here instead of travelling to their owner and back.  Built from the shards:
            # every owner cuts its hot vertices' rows (its edge ids, its row order) and the pieces are all-gathered
            # once, at load time (glx_dist_build_graph_replica).
            t_rep = time.time()
            g_hot = hot if args.graph_hot_fraction is None else hot[:int(V * args.graph_hot_fraction)]
            graph_replica = st_smp.build_graph_replica(g_hot)
            torch.cuda.synchronize()
            log("graph replica: the out-edges of the hottest %d vertices, %d of %d edges, built in %.1fs"
                % (g_hot.shape[0], graph_replica.num_edges, E, time.time() - t_rep))
        if args.features != "sharded" and V * D * 4 <= 96 * (1 << 30):
            # the other end of the placement space: the whole table on every GPU (one load-time all-gather)
            full = gdist.replicate_features(x_shard, V) if world > 1 else X
            replica = glx.Features(full, device=local_rank)
            del full
        del x_shard
        placement = ("graph + features edge-cut llabs(v)%%%d%s; per hop: RCCL send/recv exchange of the request rows the "
                     "replica does not hold; "
                     "aggregation: own shard + replica of the top %.0f%% rows by %s + per-request halo exchange "
                     "of the deduplicated cold tail, prefetched beside the previous step's reduce (glx_dist_*)"
                     % (world, " + adjacency rows of the hot vertices on every GPU" if graph_replica is not None else "",
                        100 * args.hot_fraction, "access count" if args.hot_by == "access" else "in-degree"))
    del src, dst, weight
    X = None
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log("device store built in %.1fs (%d rows, %d edges on this GPU)" % (time.time() - t1, graph.num_rows, graph.num_edges))

    # synthetic request stream: uniform seeds, a fresh batch per step
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    n_steps = args.warmup + args.steps
    seeds = seed_pool[torch.randint(0, seed_pool.shape[0], (n_steps, B0), generator=gen, device=dev)]
    del seed_pool
    cur = {"seeds": seeds}  # the request stream the steps read: the request-shape legs swap it
    n1, n2 = B0 * k1, B0 * k1 * k2
    emb2 = torch.empty((n1, D), dtype=torch.float32, device=dev)
    cnt2 = torch.empty((n1,), dtype=torch.int32, device=dev)
    emb1 = torch.empty((B0, D), dtype=torch.float32, device=dev)
    cnt1 = torch.empty((B0,), dtype=torch.int32, device=dev)

    # Two-stage software pipeline over two HIP streams: the sampling stage of step
    # i+1 (id exchange over xGMI + gather kernels) overlaps the aggregation stage
    # of step i (the HBM-bound segmented reduce).  Every step's work still completes
    # inside the timed region; buffers are double-buffered and the sampler may run
    # at most one step ahead.
    pipelined = args.pipeline == "on" or (args.pipeline == "auto" and sharded)
    bufs = []
    both_ids = []  # per buffer set: the hop-2 neighbours followed by the hop-1 neighbours, one allocation
    for _ in range((3 if sharded else 2) if pipelined else 1):
        # hop-2 and hop-1 neighbour ids back to back: the partitioned aggregation resolves and fetches both with ONE
        # glx_dist_aggregate_begin (one count exchange, one deduplicated halo fetch) and reduces them as two ranges
        both = torch.empty(n2 + n1, dtype=torch.int64, device=dev)
        b2 = both[:n2].view(n1, k2)
        b1 = both[n2:].view(B0, k1)
        both_ids.append(both)
        bufs.append((b1, torch.empty_like(b1), b2, torch.empty_like(b2)))

    def do_sample(i):
        cc = 4 * i
        nb1, ed1, nb2, ed2 = bufs[i % len(bufs)]
        if st_smp is None:
            graph.sample(sampler, cur["seeds"][i], k1, seed=42, call_counter=cc, out=(nb1, ed1))
            graph.sample(sampler, nb1.view(-1), k2, seed=42, call_counter=cc + 1, out=(nb2, ed2))
        else:
            st_smp.sample(sampler, cur["seeds"][i], k1, seed=42, call_counter=cc, out=(nb1, ed1))
            st_smp.sample(sampler, nb1.view(-1), k2, seed=42, call_counter=cc + 1, out=(nb2, ed2))
        return nb1, nb2

    def halo_begin(a, b, i):
        # the collective half of the two aggregations: ids are resolved (replica / own shard / halo) and the halo
        # rows fetched from their owners -- on its own stream, beside the previous step's reduce and the next
        # step's sampling.  ONE request for both id sets (they sit back to back: both_ids)
        st_agg.aggregate_begin(i % 3, both_ids[i % len(both_ids)])

    # A dense sampler response implies its segments (segment i = the neighbours of request row i):
    # segment_ids = None skips the segment bookkeeping kernels and the read of a segment tensor.
    def agg_local(table, seg=(None, None)):
        # seg = (hop-2, hop-1) int32 segment_ids tensors: the reference's request shape (AggregatingRequest::Set(node_ids,
        # segment_ids, ...), include/aggregating_request.h:34-37), which runs the segment bookkeeping kernels
        # (glx_seg_valid / glx_seg_start) and reads 4 B per input vertex on top
        def run(a, b, i):
            table.aggregate(agg, b.view(-1), seg[0], n1, out=(emb2, cnt2))
            table.aggregate(agg, a.view(-1), seg[1], B0, out=(emb1, cnt1))
        return run

    def agg_halo(a, b, i):
        # the local half: the segmented reduce over own shard + hot-row replica + the halo rows begun above
        st_agg.aggregate_end_range(i % 3, 0, n2, agg, None, n1, out=(emb2, cnt2))
        st_agg.aggregate_end_range(i % 3, n2, n1, agg, None, B0, out=(emb1, cnt1), release=True)

    if pipelined:
        # the sampling and halo-prefetch stages are chains of short kernels beside the long reduce; serving their queues
        # first (high-priority HIP streams) shortens them (a one-workgroup scan: 0.7 ms -> 45 us) but costs throughput:
        # 3.46 / 3.97 vs 2.93 / 2.80 ms per step (sharded / replicated, world size 1) -- measured, off by default
        prio = -1 if args.stage_priority == "high" else 0
        s_smp, s_agg = torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)
        s_pre = torch.cuda.Stream(device=dev, priority=prio)  # sharded leg: the halo prefetch stage

    def barrier():
        # Drain the local queue first: an RCCL barrier issued while the GPU still has queued work
        # was seen to take 100+ ms sporadically (world-size-1 probe), which would be charged to
        # the timed region; with an idle GPU it costs ~0.06 ms.
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_leg(do_aggregate, steps_from, steps_to, warm, serial=False):
        """Times steps [steps_from, steps_to) after `warm` untimed ones; -> (seconds, t_agg, t_smp).  serial: every
        kernel and every collective of a step on ONE stream, in program order."""
        agg_done = []

        def step(i):
            if serial or not pipelined:
                a, b = do_sample(i)
                do_aggregate(a, b, i)
                return
            with torch.cuda.stream(s_smp):
                if len(agg_done) >= 2:
                    s_smp.wait_event(agg_done[-2])  # the buffers of step i-2 are free again
                a, b = do_sample(i)
                sampled = torch.cuda.Event()
                sampled.record(s_smp)
            with torch.cuda.stream(s_agg):
                s_agg.wait_event(sampled)
                do_aggregate(a, b, i)
                done = torch.cuda.Event()
                done.record(s_agg)
                agg_done.append(done)
        for i in range(warm):
            step(i)
        barrier()
        glx.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(steps_from, steps_to):
            step(i)
        barrier()
        dt = time.perf_counter() - t0
        glx.profile_enable(False)
        t_a = glx.profile_collect(glx.KERNEL_AGGREGATE)
        t_s = glx.profile_collect(glx.KERNEL_SAMPLE)
        if world > 1:
            t = torch.tensor([dt], device=ctl, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_a, t_s

    def timed_leg_halo(steps_from, steps_to, warm):
        """The edge-cut placement as a three-stage software pipeline over three streams (one communicator per
        collective stage): S = 2-hop sampling of step i+1 (request rows exchanged over the links), P = halo
        prefetch of step i (resolve + dedup + ids out + owners gather + rows back), R = the segmented reduce of
        step i.  Every step's work completes inside the timed region."""
        if not pipelined:
            def flat(a, b, i):
                halo_begin(a, b, i)
                agg_halo(a, b, i)
            return timed_leg(flat, steps_from, steps_to, warm)
        ev_r = {}
        pending = {}

        def stage_s(i):
            with torch.cuda.stream(s_smp):
                if i - 3 in ev_r:
                    s_smp.wait_event(ev_r[i - 3])  # buffers and slots of step i-3 are free again
                a, b = do_sample(i)
                e = torch.cuda.Event()
                e.record(s_smp)
            pending[i] = (a, b, e)

        def stage_pr(i):
            a, b, e = pending.pop(i)
            with torch.cuda.stream(s_pre):
                s_pre.wait_event(e)
                halo_begin(a, b, i)
                p = torch.cuda.Event()
                p.record(s_pre)
            with torch.cuda.stream(s_agg):
                s_agg.wait_event(p)
                agg_halo(a, b, i)
                r = torch.cuda.Event()
                r.record(s_agg)
            ev_r[i] = r
            ev_r.pop(i - 6, None)

        def run(first, last):
            if first >= last:
                return
            stage_s(first)
            for i in range(first, last):
                if i + 1 < last:
                    stage_s(i + 1)  # issued BEFORE this step's prefetch: the host then waits in begin(i)
                stage_pr(i)
        run(0, warm)
        barrier()
        ev_r.clear()
        sync0 = [st.stats() for st in (st_smp, st_agg)]
        glx.profile_enable(True)
        t0 = time.perf_counter()
        run(steps_from, steps_to)
        barrier()
        dt = time.perf_counter() - t0
        glx.profile_enable(False)
        sync1 = [st.stats() for st in (st_smp, st_agg)]
        nst = max(steps_to - steps_from, 1)
        # count exchanges of the timed steps: each blocks the issuing host thread until every rank's counts are in -- and,
        # before that, until the stream has run the kernels queued ahead of the exchange, which in this GPU-bound
        # pipeline is most of the wait (the other stages' streams keep the GPU busy meanwhile): time the HOST spends
        # blocked, not time lost
        last_host_syncs.update(
            count_exchanges_per_step=sum(b["host_syncs"] - a["host_syncs"] for a, b in zip(sync0, sync1)) / nst,
            host_blocked_in_count_exchanges_ms_per_step=sum(b["host_stall_us"] - a["host_stall_us"]
                                                            for a, b in zip(sync0, sync1)) / nst / 1e3)
        t_a = glx.profile_collect(glx.KERNEL_AGGREGATE)
        t_s = glx.profile_collect(glx.KERNEL_SAMPLE)
        if world > 1:
            t = torch.tensor([dt], device=ctl, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_a, t_s

    verified_legs = {}
    last_host_syncs = {}

    def verify_sharded():
        """One step through the partitioned stores as they are NOW against unpartitioned copies held by this rank."""
        i = n_steps - 1
        a, b = do_sample(i)
        halo_begin(a, b, i)
        agg_halo(a, b, i)
        wa, wae = whole[0].sample(sampler, seeds[i], k1, seed=42, call_counter=4 * i)
        wb, wbe = whole[0].sample(sampler, wa.view(-1), k2, seed=42, call_counter=4 * i + 1)
        we2, wc2 = whole[1].aggregate(agg, wb.view(-1), None, n1)
        we1, wc1 = whole[1].aggregate(agg, wa.view(-1), None, B0)
        torch.cuda.synchronize()
        nb1, ed1, nb2, ed2 = bufs[i % len(bufs)]
        ok = bool(torch.equal(nb1, wa) and torch.equal(ed1, wae) and torch.equal(nb2, wb) and torch.equal(ed2, wbe)
                  and torch.equal(cnt2, wc2) and torch.equal(emb2.view(torch.int32), we2.view(torch.int32))
                  and torch.equal(cnt1, wc1) and torch.equal(emb1.view(torch.int32), we1.view(torch.int32)))
        if replica is not None:
            re2, _ = replica.aggregate(agg, wb.view(-1), None, n1)
            ok = ok and bool(torch.equal(re2.view(torch.int32), we2.view(torch.int32)))
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        return ok

    edges_per_step = n1 + n2  # response slots, padding included
    kernel_steps = args.steps  # steps the per-kernel timers covered
    ctl = dev if args.backend == "nccl" else torch.device("cpu")  # control-plane tensors (gloo rig: host)
    legs = {}
    request_legs = {}
    use_graph = not sharded and (args.graph == "on" or (args.graph == "auto" and B0 <= 8192))
    if use_graph:
        # launch-bound batch sizes: the whole step (2 sample + 2 aggregate kernels) is ONE hipGraph launch
        # (glx_plan); seeds and call counter enter through the graph's stage node
        # two plans on two streams, steps alternate: a step's small kernels (hop 1 is 100 workgroups) leave most
        # of the GPU idle, and the gaps between the nodes of one graph are filled by the other stream's step
        plans = [glx.Plan([graph, graph], sampler, [k1, k2], B0, features=[feats, feats], agg=agg, seed=42)
                 for _ in range(args.graph_streams)]
        streams = [torch.cuda.Stream(device=dev) for _ in plans]

        def graph_step(i):
            with torch.cuda.stream(streams[i % len(plans)]):
                plans[i % len(plans)].run(seeds[i], call_counter=4 * i)
        for i in range(args.warmup):
            graph_step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            graph_step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        # kernel durations for the roofline line: a few steps issued kernel by kernel, outside the timed region
        kernel_steps = min(args.steps, 10)
        _, t_agg, t_smp = timed_leg(agg_local(feats), 0, kernel_steps, 2)
        headline = "single GPU, one hipGraph launch per step"
        placement = "1 GPU; step = one hipGraph launch (glx_plan), %d plan(s) alternating on as many streams" % len(plans)
    elif not sharded:
        request_legs = {}
        leg_runs = {}
        if extras["request_shape_legs"]:
            # the two request shapes VERDICT r04 asks for beside the headline's, each the same args.steps steps:
            #  (b) every aggregation carries an explicit segment_ids tensor, as the reference's request always does
            #  (c) seeds uniform over ALL V vertices, half of which have no out-edges in an RMAT graph
            seg = (torch.arange(n2, dtype=torch.int32, device=dev) // k2, torch.arange(n1, dtype=torch.int32, device=dev) // k1)
            gen_v = torch.Generator(device=dev)
            gen_v.manual_seed(2000 + rank)
            seeds_v = torch.randint(0, V, (n_steps, B0), generator=gen_v, device=dev)
            cur["seeds"] = seeds_v
            leg_runs["seeds_uniform_over_V"] = timed_leg(agg_local(feats), args.warmup, n_steps, args.warmup)
            cur["seeds"] = seeds
            leg_runs["with_segment_ids"] = timed_leg(agg_local(feats, seg), args.warmup, n_steps, args.warmup)
        leg_runs["headline_shape"] = timed_leg(agg_local(feats), args.warmup, n_steps, args.warmup)
        chosen = "headline_shape"
        for name in ("with_segment_ids", "seeds_uniform_over_V"):
            if name in leg_runs and leg_runs[name][0] > 1.02 * leg_runs[chosen][0]:
                chosen = name  # more than 2 % slower: that is the number to quote
        for name, (el, _, _) in leg_runs.items():
            request_legs[name + "_ms"] = el / args.steps * 1e3
        request_legs["headline_is"] = chosen
        elapsed, t_agg, t_smp = leg_runs[chosen]
        if chosen != "headline_shape":
            # the buffers hold the headline-shaped leg's last step (it ran last); redo the chosen leg's last step so
            # that the oracle check below looks at the request that was timed (the steps are deterministic)
            if chosen == "seeds_uniform_over_V":
                cur["seeds"] = seeds_v
            a_r, b_r = do_sample(n_steps - 1)
            agg_local(feats, seg if chosen == "with_segment_ids" else (None, None))(a_r, b_r, n_steps - 1)
            torch.cuda.synchronize()
        headline = "single GPU"
    else:
        # A collective that never completes (a rank died, a link fault) would hang every rank forever: after
        # --watchdog seconds rank 0 prints what has been measured so far and every rank leaves.
        import threading
        if setup_dog is not None:
            setup_dog.cancel()
        progress = {"stage": "first leg"}

        def give_up(reason=None):
            # the headline placement if it finished; otherwise the other placement, named as such
            for name in ("features_sharded", "features_replicated", "features_sharded_serial"):
                if name in legs:
                    best = name
                    break
            else:
                best = None
            if rank == 0:
                emit_result(result_out, {
                    "metric": "sampled-edges/sec + aggregated-vertices/sec",
                    "value": legs[best]["value"] if best else None, "unit": "edges/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": legs[best]["ms_per_step"] if best else None,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "%s: value = %s placement (the only leg that finished) -- %s"
                                           % (args.workload, best, desc)},
                    "error": reason or ("watchdog: no progress for %.0f s in the %s" % (args.watchdog, progress["stage"])),
                    "placements": legs}, args)
            os._exit(4)
        dog = threading.Timer(args.watchdog, give_up)
        dog.daemon = True
        dog.start()

        def guarded(name, fn):
            """A leg that raises on one rank leaves its peers inside a collective: rank 0 reports what it has at
            once; another rank keeps its process alive (so the launcher does not tear rank 0 down) until rank
            0's watchdog has spoken."""
            progress["stage"] = name + " leg"
            try:
                if os.environ.get("GLX_BENCH_FAULT") == "%d:%s" % (rank, name):  # test knob: this rank fails this leg
                    raise RuntimeError("injected fault (GLX_BENCH_FAULT)")
                return fn()
            except Exception as ex:  # noqa: BLE001
                log("rank %d: %s leg failed: %r" % (rank, name, ex))
                if rank == 0:
                    give_up("%s leg failed on rank 0: %r" % (name, ex))
                time.sleep(args.watchdog + 30)
                os._exit(4)
        # First a short SERIAL leg of the headline placement (all kernels and all collectives of a step on one stream, in
        # program order: the most conservative way to drive two communicators): should the concurrent stages of the
        # pipelined legs below ever wedge on real links -- RCCL with more than one rank has never run under this code --
        # the watchdog still has a measured, labelled number of north_star's placement to report.
        if world > 1 or args.force_sharded:
            def flat(a, b, i):
                halo_begin(a, b, i)
                agg_halo(a, b, i)
            ns = min(args.steps, 5)
            el_0, _, _ = guarded("features_sharded_serial",
                                 lambda: timed_leg(flat, args.warmup, args.warmup + ns, min(args.warmup, 2), serial=True))
            legs["features_sharded_serial"] = {"ms_per_step": el_0 / ns * 1e3, "value": world * edges_per_step * ns / el_0,
                                               "steps": ns, "note": "unpipelined: one stream, stages back to back"}
            dog.cancel()
            dog = threading.Timer(args.watchdog, give_up)
            dog.daemon = True
            dog.start()
        # the other end of the placement space next (it shares the sampling exchanges but has no halo step):
        # a fault in the halo leg then still leaves a measured, labelled number
        if replica is not None:
            el_r, ta_r, ts_r = guarded("features_replicated",
                                       lambda: timed_leg(agg_local(replica), args.warmup, n_steps, args.warmup))
            legs["features_replicated"] = {"ms_per_step": el_r / args.steps * 1e3,
                                           "value": world * edges_per_step * args.steps / el_r}
            dog.cancel()
            dog = threading.Timer(args.watchdog, give_up)
            dog.daemon = True
            dog.start()
        # north_star's placement: everything edge-cut, halo-vertex feature exchange per request
        el_h, ta_h, ts_h = guarded("features_sharded", lambda: timed_leg_halo(args.warmup, n_steps, args.warmup))
        legs["features_sharded"] = dict({"ms_per_step": el_h / args.steps * 1e3,
                                         "value": world * edges_per_step * args.steps / el_h}, **last_host_syncs)
        torch.cuda.synchronize()
        if args.features == "replicated" and replica is not None:
            elapsed, t_agg, t_smp, headline = el_r, ta_r, ts_r, "features_replicated"
        else:
            elapsed, t_agg, t_smp, headline = el_h, ta_h, ts_h, "features_sharded"
        # what crossed the links for one hop-2 request (this rank's view)
        a_last, b_last = do_sample(n_steps - 1)
        sample_rows = st_smp.last_sample_rows()  # the hop-2 request
        st_agg.aggregate(agg, b_last.view(-1), None, n1, out=(emb2, cnt2))
        torch.cuda.synchronize()
        halo_stats = st_agg.stats()
        if args.verify:
            verified_legs["features_sharded"] = guarded("verify", verify_sharded)
        def speculated_leg(name):
            """The same steps with a ledger on both stores: after the first step has recorded the two request shapes no
            sampling request exchanges counts.  A confirmation that aborts (a message did not fit) voids the leg: the
            capacities have been raised, the leg starts over -- on every rank, the abort is collective."""
            ledger = glx.Ledger(local_rank).attach(st_smp, st_agg)
            aborts = [0]

            def run():
                while True:
                    try:
                        return timed_leg_halo(args.warmup, n_steps, args.warmup)
                    except glx.GlxError as ex:
                        if ex.code != glx.ABORTED or aborts[0] >= 3:
                            raise
                        aborts[0] += 1
                        torch.cuda.synchronize()
                        log("rank %d: %s: confirmation aborted (%s); repeating the leg" % (rank, name, ex))
            el, ta, ts = guarded(name, run)
            legs[name] = dict({"ms_per_step": el / args.steps * 1e3, "value": world * edges_per_step * args.steps / el},
                              **last_host_syncs)
            legs[name]["ledger"] = dict(ledger.stats(), legs_repeated_after_abort=aborts[0])
            if args.verify:
                verified_legs[name] = guarded("verify", verify_sharded)
            ledger.close()
            return el, ta, ts
        exchange_mode = "one count exchange per request"
        if args.speculate == "on":
            dog.cancel()
            dog = threading.Timer(args.watchdog, give_up)
            dog.daemon = True
            dog.start()
            el_s, ta_s, ts_s = speculated_leg("features_sharded_speculated")
            ok_s = verified_legs.get("features_sharded_speculated", True)
            if headline == "features_sharded" and ok_s and el_s < elapsed:
                elapsed, t_agg, t_smp = el_s, ta_s, ts_s
                exchange_mode = "speculated sampling exchanges (ledger), one count exchange per step"
        if args.pure_leg == "on":
            # north_star's path without any cache: drop the hot-row replica and detach the graph replica, then the same steps
            dog.cancel()
            dog = threading.Timer(args.watchdog, give_up)
            dog.daemon = True
            dog.start()
            st_agg.set_cache(np.empty(0, np.int64))
            st_smp.set_graph_replica(None)
            torch.cuda.synchronize()
            el_p, _, _ = guarded("edge_cut_pure", lambda: timed_leg_halo(args.warmup, n_steps, args.warmup))
            legs["edge_cut_pure"] = dict({"ms_per_step": el_p / args.steps * 1e3,
                                          "value": world * edges_per_step * args.steps / el_p}, **last_host_syncs)
            a_p, b_p = do_sample(n_steps - 1)
            pure_rows = st_smp.last_sample_rows()
            st_agg.aggregate(agg, b_p.view(-1), None, n1, out=(emb2, cnt2))
            torch.cuda.synchronize()
            legs["edge_cut_pure"]["halo_exchange_hop2"] = st_agg.stats()
            legs["edge_cut_pure"]["sampling_exchange_hop2"] = pure_rows
            if args.verify:
                verified_legs["edge_cut_pure"] = guarded("verify", verify_sharded)
            if args.speculate == "on":
                dog.cancel()
                dog = threading.Timer(args.watchdog, give_up)
                dog.daemon = True
                dog.start()
                speculated_leg("edge_cut_pure_speculated")
            if args.design_r == "on":
                # SURVEY 8(e)'s ablation: the reference's own shape of a distributed aggregation -- every owner reduces the
                # ids it holds, the requester folds the P partial results (AggregatingResponse::Stitch,
                # aggregating_request.cc:172-213) -- instead of the halo-row exchange; same steps, same placement (R keeps
                # no replica).  Traffic per request: P x segments x (4D + 4) bytes back instead of 4D per distinct remote id.
                dog.cancel()
                dog = threading.Timer(args.watchdog, give_up)
                dog.daemon = True
                dog.start()

                def agg_r(a, b, i):
                    st_agg.aggregate(agg, b.view(-1), None, n1, out=(emb2, cnt2), partial=True)
                    st_agg.aggregate(agg, a.view(-1), None, B0, out=(emb1, cnt1), partial=True)
                el_d, _, _ = guarded("edge_cut_pure_design_r", lambda: timed_leg(agg_r, args.warmup, n_steps, args.warmup))
                legs["edge_cut_pure_design_r"] = {
                    "ms_per_step": el_d / args.steps * 1e3, "value": world * edges_per_step * args.steps / el_d,
                    "note": "partial reduce on the owners + fold on the requester (glx_dist_aggregate_partial) instead of the "
                            "halo-row exchange; Max / Min / counts equal the single store exactly, Sum / Mean / Prod within "
                            "1e-5 relative (per-shard partials are folded: the reference's distributed mode reassociates "
                            "the same way)"}
                if args.verify:
                    def verify_r():
                        i = n_steps - 1
                        a, b = do_sample(i)
                        agg_r(a, b, i)
                        wa, _ = whole[0].sample(sampler, seeds[i], k1, seed=42, call_counter=4 * i)
                        wb, _ = whole[0].sample(sampler, wa.view(-1), k2, seed=42, call_counter=4 * i + 1)
                        we2, wc2 = whole[1].aggregate(agg, wb.view(-1), None, n1)
                        we1, wc1 = whole[1].aggregate(agg, wa.view(-1), None, B0)
                        torch.cuda.synchronize()
                        exact = agg in ("MaxAggregator", "MinAggregator")
                        same = (lambda x, y: torch.equal(x.view(torch.int32), y.view(torch.int32))) if exact else \
                               (lambda x, y: bool(((x - y).abs() <= 1e-5 * y.abs() + 1e-6).all()))
                        ok = bool(torch.equal(b, wb) and torch.equal(cnt2, wc2) and torch.equal(cnt1, wc1)
                                  and same(emb2, we2) and same(emb1, we1))
                        if world > 1:
                            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=ctl)
                            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                            ok = bool(flag.item())
                        return ok
                    verified_legs["edge_cut_pure_design_r"] = guarded("verify", verify_r)
        dog.cancel()

    # the last timed step's outputs against the oracle, before anything else touches the buffers
    oracle_check = None
    if args.verify_oracle == "on" and not sharded and world == 1:
        last_i = (kernel_steps - 1) if use_graph else (n_steps - 1)
        nb1, ed1, nb2, ed2 = bufs[last_i % len(bufs)]
        torch.cuda.synchronize()
        try:
            oracle_check = verify_vs_oracle(args, wl, dev, cur["seeds"][last_i],
                                            dict(n1=nb1, e1=ed1, n2=nb2, e2=ed2, emb2=emb2, cnt2=cnt2, emb1=emb1, cnt1=cnt1),
                                            (4 * last_i, 4 * last_i + 1))
        except Exception as ex:  # noqa: BLE001 -- a checker that cannot run is reported, not fatal to the line
            oracle_check = {"ok": None, "error": repr(ex)}

    cpu = None
    if host_edges is not None:
        t1 = time.time()
        cpu = cpu_baseline(wl, host_edges[0], host_edges[1], host_edges[2], args, host_edges[3])
        log("cpu baseline done in %.1fs: %s" % (time.time() - t1, cpu and "%.3g edges/s" % cpu["value"]))

    # how much of the hop-2 work is the reference's default-fill path (frontier vertices
    # without out-edges)?  Reported so the workload can be judged.
    empty_frac = None
    if st_smp is None:
        a_last, _ = do_sample(n_steps - 1)
        torch.cuda.synchronize()
        empty_frac = float((graph.degrees(a_last.view(-1)) == 0).double().mean().item())

    verified = None
    if verified_legs:
        verified = all(bool(v) for v in verified_legs.values())
        log("verify: sharded (halo exchange) == unpartitioned, bit for bit, on every rank: %s" % verified_legs)

    value = world * edges_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel: the hop-2 segmented reduce (first aggregate launch of each step)
    agg2 = t_agg[0::2]
    agg1 = t_agg[1::2]
    smp2 = t_smp[1::2]  # hop-2 sampler launches (second sample launch of each step)
    avg_agg2_ms = float(np.mean(agg2)) if len(agg2) else float("nan")
    if sharded:
        last_ids2 = b_last
    else:
        last_ids2 = bufs[(n_steps - 1) % len(bufs)][2]  # hop-2 neighbour ids of the last TIMED step
    roof = roofline_aggregate(agg, D, n1, n2, avg_agg2_ms, int(len(agg2)), last_ids2, args.workload, B0,
                              sources=3 if headline == "features_sharded" else 1, offline_ok=not sharded)
    roof_smp = None
    if len(smp2) and not sharded:
        roof_smp = roofline_sampler(sampler, k2, n1, n2, float(np.mean(smp2)), int(len(smp2)), args.workload, B0)
    if args.roofline_probes == "on" and not sharded:
        # what the memory system of THIS box delivers to hand-written kernels with known byte counts (glx_probe.hip),
        # and the reduce itself on uniformly random rows of the whole table (no reuse a cache could serve)
        peaks = {k: glx.probe_bandwidth(k, 2 << 30, reps=10, device=local_rank)["gbps"] for k in ("stream_read", "copy", "triad")}
        row_b = 4 * D
        table_b = (min(V * row_b, 8 << 30) // (4096 * row_b)) * (4096 * row_b)
        peaks["gather_rows_%dB_uniform" % row_b] = glx.probe_bandwidth("gather_rows", table_b, units=n2, unit_bytes=row_b,
                                                                         reps=5, device=local_rank)["gbps"]
        roof["peak_measured"] = dict(peaks, unit="GB/s", note="glx_probe_bandwidth in this run: 2 GiB streaming read / copy / "
                                     "STREAM triad kernels (16 B per lane, non-temporal), and the reduce's row gather on "
                                     "uniformly random rows without the reduce")
        best = max(peaks["stream_read"], peaks["copy"], peaks["triad"])
        fake = torch.randint(0, V, (n2,), generator=gen, device=dev)
        for _ in range(2):
            feats.aggregate(agg, fake, None, n1, out=(emb2, cnt2))
        torch.cuda.synchronize()
        glx.profile_enable(True)
        for _ in range(5):
            feats.aggregate(agg, fake, None, n1, out=(emb2, cnt2))
        torch.cuda.synchronize()
        glx.profile_enable(False)
        ms = float(np.mean(glx.profile_collect(glx.KERNEL_AGGREGATE)))
        bytes_alg = roof["algorithmic_bytes_per_launch"]
        cf = bytes_alg / (ms * 1e-3) / 1e9
        roof["cache_free"] = {"rows": "uniform over all %d rows: no reuse a cache could serve, so the algorithmic bytes ARE the "
                                      "HBM traffic (checked with FETCH_SIZE / WRITE_SIZE: profiles/r02/SUMMARY.md)" % V,
                              "table_gb": V * D * 4 / 1e9,
                              "infinity_cache_share_upper_bound": min(1.0, 0.256 / (V * D * 4 / 1e9)),
                              "avg_launch_ms": ms, "launches_timed": 5, "achieved": cf, "frac": cf / HBM_PEAK_GBS,
                              "frac_of_measured_stream_peak": min(1.0, cf / best),
                              # the same bytes/s against what THIS box gives a bare gather of uniformly random rows (no ids,
                              # no reduce, almost no output): HBM's random-access ceiling, which no row layout moves
                              # (profiles/r04/layout_probe.txt); not clamped -- the probe gathers from <= 8 GB
                              "frac_of_measured_row_gather": cf / peaks["gather_rows_%dB_uniform" % row_b]}
        # the roofline fraction of the kernel: same kernel, same request shape, measured in this run, on the input
        # where bytes moved are known exactly
        roof["frac"] = cf / HBM_PEAK_GBS
        roof["achieved"] = cf
        roof["achieved_basis"] = "cache-free leg: algorithmic bytes (== memory-side traffic there) / its average launch time"
        roof["frac_basis"] = ("cache-free leg of this run: the same kernel on the same request shape with ids uniform over the "
                              "table (algorithmic bytes == memory-side traffic; at most the 256 MB Infinity Cache's share of the "
                              "table -- cache_free.infinity_cache_share_upper_bound -- can be on-die hits) / 8 TB/s; the timed "
                              "launches themselves run %.2fx "
                              "faster than that because hub rows are re-read from cache (algorithmic_over_peak), and their "
                              "HBM traffic is only bounded in-run (frac_compulsory) or known offline (frac_traffic_offline)"
                              % (ms / avg_agg2_ms))
        del fake
        if roof_smp is not None:
            g32 = glx.probe_bandwidth("gather32", (E * 32 // 4096) * 4096, units=n2, reps=5, device=local_rank)
            rate = n2 / (g32["ms"] * 1e-3)
            roof_smp["gather32_uniform"] = {"records_per_s": rate, "ms": g32["ms"], "table_bytes": (E * 32 // 4096) * 4096,
                                            "note": "glx_probe_bandwidth GATHER32 in this run: as many aligned 32-byte records as the "
                                                    "hop-2 launch draws, from uniformly random positions of a table the size of the "
                                                    "packed alias records, 16 B written per record"}
            roof_smp["draws_over_uniform_gather_rate"] = roof_smp["draws_per_s"] / rate
            roof_smp["frac_of_gather_ceiling"] = min(1.0, roof_smp["draws_per_s"] / rate)
    smp_ms = float(np.sum(t_smp)) / max(kernel_steps, 1)
    agg_ms = float(np.sum(t_agg)) / max(kernel_steps, 1)
    res = {
        "metric": "sampled-edges/sec + aggregated-vertices/sec (2-hop sample + aggregate per step)",
        "metric_note": "every sampled vertex is aggregated once, so the step rate counts both",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s%s" % (args.workload,
                                             "" if not sharded else "value = %s placement, %s (the faster of the two "
                                             "exchange modes timed; both are in placements) -- " % (headline, exchange_mode), desc),
                   "seeds_per_step_per_gpu": B0,
                   "seeds": ("uniform over ALL vertices (SURVEY 8(d)); about half have no out-edges and default-fill"
                             if (not sharded and not use_graph and request_legs.get("headline_is") == "seeds_uniform_over_V")
                             else "uniform over the vertices that have out-edges (a training set has neighbours; harder than "
                                  "SURVEY 8(d)'s uniform over V, timed beside as request_shapes.seeds_uniform_over_V_ms)"
                             if args.seed_dist == "uniform" else
                             "degree-biased: the sources of uniformly drawn edges") + ", fresh batch every step",
                   "segment_ids": ("explicit int32 tensor per aggregation (the reference's request shape)"
                                   if (not sharded and not use_graph and request_legs.get("headline_is") == "with_segment_ids")
                                   else "implied by the dense sampler response (segment i = row i's neighbours); the explicit-"
                                        "tensor shape is timed beside as request_shapes.with_segment_ids_ms"),
                   "fanout": [k1, k2], "sampler": sampler, "aggregator": agg, "dim": D,
                   "arithmetic": "int64 ids / edge ids (bit-exact), f32 features and aggregates",
                   "nodes": V, "edges": E, "hop2_rows_without_out_edges_fraction": empty_frac,
                   "vertex_labels": "raw RMAT ids" if args.no_scramble else "RMAT ids relabeled by a fixed random permutation (Graph500-style)",
                   "parallelism": placement,
                   "pipelined_two_streams": bool(pipelined) and not use_graph, "hipgraph_step": bool(use_graph)},
        "phases": {
            "sampling_kernels_ms_per_step": smp_ms,
            "aggregation_kernels_ms_per_step": agg_ms,
            "sampled_edges_per_s_sampling_only": world * edges_per_step / (smp_ms * 1e-3) if smp_ms > 0 else None,
            "aggregated_vertices_per_s_aggregation_only": world * edges_per_step / (agg_ms * 1e-3) if agg_ms > 0 else None,
            "aggregate_hop1_avg_ms": float(np.mean(agg1)) if len(agg1) else None,
        },
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    if roof_smp is not None:
        res["roofline_sampler"] = roof_smp
    if not sharded and not use_graph and request_legs:
        res["request_shapes"] = request_legs
    if oracle_check is not None:
        res["verified_vs_oracle"] = oracle_check["ok"]
        res["oracle_check"] = oracle_check
    if sharded:
        res.update(comm_facts(comm_s, args.share_device))
        # what the leg `value` reports keeps on EVERY GPU besides its own shard (placements.edge_cut_pure keeps nothing)
        res["config"]["replicated_per_gpu"] = {
            "feature_rows": int(hot.shape[0]) if headline == "features_sharded" else V,
            "feature_row_fraction": (float(hot.shape[0]) / V) if headline == "features_sharded" else 1.0,
            "feature_bytes": (int(hot.shape[0]) if headline == "features_sharded" else V) * D * 4,
            "graph_edges": int(graph_replica.num_edges) if graph_replica is not None else 0,
            "graph_edge_fraction": (float(graph_replica.num_edges) / E) if graph_replica is not None else 0.0,
            "graph_replica": ("on" if graph_replica is not None else "off") + " (--graph-replica %s)" % args.graph_replica}
        res["value_features_sharded"] = legs["features_sharded"]["value"]
        res["value_features_sharded_speculated"] = legs.get("features_sharded_speculated", {}).get("value")
        res["value_edge_cut_pure"] = legs.get("edge_cut_pure", {}).get("value")
        res["value_edge_cut_pure_design_r"] = legs.get("edge_cut_pure_design_r", {}).get("value")
        res["value_features_replicated"] = legs.get("features_replicated", {}).get("value")
        res["placements"] = legs
        res["halo_exchange_hop2"] = dict(halo_stats, hot_rows=int(hot.shape[0]), hot_fraction=args.hot_fraction,
                                         hot_rows_chosen_by=args.hot_by,
                                         note="one rank's last hop-2 request: ids by source, distinct halo rows, "
                                              "bytes over the transport")
        res["sampling_exchange_hop2"] = dict(sample_rows, graph_replica_edges=(graph_replica.num_edges if graph_replica
                                                                               is not None else 0), graph_edges=E,
                                             note="one rank's last hop-2 request: rows served by the local graph "
                                                  "replica / sent to another rank")
    if verified is not None:
        res["verified_sharded_equals_unpartitioned"] = verified
        res["verified_legs"] = verified_legs
    if cpu:
        res["gpu_over_cpu"] = value / cpu["value"]
    if extras["host_boundary"] and rank == 0:
        res["host_boundary"] = host_boundary_rate(args)
    if extras["small_batches"]:
        # SURVEY 8(a) fixes B0 in {1024, 8192, 65536}: the launch-bound sizes on the same resident store, each step ONE
        # hipGraph launch (glx_plan), plans alternating over --graph-streams streams
        small = {}
        for b in (1024, 8192):
            plans = [glx.Plan([graph, graph], sampler, [k1, k2], b, features=[feats, feats], agg=agg, seed=42)
                     for _ in range(args.graph_streams)]
            streams = [torch.cuda.Stream(device=dev) for _ in plans]
            reps = 400

            def run(i):
                with torch.cuda.stream(streams[i % len(plans)]):
                    plans[i % len(plans)].run(seeds[i % n_steps, :b].contiguous(), call_counter=4 * i)
            for i in range(40):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40, 40 + reps):
                run(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            small[str(b)] = {"ms_per_step": dt / reps * 1e3, "value": (b * k1 + b * k1 * k2) * reps / dt, "steps": reps}
            for pl in plans:
                pl.close()  # their output tensors would otherwise keep the plans -- and through them the store -- alive
            del plans, streams
        res["small_batches"] = dict(small, note="same store, B0 seeds per step, step = one hipGraph launch (glx_plan), %d plans "
                                                "alternating on as many streams; value in edges/s" % args.graph_streams)
    if extras["edge_cut_probe"]:
        res["edge_cut_world1"] = edge_cut_world1(args)
        # (after this process's store is gone: eight ranks' shards, replicas and buffers want ~110 GB of the HBM)
        want_p8 = True
    else:
        want_p8 = False
    if extras["other_configs"]:
        # free this process's store first: c5 needs most of the HBM for its build
        del graph, feats
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info(dev)
        log("before the other configs: %.1f of %.1f GB of HBM free in this process's view" % (free_b / 1e9, total_b / 1e9))
        res["other_configs"] = other_configs(args)
    if want_p8 and rank == 0:
        if "graph" in dir():
            del graph, feats
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        res["edge_cut_p8_one_gpu"] = edge_cut_p8_one_gpu(args)
        log("eight ranks on this one GPU: %s" % res["edge_cut_p8_one_gpu"])
    if rank == 0:
        emit_result(result_out, res, args)
    if sharded:
        if world > 1:
            # the result line is out; a communicator teardown that waits for a peer must not keep the node busy
            import threading
            bye = threading.Timer(60.0, lambda: os._exit(3 if verified is False else 0))
            bye.daemon = True
            bye.start()
        st_smp.close()
        st_agg.close()
        comm_s.close()
        comm_a.close()
        dist.destroy_process_group()
    if verified is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
