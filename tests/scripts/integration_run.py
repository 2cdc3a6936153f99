"""Runs ONE library that speaks the glref_* driver API -- oracle/_ref/libglref.so (the reference as it is) or
integration/_build/libgl_glx.so (the reference's registry / requests / storages with glx operator bodies) -- over a
seeded workload and saves every response to an .npz.  One library per process: GLX_REF_LIB selects it.
Usage: integration_run.py <out.npz> <seed>"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_bindings import RefLib, REF_SO  # noqa: E402


def workload(seed):
    rng = np.random.default_rng(seed)
    V, E, D = 400, 6000, 24
    src = rng.integers(0, V // 2, E).astype(np.int64) * 2 + 1      # odd raw ids: rows are found through the id map
    dst = rng.integers(0, V, E).astype(np.int64)
    w = (rng.random(E) * 0.99 + 0.01).astype(np.float32) + np.arange(E, dtype=np.float32) * 2.0 ** -20  # tie-free per row
    ts = rng.integers(0, 1000, E).astype(np.int64)
    ids = rng.permutation(V).astype(np.int64) * 3 + 5
    X = rng.standard_normal((V, D)).astype(np.float32)
    X[rng.random((V, D)) < 0.02] = -50.0
    seeds = np.concatenate([rng.choice(src, 300), np.array([0, 2, 10 ** 9], np.int64)])  # three ids without a row
    sizes = rng.integers(0, 9, 120)
    sizes[[0, 5, 119]] = 0
    seg = np.repeat(np.arange(120, dtype=np.int32), sizes)
    nid = ids[rng.integers(0, V, seg.shape[0])].copy()
    nid[rng.random(seg.shape[0]) < 0.1] = -7                          # unknown ids -> the default attribute
    return dict(V=V, E=E, D=D, src=src, dst=dst, w=w, ts=ts, ids=ids, X=X, seeds=seeds, seg=seg, nid=nid)


def main():
    out_path, seed = sys.argv[1], int(sys.argv[2])
    wl = workload(seed)
    out = {"library": np.array(os.path.basename(REF_SO))}
    for padding in (1, 0):
        ref = RefLib(storage_mode=2, padding_mode=padding, default_neighbor_id=-3, default_float_attr=0.5)
        is_glx = hasattr(ref.L, "glx_integration_set_stream")
        if is_glx:
            ref.L.glx_integration_set_stream.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
            ref.L.glx_integration_set_stream(1234, 10)
        ref.add_edges("e", wl["src"], wl["dst"], wl["w"])
        ref.add_edges_timestamped("t", wl["src"], wl["dst"], wl["ts"], wl["w"])
        ref.add_nodes("n", wl["ids"], wl["X"])
        rows = np.unique(wl["src"])
        rp, col, eid, ws = ref.export_csr("e", rows, 4096)
        out.update({"rows": rows, "row_ptr": rp, "col": col, "eid": eid, "weight": ws})
        tag = "pad%d_" % padding
        # call order fixes the call counters of the glx bodies: 10, 11, 12, ... (set_stream above)
        for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"):
            for k in (3, 40):
                if padding == 0 and k == 40 and name == "EdgeWeightSampler" and not is_glx:
                    continue  # the reference reads past the row here: not run, not compared
                nbr, ed = ref.sample("e", name, wl["seeds"], k, fresh_thread=False)
                out[tag + "%s_k%d_nbr" % (name, k)] = nbr
                out[tag + "%s_k%d_eid" % (name, k)] = ed
        # operators this build leaves to the reference's own bodies, through the same registry
        deg, nbr, ed = ref.sample_full("e", wl["seeds"], 7)
        out.update({tag + "full_deg": deg, tag + "full_nbr": nbr, tag + "full_eid": ed})
        # filters: neighbour id == value, and edge timestamp > value
        fv = wl["dst"][:wl["seeds"].shape[0]].copy()
        nbr, ed = ref.sample_filtered("e", "TopkSampler", wl["seeds"], 4, dict(type=1, field=1, values=fv), fresh_thread=False)
        out.update({tag + "topk_flt_id_nbr": nbr, tag + "topk_flt_id_eid": ed})
        tv = np.full(wl["seeds"].shape[0], 500, np.int64)
        nbr, ed = ref.sample_filtered("t", "TopkSampler", wl["seeds"], 4, dict(type=2, field=2, values=tv), fresh_thread=False)
        out.update({tag + "topk_flt_ts_nbr": nbr, tag + "topk_flt_ts_eid": ed})
        for name in ("SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"):
            emb, cnt = ref.aggregate("n", name, wl["nid"], wl["seg"], 120, wl["D"])
            out[tag + name + "_emb"] = emb
            out[tag + name + "_cnt"] = cnt
        # AggregatingResponse::Stitch on the host keeps working on the operators' Init / Agg / FinalFunc
        parts = np.stack([out[tag + "SumAggregator_emb"], out[tag + "SumAggregator_emb"] * 2])
        cnts = np.stack([out[tag + "SumAggregator_cnt"], out[tag + "SumAggregator_cnt"]])
        for name in ("SumAggregator", "MeanAggregator", "MaxAggregator"):
            emb, cnt = ref.aggregate_stitch(name, parts, cnts)
            out[tag + name + "_stitch_emb"] = emb
            out[tag + name + "_stitch_cnt"] = cnt
        ref.close()
    np.savez(out_path, **out)
    print("integration_run ok: %s, %d arrays" % (os.path.basename(REF_SO), len(out)))


if __name__ == "__main__":
    main()
