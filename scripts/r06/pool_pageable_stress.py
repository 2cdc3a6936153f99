"""Round 6: the C++ layer's response pool (host/src/base.cc: blocks registered with the GPU runtime, since this round cut
explicitly from anonymous mappings of their own; the old posix_memalign pool passes this stress too) beside numpy arrays of 4 MiB and more (MADV_HUGEPAGE) and pageable host-to-device
copies, in ONE process: rounds of NeighborSampler requests through the Python API (responses of 0.3 - 8 MB in pool
blocks) interleaved with glx.Graph builds from fresh numpy arrays.  Prints the round of the first GPU error, or none.
    python scripts/r06/pool_pageable_stress.py [rounds]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("graph-learn_amd", os.path.join("graph-learn_amd", "python"), "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import glx  # noqa: E402
import graphlearn as gl  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(3)
d = tempfile.mkdtemp()
with open(os.path.join(d, "v"), "w") as f:
    f.write("id:int64\tfeature:string\n")
    for i in range(2000):
        f.write("%d\t%f:%f:%f:%f\n" % (i, i * 0.1, i * 0.2, i * 0.3, i * 0.4))
with open(os.path.join(d, "e"), "w") as f:
    f.write("sid:int64\tdid:int64\tweight:float\n")
    for i in range(2000):
        for j in rng.integers(0, 2000, 12):
            f.write("%d\t%d\t%f\n" % (i, j, 0.1 + rng.random()))
g = gl.Graph().node(os.path.join(d, "v"), "v", decoder=gl.Decoder(attr_types=["float"] * 4)) \
    .edge(os.path.join(d, "e"), ("v", "v", "e"), decoder=gl.Decoder(weighted=True)).init()
held = []
try:
    for r in range(rounds):
        ids = rng.integers(0, 2000, int(rng.integers(2000, 40000))).astype(np.int64)
        res = g.neighbor_sampler("e", expand_factor=[int(rng.integers(5, 26))], strategy="edge_weight").get(ids)
        nodes = res.layer_nodes(1)
        assert nodes.ids.shape[0] == ids.shape[0] and nodes.float_attrs.shape[-1] == 4
        held.append(nodes)
        if len(held) > 6:
            held.pop(int(rng.integers(0, len(held))))
        for _ in range(3):
            E = int(rng.integers(12_000, 1_000_000))
            deg = rng.multinomial(E, np.ones(300) / 300)
            rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
            w = (rng.random(E) + 1e-3).astype(np.float32)
            dev = glx.Graph(rp, (np.arange(E, dtype=np.int64) * 3) % 500, np.arange(E, dtype=np.int64), w)
            dev.sample("TopkSampler", np.arange(50, dtype=np.int64), 4)
            del dev
except Exception as ex:  # noqa: BLE001
    print("GPU error in round %d of %d: %s" % (r, rounds, str(ex)[:300]), flush=True)
    os._exit(3)
print("%d rounds of pool-backed responses beside pageable copies of fresh numpy arrays: no error" % rounds, flush=True)
g.close()
