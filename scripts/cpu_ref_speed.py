import sys, time; sys.path.insert(0,'graph-learn_amd'); sys.path.insert(0,'tests')
import numpy as np, synth, ctypes, os
from oracle_bindings import RefLib, _p
V, E = 1_000_000, 10_000_000
t=time.time(); src,dst = synth.rmat_edges_numpy(20, E, V, 4); print("gen", time.time()-t)
w = (np.random.default_rng(1).random(E)*0.99+0.01).astype(np.float32)
for mode in (2,3):
    ref = RefLib(storage_mode=mode)
    t=time.time(); ref.L.glref_add_edges(ref.h, b"e", _p(src), _p(dst), _p(w), E); t1=time.time()-t
    t=time.time(); ref.L.glref_build_graph(ref.h, b"e"); t2=time.time()-t
    print("mode", mode, "add %.1fs build %.1fs" % (t1,t2))
    seeds = np.random.default_rng(2).integers(0, V, 1024*8*4).astype(np.int64)
    for T in (1, 8):
        out = ctypes.c_int64()
        dt = ref.L.glref_time_sample_2hop(ref.h, b"e", b"EdgeWeightSampler", _p(seeds), 1024, 25, 10, 2, T, ctypes.byref(out))
        print("  EdgeWeight T=%d: %.2fs %d edges -> %.2f M edges/s" % (T, dt, out.value, out.value/dt/1e6))
    for name in (b"RandomSampler", b"RandomWithoutReplacementSampler", b"TopkSampler"):
        out = ctypes.c_int64()
        dt = ref.L.glref_time_sample_2hop(ref.h, b"e", name, _p(seeds), 1024, 25, 10, 2, 8, ctypes.byref(out))
        print("  %s T=8: %.2fs -> %.2f M edges/s" % (name.decode(), dt, out.value/dt/1e6))
    ref.close()
