"""Test aid, run in a process of its own by tests/test_gpu_parity.py::test_pinned_host_buffers_are_written_directly:
host-pointer calls write their results straight into caller buffers pinned with glx_host_register (no staging copy);
the answers equal those of pageable buffers, for sampling (plain and filtered) and aggregation, and registration can
be undone.

Why a process of its own (round 6): after glx_host_unregister the ROCm 7.0 runtime was seen to fault on LATER pageable
host-to-device copies of the same process -- deterministically when the formerly registered pages went back to the heap
(include/glx.h says so), and still once in ~10 full `pytest -m gpu` runs with those pages kept allocated: the test that
ran NEXT in the suite's process got "an illegal memory access was encountered" from its first hipMemcpyAsync of a plain
numpy array, and every GPU test after it failed (or the process aborted: scripts/r06/crash_hunt.sh,
profiles/r06/crash_hunt.txt).  Here the process ends right after the last unregistration."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "graph-learn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import glx  # noqa: E402
import synth  # noqa: E402

_KEEP_FORMERLY_PINNED = []


def check():
    rng = np.random.default_rng(12)
    rp, col, eid, w = synth.small_graph(2000, 40000, seed=9, weighted=True, hub_degree=400)
    X = rng.standard_normal((2000, 48)).astype(np.float32)
    g, f = glx.Graph(rp, col, eid, w), glx.Features(X)
    L = glx.lib()
    ids = rng.integers(-2, 2003, 3000).astype(np.int64)
    vals = rng.integers(0, 2000, 3000).astype(np.int64)

    owners = []

    def pinned(shape, dtype):
        # Whole pages of an anonymous mapping of their own, 2 MiB aligned -- NOT a range of the malloc heap: a process that
        # has registered heap ranges and also holds heap memory marked MADV_HUGEPAGE (numpy marks every array of 4 MiB
        # or more) gets "an illegal memory access" from later pageable host-to-device copies on ROCm 7.0
        # (scripts/r06/repro/hostreg_pageable.hip, no glx code involved; include/glx.h states the rule).
        import mmap
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        span = (nbytes + 4095) // 4096 * 4096
        gran = 2 << 20
        mm = mmap.mmap(-1, span + gran, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        raw = np.frombuffer(mm, np.uint8)
        owners.append((mm, raw))
        off = (-raw.ctypes.data) % gran
        a = raw[off:off + nbytes].view(dtype).reshape(shape)
        assert a.ctypes.data % gran == 0
        assert L.glx_host_register(ctypes.c_void_p(a.ctypes.data), span) == 0, L.glx_last_error()
        a.fill(0)
        return a
    # a range of the brk heap is refused by name (the hazard of include/glx.h); nothing is registered
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    blocks = [libc.malloc(48 << 10) for _ in range(4)]  # below the mmap threshold: the main arena
    heap_lo = heap_hi = 0
    with open("/proc/self/maps") as fh:
        for ln in fh:
            if "[heap]" in ln:
                heap_lo, heap_hi = (int(x, 16) for x in ln.split()[0].split("-"))
    in_heap = [b for b in blocks if heap_lo <= b < heap_hi]
    assert in_heap, "no small malloc block landed in the brk heap"
    page = (in_heap[0] + 4095) // 4096 * 4096
    assert L.glx_host_register(ctypes.c_void_p(page), 8192) == 3
    assert b"malloc heap" in L.glx_last_error()
    for b in blocks:
        libc.free(b)
    bufs = []
    try:
        for name in glx.SAMPLER_IDS:
            want = g.sample(name, ids, 7, seed=3, call_counter=5)
            n, e = pinned((3000, 7), np.int64), pinned((3000, 7), np.int64)
            bufs += [n, e]
            g.sample(name, ids, 7, seed=3, call_counter=5, out=(n, e))
            assert np.array_equal(n, want[0]) and np.array_equal(e, want[1]), name
        want = g.sample_filtered("TopkSampler", ids, 5, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals)
        nbr = want[0]
        seg = (np.arange(nbr.size) // 5).astype(np.int32)
        for name in glx.AGGREGATOR_IDS:
            we, wc = f.aggregate(name, nbr.reshape(-1), seg, 3000, default_attr=0.5)
            emb, cnt = pinned((3000, 48), np.float32), pinned((3000,), np.int32)
            bufs += [emb, cnt]
            f.aggregate(name, nbr.reshape(-1), seg, 3000, default_attr=0.5, out=(emb, cnt))
            assert np.array_equal(cnt, wc) and np.array_equal(emb.view(np.uint32), we.view(np.uint32)), name
    finally:
        for a in bufs:
            assert L.glx_host_unregister(ctypes.c_void_p(a.ctypes.data)) == 0
        # the mappings stay for the life of this (short) process
        _KEEP_FORMERLY_PINNED.extend(owners)


if __name__ == "__main__":
    assert glx.device_count() >= 1
    check()
    print("PINNED_OK", flush=True)
