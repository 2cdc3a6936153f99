#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run8; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_pyapi.py -m gpu -q -x > $O/pytest.log 2>&1; tail -12 $O/pytest.log
python - <<'PY'
import sys, time, numpy as np
sys.path[:0]=['graph-learn_amd','tests']
import glx, os
rng=np.random.default_rng(0)
U=200000; items=np.arange(U,dtype=np.int64); keys=np.stack([rng.integers(0,50,U), rng.integers(0,7,U)]).astype(np.int64)
n_users=100000; deg=8
src=np.repeat(np.arange(n_users),deg).astype(np.int64); dst=rng.integers(0,U,n_users*deg).astype(np.int64)
rp=np.arange(0,n_users*deg+1,deg,dtype=np.int64)
g=glx.Graph(rp,dst,np.arange(dst.shape[0],dtype=np.int64))
tab=glx.CondTable(items,None,keys)
for batch in (1024, 65536):
    s=rng.integers(0,n_users,batch).astype(np.int64); d=dst[s*deg]; dk=np.stack([keys[0,d],keys[1,d]],axis=1)
    props=np.array([0.5,0.3],np.float32)
    for mode in ("parallel","sequential"):
        if mode=="sequential": os.environ["GLX_COND_SEQUENTIAL"]="1"
        else: os.environ.pop("GLX_COND_SEQUENTIAL",None)
        tab.sample(g,s,d,dk,props,10,seed=1,call_counter=0)
        t0=time.perf_counter(); out=tab.sample(g,s,d,dk,props,10,seed=1,call_counter=1); dt=time.perf_counter()-t0
        print("cond negative batch %6d x 10, %-10s rows: %8.2f ms (host pointers, incl. copies)"%(batch,mode,dt*1e3), flush=True)
PY
