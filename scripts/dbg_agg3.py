import sys; sys.path.insert(0,'graph-learn_amd'); sys.path.insert(0,'tests')
import numpy as np, glx, os
import test_gpu_parity as T
T.test_golden_kat_topk(); print("kat ok")
if len(sys.argv) > 1 and sys.argv[1] >= "1":
    T.test_golden_python_fixture_topk(); print("py ok")
if len(sys.argv) > 1 and sys.argv[1] >= "2":
    T.test_golden_rand_graph_alias_and_topk(); print("rand ok")
for D in (1, 4):
    X=np.arange(100*D,dtype=np.float32).reshape(100,D).copy()
    f=glx.Features(X)
    nid=np.arange(10,dtype=np.int64); seg=np.array([1,2,2,3,3,3,4,4,4,4],np.int32)
    for rep in range(3):
        e,c=f.aggregate("SumAggregator",nid,seg,5)
        print(D, rep, c, e[:, :1].T.tolist())
