import sys; sys.path.insert(0,'graph-learn_amd'); sys.path.insert(0,'tests')
import numpy as np, glx, os
g = dict(np.load('tests/golden/kat_sampler.npz'))
dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
nbr, _ = dev.sample("TopkSampler", np.array([0, 1], np.int64), 2)
print(nbr)
for D in (1, 4):
    X=np.arange(100*D,dtype=np.float32).reshape(100,D).copy()
    f=glx.Features(X)
    nid=np.arange(10,dtype=np.int64); seg=np.array([1,2,2,3,3,3,4,4,4,4],np.int32)
    for rep in range(3):
        e,c=f.aggregate("SumAggregator",nid,seg,5)
        print(D, rep, c, e[:, :1].T.tolist())
