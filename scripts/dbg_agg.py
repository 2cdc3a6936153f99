import sys; sys.path.insert(0,'graph-learn_amd'); sys.path.insert(0,'tests')
import numpy as np, glx
from oracle_bindings import Oracle
o=Oracle()
for D in (1,4,8,64,128,256):
    X=np.arange(100*D,dtype=np.float32).reshape(100,D).copy()
    f=glx.Features(X)
    nid=np.arange(10,dtype=np.int64); seg=np.array([1,2,2,3,3,3,4,4,4,4],np.int32)
    for rep in range(3):
        e,c=f.aggregate("SumAggregator",nid,seg,5)
        oe,oc=o.aggregate(X,"SumAggregator",nid,seg,5)
        print(D, rep, c, oc, np.array_equal(e,oe), e[:, :2].T.tolist())
