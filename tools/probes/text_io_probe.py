import time, numpy as np, sys, os
sys.path.insert(0,'/root/repo')
from city2ba_amd.baproblem import write_bal, read_bal
rng=np.random.default_rng(1)
n_cam,n_pts=170000,500000
counts=rng.integers(20,40,size=n_cam)
row=np.concatenate([[0],np.cumsum(counts)]).astype(np.uint64)
n=int(row[-1])
pt=rng.integers(0,n_pts,size=n).astype(np.uint64)
uv=rng.uniform(-1,1,size=(n,2))
bal9=rng.normal(size=(n_cam,9)); pts=rng.normal(size=(n_pts,3))*100
t=time.time(); write_bal('/tmp/t2.bal',bal9,pts,row,pt,uv); t1=time.time()-t
t=time.time(); out=read_bal('/tmp/t2.bal'); t2=time.time()-t
print("threads %s: write %.2f s, read %.2f s"%(os.environ.get("C2B_IO_THREADS"),t1,t2))
