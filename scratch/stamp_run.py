import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
from imageclust_amd import _lib
ctx=_lib.Context(0)
rng=np.random.default_rng(0)
n,d=10000,2048
cen=rng.standard_normal((500,d)).astype(np.float32)
E=(cen[rng.integers(0,500,n)]+0.1*rng.standard_normal((n,d))).astype(np.float32)
cid,rank,nc=ctx.cluster(E,5,50)
L=_lib.load()
out=(C.c_ulonglong*32)()
print('rc',L.icl_debug_stamps(out))
t=[out[i] for i in range(8)]
print('stamps (cycles rel. to start):',[ (x-t[0]) for x in t[:7]])
