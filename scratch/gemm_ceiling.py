import sys, numpy as np
sys.path.insert(0,'.')
from imageclust_amd import _lib
ctx=_lib.Context(0)
rng=np.random.default_rng(0)
for (B,H,cin,cout,k) in [(16,32,4096,4096,1),(64,32,1024,1024,1),(64,14,256,256,3),(256,14,256,256,3),(256,7,512,512,3)]:
    x=rng.standard_normal((B,H,H,cin)).astype(np.float32)
    w=(rng.standard_normal((cout,cin,k,k))*0.02).astype(np.float32)
    sc=np.ones(cout,np.float32); sh=np.zeros(cout,np.float32)
    ctx.prof_reset(); ctx.prof_enable(-1)
    for _ in range(3): y=ctx.conv2d_fused(x,w,sc,sh,1,k//2,None,True,_lib.PREC_BF16)
    q=ctx.prof_query(_lib.K_CONV)
    print("M=%d N=%d K=%d: %.1f us/launch, %.0f TFLOP/s"%(B*H*H,cout,cin*k*k,q['ms']*1e3/q['launches'], q['flops']/q['ms']/1e9))
