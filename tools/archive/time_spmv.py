import sys, time; sys.path.insert(0,'.')
import numpy as np, lfpsqp_jl_amd as L
ctx=L.Context(0); n,m,k=10_000_000,128,4
ii=np.arange(n,dtype=np.int64); rows=np.repeat(ii,k); cols=((((ii*m)//n)[:,None]+np.arange(k)[None,:])%m).ravel()
S=L.SparseMatrix(ctx,n,m,rows,cols,np.ones(n*k)); xs=ctx.vector(n).hash_fill(2); t=ctx.vector(m); y=ctx.vector(n)
for name,fn in (("spmv_t",lambda: L.spmv_t(S,xs,t)),("spmv_n",lambda: L.spmv_n(S,t,y))):
    fn(); ctx.timer_begin()
    for _ in range(20): fn()
    ms=ctx.timer_end()/20; print(name, ms, "ms", (12.0*S.nnz+8*n)/ms/1e6, "GB/s")
ref = np.zeros(m); 
print(np.abs(t.download()).max())
