"""Timings of the tangent setup at full size: python tools/time_factorize.py N M  (best of 5)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
ctx=L.Context(0, L.load_library(os.environ['LFPSQP_LIB']) if 'LFPSQP_LIB' in os.environ else None)
n,m=int(float(sys.argv[1])),int(sys.argv[2])
J=ctx.matrix(n,m).hash_fill(1); Z=ctx.matrix(n,m)
best=[1e9,1e9,1e9]
for rep in range(6):
    ctx.sync(); t=time.perf_counter(); G=L.gram(J); t1=time.perf_counter()-t
    W=np.eye(m); t=time.perf_counter(); L.rmul(J,W,Z); ctx.sync(); t2=time.perf_counter()-t
    t=time.perf_counter(); S,Vt,r=L.ksvd_(J,Z); ctx.sync(); t3=time.perf_counter()-t
    if rep: best=[min(a,b) for a,b in zip(best,(t1,t2,t3))]
t1,t2,t3=best
print(f"n={n} m={m} gram {t1*1e3:.1f} ms ({2*n*m*m/t1/1e12:.1f} TF)  rmul {t2*1e3:.1f} ms ({2*n*m*m/t2/1e12:.1f} TF)  factorize total {t3*1e3:.1f} ms  rank {r} S[0]={S[0]:.3f} S[-1]={S[-1]:.3f}")
G2=L.gram(Z); print('orth err', np.abs(G2-np.eye(m)).max())
