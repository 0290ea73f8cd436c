"""Stress: many tiny projcg solves vs the oracle; prints the number of mismatching calls."""
import sys, os; sys.path.insert(0,'.')
import numpy as np
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R, synth
from tests.helpers import DiagOpRef
lib = L.load_library(os.environ["LFPSQP_LIB"]) if "LFPSQP_LIB" in os.environ else None
ctx=L.Context(0, lib)
cases=[]
for n,m in ((1024,0),(1000,3),(2048,0),(3000,5),(1024,4),(512,0),(100,2),(6000,8)):
    Uh=np.asfortranarray(np.linalg.qr(synth.hash_matrix(1,n,m))[0]) if m else np.zeros((n,0),order='F')
    a=4*synth.hash_vector(3,n)+5; bh=synth.hash_vector(4,n)
    ref={}
    for k in (1,2,3,7,1000):
        x0=np.zeros(n); l0=np.zeros(m); i0,nr0=R.projcg_(x0,l0,DiagOpRef(a),Uh,bh,np.zeros(m),tol=1e-9,maxit=k); ref[k]=(x0,i0,nr0)
    cases.append((n,m,Uh,a,bh,ref))
bad=0; tot=0
reps=int(sys.argv[1]) if len(sys.argv)>1 else 20
for rep in range(reps):
    for n,m,Uh,a,bh,ref in cases:
        x=ctx.vector(n); lam=ctx.vector(max(m,1)); A=L.DiagOperator(0.0,ctx.vector(n,a)); U=L.DeviceBasis(ctx.matrix(n,m,Uh)); b=ctx.vector(n,bh)
        for k,(x0,i0,nr0) in ref.items():
            i1,nr1=L.projcg_(x,lam,A,U,b,None,tol=1e-9,maxit=k)
            e=np.linalg.norm(x.download()-x0)/np.linalg.norm(x0)
            tot+=1
            if not (i1==i0 and e<1e-12 and abs(nr1-nr0)<=1e-6*nr0):
                bad+=1
                if bad<6: print('MISMATCH',n,m,k,i0,i1,nr0,nr1,e)
print('bad',bad,'of',tot)
