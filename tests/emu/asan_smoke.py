"""Sanitizer smoke of the C-ABI sources on the CPU emulator (see tests/emu/Makefile target `asan`): odd sizes\n(padding / tails), factorize, projcg, NR and ProjPenalty with bounds.  Test infrastructure only."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lfpsqp_jl_amd as L
from oracle import synth
lib=L.load_library('tests/emu/_build_asan/liblfpsqp_emu.so')
ctx=L.Context(0,lib)
# odd sizes to stress padding / tails
for n,m in ((1,1),(1023,3),(2049,5),(4097,130),(77,4),(1200,300)):   # incl. the one-pass kernels: narrow (m >= 4) and wide (m > 256)
    Mh=synth.hash_matrix(1,n,m); J=ctx.matrix(n,m,Mh); Z=ctx.matrix(n,m)
    S,Vt,r=L.ksvd_(J,Z)
    x,lam=ctx.vector(n),ctx.vector(m)
    it,nr=L.projcg_(x,lam,L.DiagOperator(0.0,ctx.vector(n).hash_fill(3,0,4.0,5.0)),L.DeviceBasis(Z),ctx.vector(n).hash_fill(4),None,tol=1e-10,maxit=50)
    print(n,m,r,it,nr)
# bounds path
n,m=301,4
P0=synth.BallBoxProblem(n,m)
Jct=ctx.matrix(n+1,m+1).hash_fill(1,0,n,1.0,n,m)
P=L.QuadLinearBallBox(ctx,n,m,Jct,P0.eq.b,R2=P0.R2,xl=P0.xl,xu=P0.xu)
for dpr in (False,True):
    out=P.optimize(0.97*synth.hash_vector(2,n)+0.015,L.LFPSQPParams(do_project_retract=dpr,disp=L.DisplayOption.off,maxiter=2))
    print('opt',dpr,out[3].iter)
print("ASAN RUN DONE")
# bounds + ball + sparse twin (Jsp): the factored stacked basis with a dense extra column (round-2 advisor finding: its scratch n-vector)
import scipy.sparse as sp
Jh=Jct.download()[:n+1,:m]
Jsp=L.SparseMatrix.from_scipy(ctx,sp.csr_matrix(Jh))
P=L.QuadLinearBallBox(ctx,n,m,Jct,P0.eq.b,R2=P0.R2,xl=P0.xl,xu=P0.xu,Jsp=Jsp)
for dpr in (False,True):
    out=P.optimize(0.97*synth.hash_vector(2,n)+0.015,L.LFPSQPParams(do_project_retract=dpr,disp=L.DisplayOption.off,maxiter=2))
    print('opt sparse twin',dpr,out[3].iter)
G=Jsp.gram(Jct,ctx.vector(n+1).hash_fill(9,0,0.5,1.0))           # weighted Gram with a dense extra column (its scratch vector)
# the nonlinear (elementwise) constraint class: sin system on the nonzeros, sphere system dense, through optimize
for cons in (L.sin_system_constraints(ctx,301,7),):
    prob=L.SeparableElementwiseBox(ctx,cons,0,1.0,0.3*synth.hash_vector(5,301),xl=-np.ones(301),xu=np.ones(301))
    for dpr in (False,True):
        out=prob.optimize(np.zeros(301),L.LFPSQPParams(do_project_retract=dpr,disp=L.DisplayOption.off,maxiter=2))
        print('opt elementwise',dpr,out[3].iter)
print("ASAN RUN 2 DONE")
# round 3: staged stores with many short bursts and a ragged last round (narrow, wide and stacked forms), the sparse Gram kernels for rows
# of 9 .. 16 nonzeros over several column boxes, and the integer fold of many row slices
os.environ["LFPSQP_STAGE_ROUNDS"]="2"
ctx2=L.Context(0,lib)
for n,m in ((5003,20),(4999,130),(3001,300)):
    Z=ctx2.matrix(n,m).hash_fill(1,0,n,0.02)
    x=ctx2.vector(n)
    it,nr=L.projcg_(x,None,L.DiagOperator(0.0,ctx2.vector(n).hash_fill(3,0,4.0,5.0)),L.DeviceBasis(Z),ctx2.vector(n).hash_fill(4),None,tol=1e-300,maxit=5,want_lambda=False)
    print('staged',n,m,it,nr)
n,m=1201,6
P0=synth.BallBoxProblem(n,m)
Jct=ctx2.matrix(n+1,m+1).hash_fill(1,0,n,1.0,n,m)
P=L.QuadLinearBallBox(ctx2,n,m,Jct,P0.eq.b,R2=P0.R2,xl=P0.xl,xu=P0.xu)
for dpr in (False,True):
    out=P.optimize(0.97*synth.hash_vector(2,n)+0.015,L.LFPSQPParams(do_project_retract=dpr,disp=L.DisplayOption.off,maxiter=2))
    print('staged stacked opt',dpr,out[3].iter)
del os.environ["LFPSQP_STAGE_ROUNDS"]
for n,m,k in ((9001,70,12),(8200,150,16),(5000,40,9)):
    rows=np.repeat(np.arange(n),k); cols=((((np.arange(n)*m)//n)[:,None]+np.arange(k)[None,:])%m).ravel()
    vals=np.random.default_rng(k).standard_normal(n*k)
    S=L.SparseMatrix(ctx,n,m,rows,cols,vals)
    G=S.gram(None,ctx.vector(n).hash_fill(9,0,0.5,1.0))
    print('sparse gram',n,m,k,float(np.abs(G).max()))
os.environ["LFPSQP_SPGRAM_SLICES"]="64"
n,m,k=4096*66+5,24,10
rows=np.repeat(np.arange(n),k); cols=((((np.arange(n)*m)//n)[:,None]+np.arange(k)[None,:])%m).ravel()
S=L.SparseMatrix(ctx,n,m,rows,cols,np.random.default_rng(1).standard_normal(n*k))
print('sparse gram fold',float(np.abs(S.gram()).max()))
del os.environ["LFPSQP_SPGRAM_SLICES"]
print("ASAN RUN 3 DONE")
# the nonlinear class with a DENSE A: one-pass Newton step (generator known) and c! as the same launch, with and without bounds, with the common quadratic term
rng=np.random.default_rng(4)
n,m=1203,9
Ad=np.asfortranarray(rng.standard_normal((n,m))/np.sqrt(n))
kd=rng.integers(0,3,n).astype(np.float64)
consd=L.ElementwiseConstraints(ctx,ctx.matrix(n,m,Ad),0.1*rng.standard_normal(m),kind=kd,qw=0.01*rng.standard_normal(m))
for bounds in (False,True):
    kw=dict(xl=-2*np.ones(n),xu=2*np.ones(n)) if bounds else {}
    prob=L.SeparableElementwiseBox(ctx,consd,0,1.0,0.3*synth.hash_vector(5,n),**kw)
    out=prob.optimize(0.1*synth.hash_vector(6,n),L.LFPSQPParams(do_project_retract=False,disp=L.DisplayOption.off,maxiter=2))
    print('opt elementwise dense',bounds,out[3].iter)
print("ASAN RUN 4 DONE")
# round 5: tridiagonal / low-rank one-pass solvers (sizes at tile and padding multiples: the neighbour loads at the ends), the chain-objective class
# through optimize (tangent step + START_GIVEN + tridiagonal solver), the wide matrix-core batched Newton step
for n,m,fac in ((1024,128,False),(2048,33,True),(300,260,False),(64,4,False)):
    Jh=synth.hash_matrix(5,n,m); J=ctx.matrix(n,m,np.asfortranarray(Jh)); Z=ctx.matrix(n,m); W=np.zeros((m,m),order='F')
    S,Vt,r=L.ksvd_(J,Z,W=W)
    U=L.DeviceBasis(None,r,generator=(J,W)) if fac else L.DeviceBasis(Z)
    a=ctx.vector(n).hash_fill(3,0,4.0,6.0); off=ctx.vector(n).hash_fill(15,0,0.9,0.0); b=ctx.vector(n).hash_fill(4)
    x,lam=ctx.vector(n),ctx.vector(m)
    it,nr=L.projcg_(x,lam,L.TridiagonalOperator(0.0,a,off),U,b,None,tol=1e-9,maxit=30)
    V=ctx.matrix(n,3).hash_fill(17,0,n,n**-0.5)
    it2,nr2=L.projcg_(x,lam,L.LowRankOperator(0.0,a,V,3,np.array([3.0,-0.4,1.5])),U,b,None,tol=1e-9,maxit=30)
    print('tridiagonal / low rank',n,m,fac,it,nr,it2,nr2)
n,m=515,5
P0=synth.BallBoxProblem(n,m)
Pc=L.ChainSeparableLinear(ctx,n,m,ctx.matrix(n,m).hash_fill(1),P0.eq.b,1,0.5+synth.hash_vector(21,n)**2,0.3*synth.hash_vector(22,n),kappa=1.7)
out=Pc.optimize(synth.hash_vector(2,n),L.LFPSQPParams(do_project_retract=False,disp=L.DisplayOption.off,maxiter=3,tn_kappa=1e-6))
print('opt chain',out[3].iter)
n,m=300,140
P0=synth.BallBoxProblem(n,m)
Jct=ctx.matrix(n+1,m+1).hash_fill(1,0,n,1.0,n,m)
Pw=L.QuadLinearBallBox(ctx,n,m,Jct,P0.eq.b,R2=P0.R2,xl=P0.xl,xu=P0.xu)
ctx.options.ls_batch=8
out=Pw.optimize(P0.x0,L.LFPSQPParams(do_project_retract=False,disp=L.DisplayOption.off,maxiter=2,maxiter_retract=20))
ctx.options.ls_batch=0
print('opt wide batched',out[3].iter)
print("ASAN RUN 5 DONE")
# round 6: the EXACT batch of Newton retractions (virtual spans: several spans per workgroup through the test hook, running sums in LDS; narrow, exact
# and wide tiles, with and without bounds) through the unit cases of tests/test_exact_batch.py -- a whole failing linesearch under the sanitizers takes
# the better part of an hour per shape --, and the staged tangent step with short bursts
import tests.test_exact_batch as T6
os.environ["LFPSQP_NRB_WG_CAP"]="1"; os.environ["LFPSQP_STAGE_ROUNDS"]="2"
ctx3=L.Context(0,lib)
del os.environ["LFPSQP_NRB_WG_CAP"]; del os.environ["LFPSQP_STAGE_ROUNDS"]
for nb,bounds,mcols in ((4,True,0),(3,False,129),(4,True,129),(4,True,300),(2,False,300)):
    T6._bit_identical(ctx3,nb,bounds,mcols)
    print('exact batch',nb,bounds,mcols,flush=True)
rng=np.random.default_rng(9)
for n,m in ((703,40),(401,128)):
    Ah=np.asfortranarray(rng.standard_normal((n,m))/np.sqrt(n)); x0h=0.2*rng.standard_normal(n)
    prob=L.SeparableLinearBallBox(ctx3,n,m,ctx3.matrix(n,m,Ah),Ah.T@x0h,1,0.3,0.5*rng.standard_normal(n))
    out=prob.optimize(x0h,L.LFPSQPParams(do_project_retract=False,disp=L.DisplayOption.off,maxiter=3))
    print('opt staged tangent step',n,m,out[3].iter,flush=True)
print("ASAN RUN 6 DONE")
