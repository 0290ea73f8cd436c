import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import lfpsqp_jl_amd as L
n, m = int(float(sys.argv[1])), int(sys.argv[2])
ctx = L.Context(0)
A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
W = np.zeros((m, m), order='F')
print("COLD", file=sys.stderr, flush=True)
S0, Vt0, r0 = L.ksvd_(A, None, W=W)
G = L.gram(A)
D = Vt0 @ G @ Vt0.T
print("offdiag of Vt G Vt' relative:", np.abs(D - np.diag(np.diag(D))).max() / np.abs(np.diag(D)).max(), "orth", np.abs(Vt0 @ Vt0.T - np.eye(m)).max(), "S range", S0.min(), S0.max(), file=sys.stderr, flush=True)
print("WARM", file=sys.stderr, flush=True)
S1, Vt1, r1 = L.ksvd_(A, None, W=W, Vt_prev=Vt0)
print("S diff", np.abs(S1 - S0).max() / S0.max(), file=sys.stderr)
