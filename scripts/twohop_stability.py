"""Why the resident forward solve does not hop two block rows at a time: eliminating x_{b-1} from the recurrence
(x_b from x_{b-2}, x_{b-3}, x_{b-4} through U_b = tf_b tf_{b-1} - tf2_b, ... -- block cyclic reduction of the substitution) halves the number
of dependent hops and is UNSTABLE on the headline kernel: relative error 6e-13 at N = 2 048, 1e-6 at N = 4 096 (ExpSquared, l = 2.5), against
1e-14 for the form the kernel uses.  Host NumPy, fp64.  python scripts/twohop_stability.py"""
import numpy as np, scipy.linalg as sla, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from tinygp_amd import synthetic
def run(n, ell, kind):
    X, y = synthetic.make_inputs(n, 1, "float64")
    x = np.asarray(X).reshape(-1)
    r = np.abs(x[:,None]-x[None,:])
    if kind=="expsq": K = 1.5**2*np.exp(-0.5*(r/ell)**2)
    else:
        a=np.sqrt(5)*r/ell; K=1.5**2*(1+a+a*a/3)*np.exp(-a)
    K[np.diag_indices(n)] += 0.01
    L = np.linalg.cholesky(K)
    ref = sla.solve_triangular(L, y, lower=True)
    nb = n//128
    blk = lambda i,j: L[i*128:(i+1)*128, j*128:(j+1)*128]
    W = [np.linalg.inv(blk(b,b)) for b in range(nb)]
    tf = [None]+[W[b]@blk(b,b-1) for b in range(1,nb)]
    tf2 = [None,None]+[W[b]@blk(b,b-2) for b in range(2,nb)]
    tf3 = [None]*3+[W[b]@blk(b,b-3) for b in range(3,nb)]
    yb = lambda b: y[b*128:(b+1)*128]
    # (b) current: x_b = W_b(y_b - sum_{c<=b-3}) - tf2 x_{b-2} - tf x_{b-1}
    xs=[]
    for b in range(nb):
        s = yb(b).copy()
        for c in range(0,b-2): s -= blk(b,c)@xs[c]
        v = W[b]@s
        if b>=2: v -= tf2[b]@xs[b-2]
        if b>=1: v -= tf[b]@xs[b-1]
        xs.append(v)
    xb_=np.concatenate(xs)
    # (c) two-hop: x_b = [W_b(y_b - S_b) - T_b(y_{b-1} - S_{b-1})] + U x_{b-2} + V x_{b-3} + Z x_{b-4}; S_b over c<=b-4, S_{b-1} over c<=b-5
    xs=[]
    for b in range(nb):
        if b < 4:
            s = yb(b).copy()
            for c in range(0,b): s -= blk(b,c)@xs[c]
            xs.append(W[b]@s); continue
        sb = yb(b).copy()
        for c in range(0,b-3): sb -= blk(b,c)@xs[c]
        sb1 = yb(b-1).copy()
        for c in range(0,b-4): sb1 -= blk(b-1,c)@xs[c]
        T = tf[b]@W[b-1]
        U = tf[b]@tf[b-1] - tf2[b]
        V = tf[b]@tf2[b-1] - tf3[b]
        Z = tf[b]@tf3[b-1]
        xs.append(W[b]@sb - T@sb1 + U@xs[b-2] + V@xs[b-3] + Z@xs[b-4])
    xc_=np.concatenate(xs)
    sc=np.abs(ref).max()
    print(n, kind, ell, "cond(K)=%.1e"%np.linalg.cond(K), " current err %.2e   two-hop err %.2e   |U|max %.1e |tf|max %.1e"%(np.abs(xb_-ref).max()/sc, np.abs(xc_-ref).max()/sc, max(np.abs(tf[b]@tf[b-1]-tf2[b]).max() for b in range(2,nb)), max(np.abs(t).max() for t in tf[1:])))
for n,ell,kind in ((2048,2.5,"expsq"),(4096,2.5,"expsq"),(4096,2.5,"m52"),(4096,10.0,"expsq"),(4096,0.5,"expsq")):
    run(n,ell,kind)
