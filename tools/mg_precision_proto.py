#!/usr/bin/env python3
"""CPU prototype (numpy, no GPU): does the lattice multigrid-PCG survive LOWER-PRECISION STORAGE of the V-cycle's vectors?

    python tools/mg_precision_proto.py [N]          # N = 256 (seconds) ... 1024 (minutes)

Emulates the GPU solver's cycle -- 5-point Laplacian on the unit square, V(2,2) with the Chebyshev weights
(0.56, 1.39), P1 (7-point) transfers, CG with an fp32-stored search direction -- and rounds every vector the
cycle STORES (iterate after the first two sweeps, coarse right-hand side, iterate after each post-sweep) to
fp32 / fp16 / bf16, the residual being rescaled by a power of two per iteration where that matters.
Result that decided round 2: fp16 storage keeps the fp32 iteration count at 256^2 (12) but needs
17 instead of 12 at 1024^2 -- the rough 5e-4 storage noise swamps the h^2-small smooth residual the coarse grids must
see -- and bf16 needs 30; the halved cycle traffic does not pay for that, so the fp16 kernels were not built."""
import sys

import numpy as np
# numpy prototype of the lattice MG-PCG (5-point Laplacian on unit square, Dirichlet), V(2,2) Chebyshev-Jacobi,
# P1 (7-point) transfers, to test LOW-PRECISION STORAGE of the V-cycle vectors.
def rnd(x, mode):
    if mode == 'f64': return x
    if mode == 'f32': return x.astype(np.float32).astype(np.float64)
    if mode == 'f16': return x.astype(np.float16).astype(np.float64)
    if mode == 'bf16':
        a = x.astype(np.float32).view(np.uint32)
        a = ((a + 0x7FFF + ((a >> 16) & 1)) >> 16) << 16
        return a.astype(np.uint32).view(np.float32).astype(np.float64)
def A(u):  # u (N+1,N+1) with zero boundary; 5-point, diag 4
    r = np.zeros_like(u)
    r[1:-1,1:-1] = 4*u[1:-1,1:-1]-u[:-2,1:-1]-u[2:,1:-1]-u[1:-1,:-2]-u[1:-1,2:]
    return r
def restrict(r):  # P1 full weighting (7-point): centre 1, W,E,N,S 1/2, NE-of-row-above & SW-of-row-below 1/2 ; row index = y
    N = r.shape[0]-1; Nc=N//2
    rc = np.zeros((Nc+1,Nc+1))
    c = r[2:-1:2,2:-1:2]
    h = r[2:-1:2,1:-2:2]+r[2:-1:2,3::2]+r[1:-2:2,2:-1:2]+r[3::2,2:-1:2]+r[1:-2:2,3::2]+r[3::2,1:-2:2]
    rc[1:-1,1:-1] = c+0.5*h
    return rc
def prolong(e):
    Nc=e.shape[0]-1; N=2*Nc
    x=np.zeros((N+1,N+1))
    x[::2,::2]=e
    x[::2,1::2]=0.5*(e[:,:-1]+e[:,1:])
    x[1::2,::2]=0.5*(e[:-1,:]+e[1:,:])
    x[1::2,1::2]=0.5*(e[:-1,1:]+e[1:,:-1])   # quad diagonal b-d: (i,j+1) and (i+1,j)
    x[0,:]=x[-1,:]=0; x[:,0]=x[:,-1]=0
    return x
W=(0.56,1.39)
def vcycle(r, mode, lev=0):
    N=r.shape[0]-1
    if N<=2:
        x=np.zeros_like(r); x[1,1]=r[1,1]/4; return x
    # first two sweeps from zero
    x = W[0]*r/4
    x = x + W[1]*(r-A(x))/4
    x = rnd(x,mode)
    res = r-A(x)
    rc = rnd(restrict(res),mode)
    ec = vcycle(rc, mode, lev+1)
    x = x+prolong(ec)
    x = rnd(x + W[1]*(r-A(x))/4, mode)
    x = rnd(x + W[0]*(r-A(x))/4, mode)
    return x
def pcg(b, mode, tol=1e-12, maxit=40, scale_each=False):
    x=np.zeros_like(b); r=b.copy()
    def prec(r):
        s = 1.0/np.max(np.abs(r)) if scale_each else 1.0
        s = 2.0**np.floor(np.log2(s)) if scale_each else 1.0
        return vcycle(rnd(r*s,mode),mode)/s
    z=prec(r); p=z.copy(); rz=np.sum(r*z); bb=np.sqrt(np.sum(b*b)); hist=[]
    for it in range(maxit):
        p32=rnd(p,'f32')
        Ap=A(p32); al=rz/np.sum(p32*Ap)
        x+=al*p32; r-=al*Ap
        hist.append(np.sqrt(np.sum(r*r))/bb)
        if hist[-1]<tol: break
        z=prec(r); rzn=np.sum(r*z); be=rzn/rz; rz=rzn
        p=z+be*p32
    return x,hist
N=int(sys.argv[1]) if len(sys.argv)>1 else 256
rng=np.random.default_rng(0)
for name,b in (('f=1',np.pad(np.ones((N-1,N-1)),1)/N**2),('random',np.pad(1+0.5*rng.standard_normal((N-1,N-1)),1)/N**2)):
    for mode,se in (('f64',False),('f32',False),('bf16',False),('f16',True),('bf16',True)):
        x,h=pcg(b,mode,scale_each=se)
        print(f"N={N} {name:7s} storage {mode:5s} rescale-each-iter={se}: its {len(h)}  relres history "+" ".join(f"{v:.1e}" for v in h[:12]))
