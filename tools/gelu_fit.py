import numpy as np
from scipy.special import erfc, erf
def fit(d, T, n=6000, lead_neg=False):
    t=np.linspace(0,T,n)
    y=np.log2(erfc(t/np.sqrt(2)))
    w=t*0.5*erfc(t/np.sqrt(2))*np.log(2)+1e-13
    A=np.vstack([t**k for k in range(1,d+1)]).T
    ww=w.copy()
    for it in range(400):
        c=np.linalg.lstsq(A*ww[:,None], y*ww, rcond=None)[0]
        e=np.abs((A@c-y)*w)
        ww=ww*(0.5+e/e.max()); ww/=ww.max()
    return c, e.max()
def fma32(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)
def gelu_exact(u): return 0.5*u*(1+erf(u/np.sqrt(2)))
def gelu_fast32(u, c, clamp=None):
    u=u.astype(np.float32); t=np.abs(u)
    if clamp: t2=np.minimum(t,np.float32(clamp))
    else: t2=t
    cc=np.concatenate([[np.float32(-1.0)], c.astype(np.float32)])
    p=np.full_like(t2, cc[-1])
    for k in range(len(cc)-2,-1,-1): p=fma32(p,t2,np.full_like(t2,cc[k]))
    with np.errstate(over='ignore'):
        e=np.exp2(p.astype(np.float64)).astype(np.float32)
    f=(np.float32(0.5)-e).astype(np.float32)
    hu=(np.float32(0.5)*u).astype(np.float32)
    return fma32(t,f,hu)
u=np.concatenate([np.linspace(-8,8,4000001), np.linspace(-300,300,600001)])
ex=gelu_exact(u)
# reference rounding floor: exact rounded to f32
floor=np.abs(ex.astype(np.float32).astype(np.float64)-ex)
print('f32 rounding floor max', floor.max())
for d,T,cl in ((5,6.0,None),(5,5.5,None),(6,6.0,6.0),(6,6.5,6.5),(7,6.5,6.5),(8,6.0,None),(8,7.0,None)):
    c,fe=fit(d,T)
    g=gelu_fast32(u,c,cl).astype(np.float64)
    err=np.abs(g-ex)
    small=np.abs(u)<=8
    print(d,T,cl,'fit',fe,'max abs err |u|<=8:',err[small].max(),'at',u[small][err[small].argmax()],' all:',err.max(),'at',u[err.argmax()],'lead',c[-1])
    if d in (5,6,8): print('   coeffs', ', '.join('%.9ef'%v for v in c))
