"""CPU check of the bound behind n3_sieve.hip: sv_prefix_beyond -- for random prefixes of a small seeded instance, the relaxed lower bound
(likelihood of the prefix alone + the constant of the freed intervals) against the minimum over sampled completions (scipy).  The bound
must lie below every completion's minimum.  python tools/prefix_bound_check.py"""
import numpy as np, itertools, sys
from scipy.optimize import minimize
rng=np.random.RandomState(5)
m=12; ML=3; K=4
L=rng.randint(2_000_000,20_000_000,m)
rN=np.maximum(rng.poisson(L*0.01),1).astype(float)
# true model
Ct=rng.randint(0,K+1,(m,2)); mu=np.array([0.3,0.45,0.25])
p=rN*(2*mu[0]+Ct@mu[1:]); p/=p.sum()
r=rng.multinomial(int(rN.sum()*1.1),p).astype(float)
N=rN.sum(); Nn=rN/N; Rtot=r.sum()
K0=-(r*np.log(Nn)).sum()
tau=2.0
def nll_w(w,C):   # C rows (x,y); homogeneous c=(tau? ...)
    # kernel form: q_i = w0 + x u1 + y u2 with normal copy = 1 unit ; z=(1,s1,s2)
    q=w[0]+C[:,0]*w[1]+C[:,1]*w[2]
    z=np.array([1.0,(Nn*C[:,0]).sum(),(Nn*C[:,1]).sum()])
    if (q<=0).any() or z@w<=0: return 1e300
    return K0-(r*np.log(q)).sum()+Rtot*np.log(z@w)
def minimize_nll(C):
    best=1e300
    for s in ([1,.3,.3],[1,.1,.5],[1,.5,.1]):
        res=minimize(lambda v: nll_w(np.array([1.0,np.exp(v[0]),np.exp(v[1])]),C),np.log(s[1:]),method='Nelder-Mead',options={'xatol':1e-10,'fatol':1e-10,'maxiter':4000})
        best=min(best,res.fun)
    # also allow negative u via direct param
    res=minimize(lambda v: nll_w(np.array([1.0,v[0],v[1]]),C),[0.2,0.2],method='Nelder-Mead',options={'xatol':1e-12,'fatol':1e-12,'maxiter':8000})
    return min(best,res.fun)
D=m-ML
for trial in range(6):
    pre=rng.randint(0,K+1,(D,2)); pre[0]=[1,2]; pre[1]=[2,0]
    # lower bound: reduced problem
    Rp=r[:D].sum(); Nrem=Nn[D:].sum()
    zred=np.array([1-Nrem,(Nn[:D]*pre[:,0]).sum(),(Nn[:D]*pre[:,1]).sum()])
    def F(w):
        q=w[0]+pre[:,0]*w[1]+pre[:,1]*w[2]
        if (q<=0).any() or zred@w<=0: return 1e300
        return K0-(r[:D]*np.log(q)).sum()+Rp*np.log(zred@w)
    fm=min(minimize(lambda v:F(np.array([1.0,v[0],v[1]])),s,method='Nelder-Mead',options={'xatol':1e-12,'fatol':1e-12,'maxiter':8000}).fun for s in ([.2,.2],[.05,.5],[.5,.05]))
    const=Rp*np.log(Rtot/Rp)-sum(r[l]*np.log(r[l]/(Rtot*Nn[l])) for l in range(D,m) if r[l]>0)
    LB=fm+const
    mins=[]
    for rows in itertools.product(range(K+1),repeat=2*ML):
        C=np.vstack([pre,np.array(rows).reshape(ML,2)])
        if (C[:,0].sum()==0) or (C[:,1].sum()==0): continue
        mins.append(minimize_nll(C)) if rng.rand()<0.02 else None
    mn=min(mins)
    print("trial",trial,"LB",LB,"min over sampled children",mn,"gap",mn-LB, "n",len(mins))
    assert LB<=mn+1e-6
