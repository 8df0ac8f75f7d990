"""In-kernel phase breakdown of the n=3 search on the bench workload (three rank ranges, best of 4 launches each)."""
import sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd, numpy as np
M=int(sys.argv[1]) if len(sys.argv)>1 else 50
K=int(sys.argv[2]) if len(sys.argv)>2 else 6
ctx=theta_amd.Context(0); r,rN,order=bench.synth(m=M,n=3,k=K) if (M,K)!=(50,6) else bench.synth()
p=theta_amd.Problem(ctx,3,M,2,r,rN,[0]*M,[K]*M,1.0)
tot=p.count
tms=[]
run=float("inf")
for rep in range(3):      # the job's running minimum over these ranges (a search is one job: later pieces start from it)
    b=tot//3+rep*(tot//7)
    res=p.search(b,b+(1<<27),window=0.5)
    if len(res['nll']): run=min(run,float(res['nll'].min()))
for rep in range(3):
    b=tot//3+rep*(tot//7)
    best=None
    for it in range(4):
        if run<float("inf"): p.hint(run)
        res=p.search(b,b+(1<<27),window=0.5); st=res['stats']
        if best is None or st['kernel_ms']<best['kernel_ms']: best=st
    st=best; pc=st['phase_cycles']; tw=pc[5]
    tms.append(st['kernel_ms'])
    print('kernel_ms %.2f  C/s %.3g iters %.2f terms/it %.1f  | phases%%: group %.1f scan %.1f newton %.1f values %.1f unrank %.1f | cycles/cand %.0f'%(
        st['kernel_ms'], st['evaluated']/st['kernel_ms']*1e3, st['iterations']/st['evaluated'], st['terms']/st['iterations'],
        *[100*x/tw for x in (pc[0],pc[1],pc[2],pc[3],pc[6])], tw/st['evaluated']))
print('mean kernel_ms %.2f  prefixes/wave %.1f'%(np.mean(tms), pc[7]/(st['evaluated']/8192.0)))
