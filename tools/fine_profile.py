import sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
ctx=theta_amd.Context(0); r,rN,order=bench.synth()
p=theta_amd.Problem(ctx,3,50,2,r,rN,[0]*50,[6]*50,1.0)
tot=p.count
for rep in range(2):
    b=tot//3+rep*(tot//7)
    res=p.search(b,b+(1<<27),window=0.5); st=res['stats']
    pc=st['phase_cycles']; tw=pc[5]; ev=st['evaluated']
    rounds=ev/256.0
    print('f64 evaluations per candidate %.4f'%(st['degenerate']/ev))
    print('kernel_ms %.1f cycles/cand %.0f | newton trips/round %.2f: refill %.0f + step %.0f of %.0f cycles/trip | scan %.0f values %.0f cycles/cand'%(
        st['kernel_ms'], tw/ev, pc[4]/rounds, pc[6]/max(pc[4],1), pc[0]/max(pc[4],1), pc[2]/max(pc[4],1), pc[1]/ev, pc[3]/ev))
