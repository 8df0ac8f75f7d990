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
    outer=max(pc[6],1)
    print('kernel_ms %.1f cycles/cand %.0f | outer iterations %d (%.1f cand each) | phase A: %.2f trips, %.0f cycles per outer; scan+eval total %.0f cycles per outer -> phase B %.0f'%(
        st['kernel_ms'], tw/ev, outer, ev/outer, pc[4]/outer, pc[0]/outer, pc[1]/outer, (pc[1]-pc[0])/outer))
