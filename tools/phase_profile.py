import sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd, numpy as np
ctx=theta_amd.Context(0); r,rN,order=bench.synth()
p=theta_amd.Problem(ctx,3,50,2,r,rN,[0]*50,[6]*50,1.0)
tot=p.count
for rep in range(2):
    b=tot//3+rep*(tot//7)
    res=p.search(b,b+(1<<27),window=0.5); st=res['stats']
    pc=st['phase_cycles']; tw=pc[5]
    print('kernel_ms %.1f  C/s %.3g iters %.2f terms/it %.1f'%(st['kernel_ms'], st['evaluated']/st['kernel_ms']*1e3, st['iterations']/st['evaluated'], st['terms']/st['iterations']))
    print('  unrank %.1f%%  prefixes/wave %.1f leaves/prefix %.0f'%(100*pc[6]/tw, pc[7]/ (st['evaluated']/16384.0), st['evaluated']/max(pc[7],1)))
    print('  phases%%: group %.1f scan %.1f newton %.1f values %.1f successor %.1f  (cycles/cand %.0f)'%tuple([100*x/tw for x in pc[:5]]+[tw/st['evaluated']]))
