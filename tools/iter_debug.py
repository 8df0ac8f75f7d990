import sys,os
sys.path.insert(0, ".")
import bench, theta_amd
ctx=theta_amd.Context(0); r,rN,order=bench.synth()
p=theta_amd.Problem(ctx,3,50,2,r,rN,[0]*50,[6]*50,1.0)
tot=p.count; batch=1<<27; n=10; stride=(tot-batch)//n
run=float("inf")
for i in range(n):
    b=i*stride
    if run<float("inf"): p.hint(run)
    res=p.search(b,b+batch,window=0.5); st=res['stats']; pc=st['phase_cycles']
    if len(res['nll']): run=min(run,float(res['nll'].min()))
    print(i,"ms %.2f iters %.3f dismissed %.4f | slow(>=8 iters) %d with %d iterations, failed %d | suspects %d"%(st['kernel_ms'], st['iterations']/st['evaluated'], st['dismissed']/st['evaluated'], pc[0], pc[4], pc[6], len(p.last_suspects[0])))
