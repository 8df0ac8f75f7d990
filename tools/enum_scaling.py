import sys; sys.path.insert(0,".")
import torch, bench, theta_amd
ctx=theta_amd.Context(0)
r,rN,o=bench.synth(seed=11,m=50,n=3,k=6)
p=theta_amd.Problem(ctx,3,50,2,r,rN,[0]*50,[6]*50,1.0)
buf=torch.empty((1<<26)*100,dtype=torch.uint8,device="cuda:0")
b=(p.count)//3
for cnt in (64, 8192, 8192*64, 8192*1024, 8192*4096, 8192*8192):
    p.enumerate_device(b,cnt,buf.data_ptr())
    print(cnt, "tasks", max(1,cnt//8192), "ms %.3f"%min(p.enumerate_device(b,cnt,buf.data_ptr()) for _ in range(3)))
