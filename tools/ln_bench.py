import sys, torch
sys.path.insert(0, '/root/repo')
from animal2vec_amd import ops
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e3
for M in (13440, 16000, 53760, 64000):
    C=1024
    x=torch.randn(M,C,device='cuda').bfloat16(); r=torch.randn(M,C,device='cuda').bfloat16()
    w=torch.ones(C,device='cuda'); b=torch.zeros(C,device='cuda')
    y=torch.empty_like(x)
    us=timed(lambda: ops.layernorm_fwd(x,w,b,1e-5,out=y))
    us2=timed(lambda: ops.layernorm_fwd(x,w,b,1e-5,res=r,out=y))
    print(M, 'plain', round(us,1), 'us', round(M*C*2*2/us/1e6,2), 'TB/s | +res', round(us2,1), round(M*C*2*3/us2/1e6,2),'TB/s')
