import sys, torch
sys.path.insert(0,'/root/repo')
from animal2vec_amd import ops
def timed(fn, iters=10, groups=7):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(groups):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b)/iters*1e3)
    ts.sort()
    return ts[len(ts)//2]
M,N,K=53760,1024,1024
import os
for kv in os.environ.get('A2V_DBG','').split(','):
    if kv: ops.debug_set(*map(int,kv.split('=')))
ONLY=os.environ.get('A2V_ONLY','')
if len(sys.argv)>1: M,N,K=map(int,sys.argv[1].split(','))
x=torch.randn(M,K,device='cuda').bfloat16(); w=(torch.randn(N,K,device='cuda')*K**-0.5).bfloat16()
b=torch.randn(N,device='cuda'); r=torch.randn(M,N,device='cuda').bfloat16(); out=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
cs=torch.zeros(N,device='cuda'); pre=torch.empty_like(out)
for name,kw in [('plain',{}),('plain again',{}),('res',dict(res=r)),('drop',dict(drop=(0.1,5))),('res+drop',dict(res=r,drop=(0.1,5))),('gelu',dict(gelu=True)),('gelu+pre',dict(gelu=True,pre_out=pre)),('gelu_bwd',dict(gelu_bwd_aux=r)),('colsum',dict(colsum=cs)),('res inplace',dict(res=out))]:
    if ONLY and name not in ONLY.split(','): continue
    bias = None if 'gelu_bwd_aux' in kw else b
    try:
        us=timed(lambda: ops.linear(x,w,bias,out=out,**kw))
        print(f"{name:12s} {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TF", flush=True)
    except Exception as e: print(name,'ERR',e)
