import torch, time
dev="cuda"
def t(fn, reps=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for MN,sk in ((4194304,2),(2097152,4),(1048576,8)):
    p=torch.randn(sk,MN,device=dev); out=torch.empty(MN,device=dev)
    us=t(lambda: torch.sum(p,0,out=out))
    print("torch.sum", MN, sk, round(us,1),"us", round((sk+1)*MN*4/us/1e6,2),"TB/s")
    a=torch.randn(MN,device=dev); b=torch.empty_like(a)
    us=t(lambda: torch.add(a,1.0,out=b))
    print("  copy-like", round(us,1),"us", round(2*MN*4/us/1e6,2),"TB/s")
big=torch.randn(64*1024*1024,device=dev); o=torch.empty_like(big)
us=t(lambda: torch.add(big,1.0,out=o),20); print("big 256MB add", round(us,1), round(2*big.numel()*4/us/1e6,2),"TB/s")
