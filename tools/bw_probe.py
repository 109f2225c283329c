import torch, time
dev=torch.device('cuda:0')
n=1<<30
a=torch.empty(n,dtype=torch.uint8,device=dev); b=torch.empty(n,dtype=torch.uint8,device=dev)
for name,fn,bytes_ in (("fill 1GiB",lambda:a.fill_(3),n),("copy 1GiB",lambda:b.copy_(a),2*n),("read-sum", lambda:a.view(torch.int32).sum(), n)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, f"{dt*1e3:.3f} ms  {bytes_/dt/1e12:.2f} TB/s")
