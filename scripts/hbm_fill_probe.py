import torch, time
dev=torch.device("cuda",0)
for nbytes in (200_000_000, 365_000_000, 1_800_000_000):
    x=torch.empty(nbytes//4,dtype=torch.float32,device=dev)
    for _ in range(3): x.fill_(1.0)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): x.fill_(2.0)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"fill {nbytes/1e6:.0f} MB: {ms*1e3:.1f} us -> {nbytes/ms/1e6:.0f} GB/s")
    y=torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    print(f"copy {nbytes/1e6:.0f} MB: {ms*1e3:.1f} us -> {2*nbytes/ms/1e6:.0f} GB/s (r+w)")
