import torch
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    x=torch.zeros(1<<20,device='cuda')
    for _ in range(10): x.add_(1)
    torch.cuda.synchronize()
    ts=[]
    for _ in range(200):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(s); b.record(s); ts.append((a,b))
    torch.cuda.synchronize()
    import statistics
    print("empty pair us: median", statistics.median(a.elapsed_time(b)*1e3 for a,b in ts))
    ts=[]
    for _ in range(200):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(s); x.add_(1); b.record(s); ts.append((a,b))
    torch.cuda.synchronize()
    print("tiny kernel pair us: median", statistics.median(a.elapsed_time(b)*1e3 for a,b in ts))
