import torch, time
dev="cuda:0"; n=112850; k=1200
def t(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e6
g = torch.cuda.CUDAGraph()
def bench(name, f):
    # replayed (launch overhead out)
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s)
    gg=torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        out=f()
    print(name, "eager %.1f us" % t(f), "replayed %.1f us" % t(gg.replay))
bench("randperm[:k]", lambda: torch.randperm(n, device=dev)[:k])
bench("rand.topk", lambda: torch.rand(n, device=dev).topk(k, sorted=False).indices)
bench("multinomial", lambda: torch.multinomial(torch.ones(n, device=dev), k, replacement=False))
bench("argsort rand[:k]", lambda: torch.rand(n, device=dev).argsort()[:k])
