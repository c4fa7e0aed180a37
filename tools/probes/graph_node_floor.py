"""How long does a dependent chain of trivial kernel nodes take per node inside a captured graph (MI355X, ROCm 7)?
The floor any per-launch optimisation of the small levels runs into."""
import torch
x = torch.zeros(64, device="cuda")
s = torch.cuda.Stream()
for n in (200, 1000):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        x.add_(1.0)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
        for _ in range(3):
            g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
        s.synchronize()
    print(f"{n} dependent 64-element add_ nodes per graph: {e0.elapsed_time(e1) / 10 / n * 1e3:.2f} us per node")
