"""achievable HBM rates for the shapes of the memory-bound layers: copy (read + write), fill (write), sum (read)"""
import torch, time
dev = "cuda:0"
n = 1708800 * 256   # layer1 expand output, fp32: 1.75 GB
x = torch.randn(n, device=dev); y = torch.empty_like(x); r = torch.randn(n, device=dev)
def t(f, nbytes, name, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - s) / it
    print(f"{name:28s}: {dt*1e3:7.3f} ms  {nbytes/dt/1e12:5.2f} TB/s", flush=True)
t(lambda: y.copy_(x), 2 * n * 4, "copy (1 read + 1 write)")
t(lambda: y.fill_(1.0), n * 4, "fill (write only)")
t(lambda: torch.add(x, r, out=y), 3 * n * 4, "add (2 reads + 1 write)")
t(lambda: x.sum(), n * 4, "sum (read only)")
