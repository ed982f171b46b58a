import torch, time
x = torch.zeros(128, 128, dtype=torch.float64, device='cuda')
y = torch.zeros_like(x)
def body(n):
    for _ in range(n):
        torch.add(x, 1., out=y)
        torch.mul(y, 0.5, out=x)
body(10); torch.cuda.synchronize()
for n in (25, 125):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(n)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print('graph of %d kernel nodes: %.1f us per replay, %.2f us per node' % (2 * n, dt * 1e6, dt * 1e6 / (2 * n)))
t0 = time.perf_counter()
for _ in range(20): body(125)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print('eager %d kernels: %.2f us per kernel' % (250, dt * 1e6 / 250))
