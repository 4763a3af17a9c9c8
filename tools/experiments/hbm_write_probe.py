"""HBM write / copy rates of plain torch kernels on this box (context for the adapters' write-heavy traffic)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 29  # 4 GiB of float64
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.0)); print(f"fill   {n*8/ms/1e6:8.1f} GB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy   {2*n*8/ms/1e6:8.1f} GB/s (read + write), {n*8/ms/1e6:.1f} written")
ms = t(lambda: x.sum()); print(f"sum    {n*8/ms/1e6:8.1f} GB/s read")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add    {2*n*8/ms/1e6:8.1f} GB/s (read + write)")
