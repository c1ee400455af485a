"""Host/GPU cost of one torch.distributed collective per iteration on a single-rank RCCL communicator (diagnostic)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.zeros(4, dtype=torch.float64, device="cuda")
a = torch.zeros(300000, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
x = torch.zeros(1 << 20, device="cuda")
def loop(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    return host / n * 1e6, tot / n * 1e6
print("tiny kernel only              host %.1f us  total %.1f us" % loop(lambda: x.add_(1.0)))
print("all_reduce(32 B)              host %.1f us  total %.1f us" % loop(lambda: dist.all_reduce(t)))
print("kernel + all_reduce           host %.1f us  total %.1f us" % loop(lambda: (x.add_(1.0), dist.all_reduce(t))))
print("all_to_all_single(1.2 MB)     host %.1f us  total %.1f us" % loop(lambda: dist.all_to_all_single(b, a, [300000], [300000])))
print("kernel + a2a + kernel + ar    host %.1f us  total %.1f us" % loop(lambda: (x.add_(1.0), dist.all_to_all_single(b, a, [300000], [300000]), x.add_(1.0), dist.all_reduce(t))))
dist.destroy_process_group()
