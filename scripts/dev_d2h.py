import time, torch
n = 45 * 1024 * 1024
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
for _ in range(3):
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print(f"pinned D2H {n/1e6:.0f} MB: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
hp = torch.empty(n, dtype=torch.uint8)
t = time.perf_counter()
for _ in range(5):
    hp.copy_(d)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"pageable D2H: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
