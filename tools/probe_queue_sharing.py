"""Does a kernel on stream B start while a spinning kernel occupies stream A, for every pair of
streams of one process (more streams than hardware queues)?  One process, one GPU.
The spinning kernel is the library's own bounded stripe_wait (through a tiny striped fill is
overkill), so this uses torch only: A runs torch._sleep-like busy kernel via a long matmul chain."""
import time
import torch

torch.cuda.init()
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream() for _ in range(10)]
flag = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.ones(1 << 20, device=dev)
big = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
# calibrate a ~50 ms kernel chain on one stream
def busy(s, reps):
    with torch.cuda.stream(s):
        y = big
        for _ in range(reps):
            y = y @ big
        return y
t0 = time.perf_counter(); busy(streams[0], 20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"busy chain 20 matmuls: {dt*1e3:.1f} ms", flush=True)
for a in range(0, 10):
    for b in range(0, 10):
        if a == b:
            continue
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        busy(streams[a], 20)
        with torch.cuda.stream(streams[b]):
            e0.record(); x.add_(1.0); e1.record()
        e1.synchronize()
        t_small = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        if t_small > 0.5 * t_all:
            print(f"stream {b} waited behind stream {a}: small kernel done after {t_small*1e3:.1f} ms of {t_all*1e3:.1f} ms", flush=True)
print("done", flush=True)
