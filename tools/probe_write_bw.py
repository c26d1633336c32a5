"""Dev tool: achievable HBM write bandwidth on this GPU (torch fill of a 100 GB buffer), the yardstick for the
write-once traceback records."""
import torch

n = 100 * (1 << 30)
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
for dt, name in ((torch.uint8, "u8"), (torch.int64, "i64")):
    v = buf.view(dt)
    v.fill_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        v.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"fill {name}: {n / 1e9:.0f} GB in {ms:.2f} ms = {n / ms / 1e9:.2f} TB/s", flush=True)
