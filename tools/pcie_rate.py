"""host-to-device and device-to-host copy rates of this box (pinned memory, one and three streams): the bound of any
host-fed path - a 1242 x 375 stereo pair is 0.93 MB"""
import time
import torch
n = 186 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for label, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"{label} one stream: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms per 186 MB)")
streams = [torch.cuda.Stream() for _ in range(3)]
parts = 24
step = n // parts
torch.cuda.synchronize()
t = time.perf_counter()
for r in range(5):
    for i in range(parts):
        with torch.cuda.stream(streams[i % 3]):
            d[i * step:(i + 1) * step].copy_(h[i * step:(i + 1) * step], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"H2D three streams, 24 pieces: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms per 186 MB)")
for nstreams, parts in ((1, 24), (1, 5), (1, 2), (2, 6), (3, 6)):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    step = n // parts
    torch.cuda.synchronize()
    t = time.perf_counter()
    for r in range(5):
        for i in range(parts):
            with torch.cuda.stream(streams[i % nstreams]):
                d[i * step:(i + 1) * step].copy_(h[i * step:(i + 1) * step], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"H2D {nstreams} stream(s), {parts} pieces: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms per 186 MB)")
