"""round 5: does enqueuing a 23 MB pinned host -> device copy block the calling thread?  (The staged leg spends 1.05 ms of
host time per step against 0.11 ms for device-fed steps, and on most boxes upload and step do not overlap.)  Times the
CALL of a non-blocking copy from pinned memory (a) on an idle stream, (b) on a stream that first has to wait for an event
behind ~3 ms of kernels on another stream, (c) back to back; and the same through libkvfe's staged step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dev = torch.device("cuda", 0)
n = 23101440
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device=dev)
a = torch.randn(4096, 4096, device=dev)
s_copy, s_work = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def call_ms(fn):
    t0 = time.perf_counter()
    fn()
    return round((time.perf_counter() - t0) * 1e3, 3)


with torch.cuda.stream(s_copy):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
idle = []
for _ in range(5):
    with torch.cuda.stream(s_copy):
        idle.append(call_ms(lambda: d.copy_(h, non_blocking=True)))
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    idle[-1] = (idle[-1], round((time.perf_counter() - t0) * 1e3, 3))
print("idle stream: (call ms, then wait ms):", idle)
res = []
for _ in range(5):
    with torch.cuda.stream(s_work):
        b = a
        for _ in range(12):
            b = b @ a
        ev = torch.cuda.Event()
        ev.record()
    with torch.cuda.stream(s_copy):
        s_copy.wait_event(ev)
        c1 = call_ms(lambda: d.copy_(h, non_blocking=True))
        c2 = call_ms(lambda: d.copy_(h, non_blocking=True))
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    res.append((c1, c2, round((time.perf_counter() - t0) * 1e3, 3)))
print("stream waiting for an event behind kernels: (call 1 ms, call 2 ms, then wait ms):", res)
