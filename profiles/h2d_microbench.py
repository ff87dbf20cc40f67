"""H2D of a VGA RGB-D frame (921600 + 614400 bytes) from pinned memory: one copy on one stream (what the ingest ring does), the two images on two
streams, the frame cut in 2 / 4 pieces on 2 / 4 streams.  python profiles/h2d_microbench.py (GPU)"""
import time, torch
dev = torch.device("cuda:0")
N = 921600 + 614400
host = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(16)]
devb = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(16)]
def run(parts, reps=200):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cuts = [N * i // parts for i in range(parts + 1)]
    if parts == 2: cuts = [0, 921600, N]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        h, d = host[r % 16], devb[r % 16]
        for p, s in enumerate(streams):
            with torch.cuda.stream(s):
                d[cuts[p]:cuts[p + 1]].copy_(h[cuts[p]:cuts[p + 1]], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e6, N / dt / 1e9
for parts in (1, 2, 4, 1, 2, 4):
    us, gbs = run(parts)
    print("streams %d: %.1f us per frame, %.1f GB/s" % (parts, us, gbs))
