"""Does data WRITTEN by one kernel come back from the 256 MiB memory-side cache when the NEXT kernel reads it?  (What a chunked
input-gradient -> weight-gradient schedule would rely on: dZ of a tile range read back before it has left the cache.)
   python tools/debug/mall_after_write.py     (GPU box)
For each size: write the buffer (torch fill: plain stores), optionally stream `gap` MB through another buffer, then time a read
(torch.sum over int32 view -> a pure streaming read) and a device copy; compared with the same read after a 2 GB flush."""
import torch
dev = "cuda:0"
flush = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)      # 2 GB
def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)
for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("after_write", "after_flush"):
        ts = []
        for _ in range(5):
            a.fill_(1.0)
            if mode == "after_flush":
                flush.fill_(0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s = a.sum(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = mb / 1024 / (min(ts) * 1e-3) / 1e3      # TB/s (GiB-ish)
    print(f"{mb:5d} MB: read right after the write {res['after_write']:.2f} TB/s, after a 2 GB flush {res['after_flush']:.2f} TB/s")
