"""Host -> device copy rate of this box from pinned memory (torch is only the
measuring stick here): one stream, 64 MB pieces, and two streams side by side."""
import time
import torch
n = 64 << 20
src = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(4)]
torch.cuda.synchronize()
for streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(64):
            with torch.cuda.stream(ss[i % streams]):
                dst[i % 4].copy_(src[i % 4], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'{streams} stream(s): {64 * n / dt / 1e9:.1f} GB/s')
big = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dbig = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
torch.cuda.synchronize()
t0 = time.perf_counter()
dbig.copy_(big, non_blocking=True)
torch.cuda.synchronize()
print(f'one 1 GiB copy: {(1 << 30) / (time.perf_counter() - t0) / 1e9:.1f} GB/s')
