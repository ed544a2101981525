"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this GPU (MI355X_MICROARCH.md §HBM):
a 512 MiB float4 copy (streaming read + streaming write) and a 512 MiB row gather of 400-byte rows."""
import torch
n = 128 * 1024 * 1024
x = torch.empty(n, dtype=torch.float32, device='cuda').normal_()
torch.cuda.synchronize()
for _ in range(3):
    y = x.clone()                       # reads 512 MiB, writes 512 MiB
torch.cuda.synchronize()
rows = torch.randint(0, n // 100, (n // 100,), device='cuda')
t = x.view(-1, 100)
for _ in range(3):
    z = t.index_select(0, rows)         # gathers 400-byte rows: reads ~512 MiB (+ indices), writes 512 MiB
torch.cuda.synchronize()
