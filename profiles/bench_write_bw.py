import torch
for nbytes in (560e6, 1.12e9, 2.24e9):
    t = torch.empty(int(nbytes // 4), device="cuda", dtype=torch.float32)
    for _ in range(3): t.fill_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): t.fill_(1.0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"fill {nbytes/1e9:.2f} GB: {ms:.3f} ms = {nbytes/ms/1e9:.2f} TB/s")
    s = torch.empty_like(t)
    for _ in range(3): s.copy_(t)
    a.record()
    for _ in range(10): s.copy_(t)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"copy {nbytes/1e9:.2f} GB: {ms:.3f} ms = {2*nbytes/ms/1e9:.2f} TB/s (read+write)")
    del t, s
