import torch
dev = torch.device("cuda:0")
for mb in (39, 160, 512, 1007):
    n = mb * 1000 * 1000 // 8        # read n*4, write n*4  -> total mb MB
    sets = max(2, int(600e6 // (n * 8)) + 1)
    a = [torch.rand(n, device=dev) for _ in range(sets)]
    b = [torch.empty(n, device=dev) for _ in range(sets)]
    for i in range(5): b[i % sets].copy_(a[i % sets])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for i in range(reps): b[i % sets].copy_(a[i % sets])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"copy moving {mb} MB per launch: {us:.1f} us = {n * 8 / us / 1e6:.2f} TB/s ({n * 8 / us / 1e6 / 8:.2f} of peak)")
