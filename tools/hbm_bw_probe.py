import torch, time
x = torch.empty(256*1024*1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
for fn, name, bytes_ in [(lambda: y.copy_(x), "copy 1GiB->1GiB", 2*x.numel()*4), (lambda: y.zero_(), "fill 1GiB", x.numel()*4), (lambda: x.sum(), "read-reduce 1GiB", x.numel()*4), (lambda: torch.add(x, 1.0, out=y), "add scalar", 2*x.numel()*4)]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print("%-20s %.3f ms  %.2f TB/s" % (name, ms, bytes_/ms/1e9))
