import sys, os, torch
sys.path.insert(0, os.getcwd())
from unet_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
N, h = 32, 512
for cin, cout in ((64, 64), (128, 64)):
    x = ops.Act((torch.randn((N, h, h, cin), device=dev) * 0.5).bfloat16(), 0, cin)
    y = ops.dense(N, h, h, cout, ops.BF16, dev)
    w = torch.randn((cout, cin, 3, 3), device=dev) * (cin * 9) ** -0.5
    wf = ops.pack_weights(w, 0, ops.BF16)
    bias = torch.zeros(cout, device=dev)
    for rep in range(2):
        for name, b, st in (("plain", None, False), ("bias", bias, False), ("stats", None, True), ("bias+stats", bias, True)):
            t = timeit(lambda: ops.conv_fwd(x, wf, b, y, 3, 3, cin, cout, stats=st))
            print(f"{cin}->{cout} {name:11s} {t:.3f} ms", flush=True)
