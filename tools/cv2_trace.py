#!/usr/bin/env python3
"""Diagnostic: in-kernel s_memtime stamps of conv3x3_v2 (needs `make -C unet_amd/csrc TRACE=1` and
UNET_HIP_LIB=unet_amd/libunet_hip_trace.so).  Prints per-segment cycle statistics for one layer."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from unet_amd._lib import lib
cin, cout, h, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0"); N = 32; dt = ops.BF16
x = ops.Act((torch.randn((N, h, h, cin), device=dev) * 0.5).bfloat16(), 0, cin)
y = ops.dense(N, h, h, cout, dt, dev)
w = torch.randn((cout, cin, 3, 3), device=dev) * (cin * 9) ** -0.5
wf = ops.pack_weights(w, 0, dt); bias = torch.zeros(cout, device=dev)
trace = torch.zeros(4096, dtype=torch.int64, device=dev)
run = (lambda: ops.conv_fwd(x, wf, bias, y, 3, 3, cin, cout, stats=True)) if mode == "fwd" else (lambda: ops.conv_fwd(x, wf, None, y, 3, 3, cin, cout))
for _ in range(3 if '--hot' not in sys.argv else 4000): run()      # --hot: ~2 s of back-to-back launches first
torch.cuda.synchronize()
L = lib(); L.unet_debug_set_trace.argtypes = [ctypes.c_void_p]
assert L.unet_debug_set_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
L.unet_debug_set_trace(None)
tc = trace.cpu()
ck = [int(v) for v in tc[2048:2052]]
if ck[3] > ck[1]:
    print(f"in-kernel clock: {(ck[2] - ck[0]) / (ck[3] - ck[1]) * 100:.0f} MHz over {ck[3] - ck[1]} ticks of the 100 MHz counter")
t = tc[:2048].view(2, 1024)[:, :1000].reshape(2, 500, 2)
names = {1: "phase start", 2: "mfma issued", 3: "dma waited", 4: "barrier passed", 5: "refill issued", 6: "epilogue done", 7: "first half mfma", 8: "late refill", 9: "epi: first group packed", 10: "epi: half the groups stored"}
for wv in range(2):
    ev = [(int(a), int(b)) for a, b in t[wv] if int(b) != 0]
    print(f"--- wave {wv*4}: {len(ev)} events; first 40 deltas (cycles @100MHz-tick? see s_memtime):")
    prev = ev[0][0]; seg = {}
    for i, (ts, tag) in enumerate(ev[1:]):
        d = ts - prev; prev = ts
        seg.setdefault(tag, []).append(d)
        if i < 40: print(f"   -> {names[tag]:16s} +{d}")
    for tag, v in sorted(seg.items()):
        v2 = v[4:] if len(v) > 8 else v
        print(f"   segment ending at '{names[tag]}': n={len(v)} mean={sum(v2)/len(v2):.0f} min={min(v2)} max={max(v2)}")
    print(f"   total span {ev[-1][0]-ev[0][0]} for {len(ev)} events")
