import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from ev2hands_amd import ops, synth
M, N, K = 4096, 256, 256
X = torch.from_numpy(synth.hash_normal("X", (M, K), 2)).float()
W = torch.from_numpy(synth.hash_normal("W", (N, K), 3) / np.sqrt(K)).float()
for prec in ("f16x2", "bf16x3"):
    for sx, sw in ((1, 1), (1e-2, 1), (1e-3, 1), (1e-4, 1), (1, 1e-2), (1, 1e-3), (1e-3, 1e-3), (1e3, 1), (3e4, 1)):
        x, w = (X * sx).cuda(), (W * sw).cuda()
        ref = x.double() @ w.double().t()
        got = ops.dense(x, w, None, False, precision=prec)
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        print(f"{prec} x*{sx:g} w*{sw:g}: rel err {err:.2e}")
