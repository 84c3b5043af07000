"""Why do the resident set-abstraction kernels of enc.sa1 cost ~250 us at 16 windows of 8192 points when 128 windows cost 816 us?
Runs ONE kernel (layer 1 from raw feature rows, f16x2) back to back at several batch sizes:  python tools/debug/sa_small_grid.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

up = lambda x, m: (x + m - 1) // m * m  # noqa: E731
prec = os.environ.get("KBENCH_PREC", "f16x2")
Npts = int(os.environ.get("NPTS", "8192"))
for (C1, C2, C3, K, S) in [(64, 96, 128, 128, 512), (64, 64, 128, 64, 512), (32, 32, 64, 32, 512), (128, 128, 256, 64, 128)]:
    for B in (4, 16, 32, 64, 128):
        d = "cuda"
        g = torch.Generator(device="cuda").manual_seed(1)
        feat = torch.rand(B, Npts, 8, device=d, generator=g)
        feat[:, :, 4:] = 0
        pts4 = torch.rand(B, Npts, 4, device=d, generator=g) * 2 - 1
        ctr4 = torch.rand(B, S, 4, device=d, generator=g) * 2 - 1
        mode = os.environ.get("GIDX", "random")
        if mode == "random":
            gidx = torch.randint(0, Npts, (B, S, K), device=d, dtype=torch.int32, generator=g)
        else:  # neighbours close in memory (sorted clouds)
            base = torch.randint(0, Npts - 4 * K, (B, S, 1), device=d, dtype=torch.int32, generator=g)
            gidx = (base + torch.arange(K, device=d, dtype=torch.int32).view(1, 1, K) * 3).contiguous()
        W1x = torch.randn(C1, 4, device=d)
        W1f, b1 = torch.randn(C1, 4, device=d) * 0.5, torch.randn(C1, device=d)
        W2 = torch.randn(up(C2, 32), C1, device=d) * C1 ** -0.5
        b2 = torch.randn(up(C2, 32), device=d)
        W3 = torch.randn(C3, up(C2, 8), device=d) * C2 ** -0.5
        b3 = torch.randn(C3, device=d)
        # ops.sa_mlp_max packs the weight images per call (host work): warm once, then time the raw descriptor launch
        import ctypes as C
        from ev2hands_amd import _lib
        from ev2hands_amd.pack import NS_OF, sa_bf16_images, plane_unscale
        i2, i3, u2, u3 = sa_bf16_images(W2[:C2].cpu().double().numpy(), W3[:, :C2].cpu().double().numpy(), NS_OF[prec])
        i2, i3 = torch.from_numpy(i2).cuda(), torch.from_numpy(i3).cuda()
        out = torch.empty(B, S, C3, device=d)
        dd = _lib.SaDesc()
        dd.ldp, dd.pts4, dd.ctr4, dd.gidx = C1, pts4.data_ptr(), ctr4.data_ptr(), gidx.data_ptr()
        dd.W1x, dd.b2, dd.b3, dd.W2s, dd.W3s = W1x.data_ptr(), b2.data_ptr(), b3.data_ptr(), i2.data_ptr(), i3.data_ptr()
        dd.out, dd.ldo = out.data_ptr(), C3
        dd.B, dd.Npts, dd.S, dd.K, dd.C1, dd.C2, dd.C3, dd.precision = B, Npts, S, K, C1, C2, C3, _lib.PREC[prec]
        dd.w2_unscale, dd.w3_unscale = u2, u3
        dd.feat, dd.ldf, dd.W1f, dd.ldw1f, dd.b1, dd.nfeat = feat.data_ptr(), 8, W1f.data_ptr(), 4, b1.data_ptr(), 4
        if prec == "f16x2":
            dd.w1f_unscale = plane_unscale(W1f.cpu().double().numpy(), 2)
            dd.w1x_unscale = plane_unscale(W1x[:, :3].cpu().double().numpy(), 2)
        if os.environ.get("AMAX"):        # the f16x2 range record of the output: one atomicMax per group on amax[b]
            amax = torch.zeros(B, device=d, dtype=torch.int32)
            dd.out_amax = amax.data_ptr()
        L = _lib.lib()
        fn = lambda: _lib.check(L.ev2h_sa_mlp_max(C.byref(dd), _lib.stream_handle()), "sa")  # noqa: E731
        ms = timeit(fn, iters=20, warm=3)
        print(f"[{prec}] sa<{C1},{C2},{C3}> K={K} S={S} Npts={Npts} gidx={mode} amax={bool(os.environ.get('AMAX'))} B={B:4d}: {ms * 1e3:8.1f} us  ({ms * 1e3 / B:6.2f} us/window)", flush=True)
