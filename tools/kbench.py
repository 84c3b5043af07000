"""Micro-benchmarks of the two MFMA kernels at the bench shapes (B=256, N=2048): python tools/kbench.py [sa|sab|gemm|gemmb|mano|all]  (the next-row benches that compare with the CPU oracles are tests/bench_next_rows.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2hands_amd import ops  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_sa(B=256, precision="f32"):
    up = lambda x, m: (x + m - 1) // m * m
    for (C1, C2, C3, K, S, Npts, tag) in [(32, 32, 64, 32, 512, 2048, "enc.sa1.0"), (64, 64, 128, 64, 512, 2048, "enc.sa1.1"),
                                          (64, 96, 128, 128, 512, 2048, "enc.sa1.2"), (128, 128, 256, 64, 128, 2048, "mano.0"),
                                          (128, 196, 256, 128, 128, 2048, "mano.1"), (128, 196, 256, 128, 128, 512, "enc.sa2.1")]:
        d = "cuda"
        P1 = torch.randn(B, Npts, C1, device=d)
        pts4 = torch.randn(B, Npts, 4, device=d)
        ctr4 = torch.randn(B, S, 4, device=d)
        gidx = torch.randint(0, Npts, (B, S, K), device=d, dtype=torch.int32)
        W1x = torch.randn(C1, 4, device=d)
        W2 = torch.randn(up(C2, 32), C1, device=d) * C1 ** -0.5
        b2 = torch.randn(up(C2, 32), device=d)
        W3 = torch.randn(C3, up(C2, 8), device=d) * C2 ** -0.5
        b3 = torch.randn(C3, device=d)
        if os.environ.get("KBENCH_ZERO"):      # same instruction stream on all-zero data: separates power (DVFS) from cycles
            for x in (P1, pts4, ctr4, W1x, W2, b2, W3, b3):
                x.zero_()
        if precision == "f32":
            fn = lambda: ops.sa_mlp_max(P1, pts4, ctr4, gidx, W1x, W2, b2, W3, b3, C2)
        else:
            import ctypes as C
            from ev2hands_amd import _lib
            from ev2hands_amd.pack import sa_bf16_images
            i2, i3, u2, u3 = sa_bf16_images(W2[:C2].cpu().double().numpy(), W3[:, :C2].cpu().double().numpy(), _lib.PREC[precision])
            i2, i3 = torch.from_numpy(i2).cuda(), torch.from_numpy(i3).cuda()
            out = torch.empty(B, S, C3, device=d)
            dd = _lib.SaDesc()
            dd.P1, dd.ldp, dd.pts4, dd.ctr4, dd.gidx = P1.data_ptr(), C1, pts4.data_ptr(), ctr4.data_ptr(), gidx.data_ptr()
            dd.W1x, dd.b2, dd.b3, dd.W2s, dd.W3s = W1x.data_ptr(), b2.data_ptr(), b3.data_ptr(), i2.data_ptr(), i3.data_ptr()
            dd.out, dd.ldo = out.data_ptr(), C3
            dd.B, dd.Npts, dd.S, dd.K, dd.C1, dd.C2, dd.C3, dd.precision = B, Npts, S, K, C1, C2, C3, _lib.PREC[precision]
            dd.w2_unscale, dd.w3_unscale = u2, u3
            if os.environ.get("KBENCH_FEAT"):      # layer 1 from raw feature rows (enc.sa1, the regressors' sa1)
                feat = torch.randn(B, Npts, 8, device=d)
                W1f, b1 = torch.randn(C1, 4, device=d) * 0.5, torch.randn(C1, device=d)
                dd.feat, dd.ldf, dd.W1f, dd.ldw1f, dd.b1, dd.nfeat = feat.data_ptr(), 8, W1f.data_ptr(), 4, b1.data_ptr(), 4
                tag += "/feat"
            L = _lib.lib()
            fn = lambda: _lib.check(L.ev2h_sa_mlp_max(C.byref(dd), _lib.stream_handle()), "sa")
        ms = timeit(fn)
        flop = 2.0 * B * S * K * (C1 * C2 + C2 * C3)
        print(f"[{precision}] sa<{C1},{C2},{C3}> K={K} S={S} {tag:10s}: {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s ({flop / ms / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak)")


def bench_gemm(B=256, N=2048, precision="f32", rows=128):
    R = B * N
    for (M, Nn, K, taps, tag) in [(R, 512, 256, 3, "qconv0"), (R, 256, 256, 3, "qconv4"), (R, 256, 128, 1, "fp1.2"), (R, 128, 128, 1, "fp1.0"),
                                  (R, 256, 256, 1, "cls0"), (R, 4, 256, 1, "cls4"), (R, 160, 8, 1, "P1a"), (R, 256, 8, 1, "P1m"),
                                  (B * 512, 256, 576, 1, "fp2.0"), (B * 512, 256, 320, 1, "P1b"), (B * 128, 256, 520, 1, "sa3.0"),
                                  (B * 128, 1024, 512, 1, "sa3.2")]:
        if os.environ.get("KBENCH_GEMM_ONLY") and tag not in os.environ["KBENCH_GEMM_ONLY"].split(","):
            continue
        X = torch.randn(M, K, device="cuda")
        W = torch.randn(Nn, K * taps, device="cuda") * (K * taps) ** -0.5
        b = torch.randn(Nn, device="cuda")
        wi = ops.make_w_image(W, precision, rows) if (precision != "f32" and Nn >= 96) else None
        ms = timeit(lambda: ops.dense(X, W, b, True, taps=taps, rows_per_seq=N if taps == 3 else 0, K=K, precision=precision, w_image=wi,
                                      w_tile_rows=rows))
        flop = 2.0 * M * Nn * K * taps
        print(f"[{precision}/{rows}] gemm M={M} N={Nn} K={K}x{taps} {tag:8s}: {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s ({flop / ms / 1e9 / 157.3 * 100:5.1f}%)")


def bench_mano(B=256):
    from ev2hands_amd import synth
    from ev2hands_amd.mano import ManoHand
    hand = ManoHand(synth.synth_mano_assets("right", 0), "cuda")
    go, hp, be, tr = (torch.randn(B, n, device="cuda") * 0.3 for n in (3, 6, 10, 3))
    ms = timeit(lambda: hand(go, hp, be, tr), iters=20)
    print(f"mano layer B={B}: {ms * 1e3:8.1f} us per hand")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("sa", "all"):
        bench_sa()
    if what in ("sab", "all"):
        for prec in os.environ.get("KBENCH_PREC", "bf16x3,f16x2,bf16").split(","):
            bench_sa(precision=prec)
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("mano", "all"):
        bench_mano()
    if what in ("gemmb", "all"):
        for prec in os.environ.get("KBENCH_PREC", "bf16x3,f16x2,bf16").split(","):
            bench_gemm(precision=prec, rows=128)
        bench_gemm(precision="bf16x3", rows=256)
