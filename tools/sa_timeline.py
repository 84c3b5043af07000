"""Phase timeline of one wave of the pipelined bf16 SA kernel (s_memtime ticks ~ shader clocks).
Needs the instrumented build:  EV2H_BUILD_DEFS=-DEV2H_SAB_TIMELINE python -m ev2hands_amd.build --force
usage: [EV2H_PRECISION=f16x2|bf16x3|bf16] python tools/sa_timeline.py C1 C2 C3 K S   (default 128 196 256 128 128)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ev2hands_amd import _lib, ops
prec = os.environ.get("EV2H_PRECISION", "f16x2")
nprod = {"bf16": 1, "f16x2": 3, "bf16x3": 6}[prec]
a = [int(x) for x in sys.argv[1:6]]
C1, C2, C3, K, S = a if len(a) == 5 else (128, 196, 256, 128, 128)
B, Npts = 256, 2048
d = "cuda"
up = lambda x, m: (x + m - 1) // m * m
P1 = torch.randn(B, Npts, C1, device=d); pts4 = torch.randn(B, Npts, 4, device=d); ctr4 = torch.randn(B, S, 4, device=d)
gidx = torch.randint(0, Npts, (B, S, K), device=d, dtype=torch.int32)
W1x = torch.randn(C1, 4, device=d); W2 = torch.randn(up(C2, 32), C1, device=d) * C1 ** -0.5; b2 = torch.randn(up(C2, 32), device=d)
W3 = torch.randn(C3, up(C2, 8), device=d) * C2 ** -0.5; b3 = torch.randn(C3, device=d)
for _ in range(3):
    ops.sa_mlp_max(P1, pts4, ctr4, gidx, W1x, W2, b2, W3, b3, C2, prec)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_longlong * 256)()
L.ev2h_sab_timeline_read(buf)
t = list(buf)
NC1, T3, T2 = C1 // 32, C3 // 32, (C2 + 31) // 32
print(f"[{prec}] sa<{C1},{C2},{C3}> K={K}: strip {t[37] - t[0]} ticks;  neighbour gather (index -> rows landed) {t[1] - t[0]}")
for c in range(NC1):
    print(f"  l2 chunk {c}: layer-1 finish+split {t[3+4*c]-t[2+4*c]:6d}  MFMA issue {t[4+4*c]-t[3+4*c]:6d}  dma wait + barrier {t[5+4*c]-t[4+4*c]:6d}   (own MFMAs {T2*2*nprod*32})")
print(f"  h2 ReLU + split: {t[39] - t[38]}")
ng = 2 * (T2 - 1) + (2 if C2 % 32 == 0 or C2 % 32 > 16 else 1)
for u in range(T3):
    print(f"  l3 tile {u}: dma issue + MFMA issue {t[41+4*u]-t[40+4*u]:6d}  max {t[42+4*u]-t[41+4*u]:5d}  dma wait + barrier {t[43+4*u]-t[42+4*u]:6d}   (own MFMAs {ng*nprod*32})")
