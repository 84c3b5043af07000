"""Time the farthest-point sampling operator alone:  python tools/debug/fps_time.py [N] [B] [S]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import ops
N, B, S = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 32768), (2, 16), (3, 512)))
g = torch.Generator().manual_seed(0)
xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
init = torch.randint(0, N, (B,), generator=g).cuda()
for _ in range(2):
    idx = ops.farthest_point_sample(xyz, S, init)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    idx = ops.farthest_point_sample(xyz, S, init)
e.record(); torch.cuda.synchronize()
print(f"fps N={N} B={B} S={S}: {s.elapsed_time(e) / 5:.3f} ms   checksum {int(idx.sum())}")
