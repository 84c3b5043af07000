"""Which summation order does this host's torch-CPU matmul use for the K=3 distance dot products?  (lattice cloud: every order
mismatch is visible; compare with tests/diag_host_matmul.py on random data where it is not).  Imports oracle/: test tooling."""
import os, sys, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2hands_amd import synth
from oracle import tehnet_oracle
B, N, S = 2, 2048, 512
xyz = synth.synth_cloud("L", B, 4, N, 41)[:, :3].permute(0, 2, 1).contiguous()
fi = tehnet_oracle.farthest_point_sample(xyz, S, torch.tensor([0, 1]))
ctr = torch.stack([xyz[b, fi[b]] for b in range(B)])
c = ctr.numpy().astype(np.float64); q = xyz.numpy().astype(np.float64)
f32 = lambda x: x.astype(np.float32).astype(np.float64)
fma = lambda a, b, cc: f32(a * b + cc)
cx, cy, cz = [c[:, :, None, i] for i in range(3)]; qx, qy, qz = [q[:, None, :, i] for i in range(3)]
cands = {"fma z(y(x))": fma(cz, qz, fma(cy, qy, f32(cx * qx))), "fma x(y(z))": fma(cx, qx, fma(cy, qy, f32(cz * qz))),
         "mul-add (x+y)+z": f32(f32(f32(cx * qx) + f32(cy * qy)) + f32(cz * qz)), "mul-add x+(y+z)": f32(f32(cx * qx) + f32(f32(cy * qy) + f32(cz * qz))),
         "exact dot": f32(cx * qx + cy * qy + cz * qz)}
for nt in (1, 8, 64):
    torch.set_num_threads(nt)
    mm = torch.matmul(ctr, xyz.transpose(1, 2)).numpy()
    print("threads", nt, {k: int((v.astype(np.float32) != mm).sum()) for k, v in cands.items()})
mm2 = torch.einsum("bsc,bnc->bsn", ctr, xyz).numpy()
print("einsum", {k: int((v.astype(np.float32) != mm2).sum()) for k, v in cands.items()})
mmT = torch.matmul(xyz, ctr.transpose(1, 2)).numpy()          # the 3-NN direction [N,3] x [3,S]
print("transposed problem", {k: int((v.astype(np.float32).transpose(0, 2, 1) != mmT).sum()) for k, v in cands.items()})
print(subprocess.run("lscpu | grep -E 'Model name'", shell=True, capture_output=True, text=True).stdout.strip())
print("MKL_ENABLE_INSTRUCTIONS", os.environ.get("MKL_ENABLE_INSTRUCTIONS"), "MKL_DEBUG_CPU_TYPE", os.environ.get("MKL_DEBUG_CPU_TYPE"))
