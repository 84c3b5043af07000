"""Diagnostic (kept for reference): how the host's torch-CPU evaluates the K=3 matmul in square_distance (fma chain), the fact the
3-NN / ball-query parity rests on.  Imports oracle/, so it is test tooling, not product code."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, subprocess
from ev2hands_amd import synth, ops
from oracle import tehnet_oracle as O
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)'; lscpu | grep -o -E 'avx512f|avx2|amx_tile' | sort -u | tr '\\n' ' '", shell=True, capture_output=True, text=True).stdout)
print(torch.__config__.show().split('\n')[6])
def f32(x): return x.astype(np.float32).astype(np.float64)
def emu(src, dst):
    A = src.numpy().astype(np.float64); Bm = dst.numpy().astype(np.float64)
    dot = f32(f32(f32(A[:,:,None,0]*Bm[:,None,:,0]) + A[:,:,None,1]*Bm[:,None,:,1]) + A[:,:,None,2]*Bm[:,None,:,2])
    sq = lambda P: f32(f32(f32(P[...,0]*P[...,0]) + f32(P[...,1]*P[...,1])) + f32(P[...,2]*P[...,2]))
    return f32(f32(-2*dot + sq(A)[:,:,None]) + sq(Bm)[:,None,:])
def emu_nofma(src, dst):
    A = src.numpy().astype(np.float64); Bm = dst.numpy().astype(np.float64)
    p = [f32(A[:,:,None,k]*Bm[:,None,:,k]) for k in range(3)]
    dot = f32(f32(p[0]+p[1])+p[2])
    sq = lambda P: f32(f32(f32(P[...,0]*P[...,0]) + f32(P[...,1]*P[...,1])) + f32(P[...,2]*P[...,2]))
    return f32(f32(-2*dot + sq(A)[:,:,None]) + sq(Bm)[:,None,:])
for kind in ("U","E"):
    B, N1, N2 = 2, 2048, 512
    xyz1 = synth.synth_cloud(kind,B,4,N1,31)[:, :3].permute(0,2,1).contiguous()
    fps = O.farthest_point_sample(xyz1, N2, torch.zeros(B,dtype=torch.long))
    xyz2 = O.gather_points(xyz1, fps)
    d = O.pairwise_sqdist(xyz1, xyz2).numpy().astype(np.float64)
    e = emu(xyz1, xyz2); e2 = emu_nofma(xyz1, xyz2)
    print(kind, 'torch-vs-fma-emu mismatches', (d!=e).sum(), 'torch-vs-nofma-emu', (d!=e2).sum(), 'of', d.size)
    # matmul alone
    m = torch.matmul(xyz1, xyz2.transpose(1,2)).numpy().astype(np.float64)
    A = xyz1.numpy().astype(np.float64); Bm = xyz2.numpy().astype(np.float64)
    dot = f32(f32(f32(A[:,:,None,0]*Bm[:,None,:,0]) + A[:,:,None,1]*Bm[:,None,:,1]) + A[:,:,None,2]*Bm[:,None,:,2])
    print('   matmul mismatches', (m!=dot).sum())
    for thr in (1, 8):
        torch.set_num_threads(thr)
        m2 = torch.matmul(xyz1, xyz2.transpose(1,2)).numpy().astype(np.float64)
        print('   threads', thr, 'matmul mismatches vs fma-emu', (m2!=dot).sum())
    # GPU weights vs emulated weights
    f2 = torch.zeros(B, N2, 4)
    out, gi, gw = ops.three_nn_interpolate(xyz1.cuda(), xyz2.cuda(), f2.cuda())
    es = np.sort(e, axis=-1)[:, :, :3].astype(np.float32)
    r = (np.float32(1.0) / (es + np.float32(1e-8))).astype(np.float32)
    nrm = ((r[...,0] + r[...,1]).astype(np.float32) + r[...,2]).astype(np.float32)
    w_emu = (r / nrm[...,None]).astype(np.float32)
    print('   gpu-vs-emu weights max abs', np.abs(gw.cpu().numpy() - w_emu).max())
    idx, w = O.three_nn_weights(xyz1, xyz2)
    print('   torch-vs-emu weights max abs', np.abs(w.numpy() - w_emu).max())
