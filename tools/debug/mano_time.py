import os, sys, torch
sys.path.insert(0, os.getcwd())
from ev2hands_amd import synth
from ev2hands_amd.mano import ManoHand
hand = ManoHand(synth.synth_mano_assets("right", 0), "cuda")
for B in (1, 8, 32):
    go, hp, be, tr = (torch.randn(B, n, device="cuda") * 0.3 for n in (3, 6, 10, 3))
    for _ in range(5): hand(go, hp, be, tr)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): hand(go, hp, be, tr)
    e.record(); torch.cuda.synchronize()
    print(f"parts env {os.environ.get('EV2H_MANO_PARTS')} B={B}: {s.elapsed_time(e) / 50 * 1e3:.1f} us per call (incl. launch overhead)")
