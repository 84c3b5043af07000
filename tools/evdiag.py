import sys; sys.path.insert(0,"/root/repo")
import numpy as np, torch
from oracle import event_window_oracle as EW
from ev2hands_amd.events import EventWindowBuilder
g = np.load("/root/repo/tests/golden/events_0.npz")
raw = g["raw0"]
bld = EventWindowBuilder("cuda:0")
table, counts = bld.accumulate([raw])
ref = g["table0"].astype(np.float32); M = ref.shape[0]
got = table[0,:M,:5].cpu().numpy()
print("M", M, int(counts[0]))
for c in range(5):
    bad = np.nonzero(got[:,c] != ref[:,c])[0]
    print("col", c, "mismatches", bad.size, (got[bad[:5],c], ref[bad[:5],c]) if bad.size else "")
# which pixels: count>1?
xi, yi = ref[:,0].astype(int), ref[:,1].astype(int)
cnt = ref[:,3]+ref[:,4]
bad = np.nonzero(got[:,2] != ref[:,2])[0]
print("counts at mismatches", np.unique(cnt[bad], return_counts=True))
print("counts overall", np.unique(cnt, return_counts=True))
