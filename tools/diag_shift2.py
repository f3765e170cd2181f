import sys, torch, numpy as np
sys.path.insert(0, '.')
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O
topo = Topology.vtcnn2(11); w = synthetic_weights(topo, seed=2016)
m = VTCNN2(topo, device=0, dtype="bf16"); m.set_weights(w)
x = synthetic_frames(1 << 14, seed=2016, device="cuda:0")
a = m.predict(x[:9000].contiguous(), tap="flat")
c = m.predict(x[3:9000].contiguous(), tap="flat")
d = (a[3:] != c)
rows = d.any(dim=1).nonzero().flatten()
print("bad rows", len(rows), rows[:10].tolist())
for r in rows[:4].tolist():
    cols = d[r].nonzero().flatten()
    o = (cols // 132).tolist(); wpos = (cols % 132).tolist()
    print("row", r + 3, "ncols", len(cols), "w positions", sorted(set(wpos))[:20], "o", sorted(set(o))[:20])
    print("   values a", a[r + 3, cols[:4]].tolist(), "c", c[r, cols[:4]].tolist())
# which is right? oracle for a few bad rows
sel = [int(r) + 3 for r in rows[:6].tolist()]
ref = O.forward("vtcnn2", x[sel].cpu().numpy(), w, dtype=np.float64, taps=True)["flat"]
for k, r in enumerate(sel):
    ea = np.abs(a[r].cpu().numpy() - ref[k]).max(); ec = np.abs(c[r - 3].cpu().numpy() - ref[k]).max()
    print("row", r, "err(a) %.3e err(c) %.3e scale %.3e" % (ea, ec, np.abs(ref[k]).max()))
