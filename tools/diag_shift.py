import sys, torch
sys.path.insert(0, '.')
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dtype)
x = synthetic_frames(1 << 17, seed=2016, device="cuda:0")
for tap in ("flat", "hidden", "dense"):
    a = m.predict(x[:70000].contiguous(), tap=tap)
    b = m.predict(x[:70000].contiguous(), tap=tap)
    print(dtype, tap, "repeat equal:", torch.equal(a, b))
    for sh in (3, 16, 37, 4096 + 5):
        c = m.predict(x[sh:70000].contiguous(), tap=tap)
        d = (a[sh:] != c)
        rows = d.flatten(1).any(dim=1).nonzero().flatten()
        print("   shift", sh, "equal:", not bool(d.any()), "bad rows:", rows[:8].tolist(), len(rows))
