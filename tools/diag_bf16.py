import numpy as np, torch, sys
sys.path.insert(0, '.')
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O
for classes in (3, 11):
    topo = Topology.vtcnn2(classes); w = synthetic_weights(topo, seed=2016)
    m = VTCNN2(topo, dtype="bf16"); m.set_weights(w)
    for n in (1, 16, 64, 200):
        x = synthetic_frames(n, seed=2016)
        ref = O.forward("vtcnn2", x, w, dtype=np.float64, taps=True)
        f1 = m.predict(x, tap="flat"); f2 = m.predict(x, tap="flat")
        l1 = m.predict(x, tap="dense"); h1 = m.predict(x, tap="hidden")
        fe = np.abs(f1 - ref["flat"]); fs = np.abs(ref["flat"]).max()
        print(f"C={classes} n={n}: flat maxerr/scale {fe.max()/fs:.4f} rms/scale {np.sqrt((fe**2).mean())/fs:.5f} deterministic {np.array_equal(f1,f2)}"
              f" hidden err {np.abs(h1-ref['dense1']).max()/np.abs(ref['dense1']).max():.4f} logits err {np.abs(l1-ref['logits']).max()/np.abs(ref['logits']).max():.4f}")
        bad = np.argwhere(fe > 0.02*fs)
        if len(bad): print("   bad flat entries:", len(bad), bad[:5].tolist())
