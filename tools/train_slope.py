import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from modulationdetectioncnn_amd import Topology, synthetic_frames, synthetic_weights
from modulationdetectioncnn_amd.training import Trainer
for name, topo in (("T1", Topology.deployed(3)), ("T2", Topology.deployed(10)), ("T4", Topology.cnnpy(10,10,5))):
    n = 65536
    x = synthetic_frames(n, seed=1, device="cuda:0")
    lab = torch.randint(0, topo.classes, (n,), device="cuda:0")
    tr = Trainer(topo, synthetic_weights(topo, seed=1), device=0)
    xd, yd = tr._frames(x), tr._targets(lab, n)
    for mode in ("grad", "eval"):
        for count in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
            f = (lambda: tr.train_batch(xd, yd, None, 0, count, apply=False)) if mode == "grad" else \
                (lambda: tr._check(tr._lib().mdc_trainer_evaluate(tr._h, xd.data_ptr(), yd.data_ptr(), n, None, 0, count, tr._stream())))
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            print(name, mode, count, "%.2f us" % (e0.elapsed_time(e1) / 50 * 1e3), flush=True)
    tr.close()
