import sys, json
sys.path.insert(0, "/root/repo")
import bench
for f in (3, 10):
    r = bench.run_q612(f, 0)
    print(f, "%.4g frames/s" % r["value"], "ms/step %.3f" % r["ms_per_step"])
