"""Where the plugin path's host time goes (developer tool): cProfile over decode tokens through the patched forward."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev, dt = torch.device("cuda", 0), torch.bfloat16
for flag in (False, True):
    t0 = time.perf_counter()
    r = bench.plugin_path_tokens_per_s(dev, dt, n_tokens=32) if not flag else None
    if r:
        print(r)
pr = cProfile.Profile()
pr.enable()
bench.plugin_path_tokens_per_s(dev, dt, n_tokens=32)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
