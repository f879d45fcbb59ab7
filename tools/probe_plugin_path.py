"""Where the plugin path's host time goes (developer tool): cProfile over decode tokens through the patched forward.
python tools/probe_plugin_path.py [fused]   (fused: assume_causal + fuse_qkv)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev, dt = torch.device("cuda", 0), torch.bfloat16
variants = ((True, True),) if "fused" in sys.argv[1:] else ((False, False),)
print(bench.plugin_path_tokens_per_s(dev, dt, n_tokens=32, variants=variants))
pr = cProfile.Profile()
pr.enable()
bench.plugin_path_tokens_per_s(dev, dt, n_tokens=32, variants=variants)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
