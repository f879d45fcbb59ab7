"""The drop-in decode step under a captured HIP graph (spatten_amd/graph.py), alone — for rocprofv3:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pg -- python tools/probe_plugin_graph.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    variants = ((False, False, False), (True, True, True)) if len(sys.argv) < 2 else (tuple(int(c) for c in sys.argv[1]),)
    print(json.dumps(bench.plugin_path_tokens_per_s(dev, torch.bfloat16, variants=variants)))
