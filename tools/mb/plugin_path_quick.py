import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
r = bench.plugin_path_tokens_per_s(dev, torch.bfloat16, variants=((True, True, True), (True, True, "fused")))
print(json.dumps({k: v for k, v in r.items() if 'tokens_per_s' in k and 'turn' not in k}, indent=0))
