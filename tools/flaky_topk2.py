import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatten_amd import _lib
lib = _lib.load()
L = 1000
rng = np.random.default_rng(3)
s = rng.standard_normal((1, L)).astype(np.float32)
s[0, 3::13] = np.nan
sd = torch.from_numpy(s).cuda()
st = torch.cuda.current_stream().cuda_stream
bad = 0
for it in range(3000):
    idx = torch.full((1, 8), -7, dtype=torch.int32, device="cuda")
    rc = lib.spatten_topk_select(0, sd.data_ptr(), L, 1, 0, L, 1, idx.data_ptr(), 8, st)
    got = idx.cpu().numpy()[0]
    if got[0] != 3 or (got[1:] != -7).any():
        bad += 1
        print(it, got)
        if bad > 5: break
print("bad", bad)
