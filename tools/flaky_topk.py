import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import spatten_oracle as orc
from spatten_amd import ops
from tests.util import dev
rng = np.random.default_rng(3)
H, L = 6, 1000
bad = 0
for dt in ("f16", "bf16", "f32"):
    s = orc.round_dt(rng.standard_normal((H, L)).astype(np.float32), dt)
    s[0] = 0.0; s[1] = np.round(s[1] * 2) / 2; s[2, ::7] = np.inf; s[2, 5::11] = -np.inf; s[3, 3::13] = np.nan
    s[4, :] = orc.round_dt(np.where(rng.random(L) < 0.5, 0.0, -0.0).astype(np.float32), dt)
    sd = dev(s, dt)
    for it in range(300):
        for lo, hi, k in ((0, L, 1), (0, L, L), (4, 900, 300), (10, 11, 1), (3, 997, 994), (100, 612, 256)):
            want = orc.topk_window(s, lo, hi, k)
            got = ops.topk_select(sd, lo, hi, k).cpu().numpy()
            if not np.array_equal(got, want):
                bad += 1
                rows = np.where((got != want).any(axis=1))[0]
                print(dt, it, (lo, hi, k), "rows", rows, "first diff", [(int(r), got[r][got[r] != want[r]][:5].tolist(), want[r][got[r] != want[r]][:5].tolist()) for r in rows[:2]])
                if bad > 8: sys.exit(1)
print("bad", bad)
