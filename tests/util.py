"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from oracle import spatten_oracle as orc

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}

# Stated tolerances, HIP vs reference/oracle (see DESIGN.md "Parity bar"):
#  * indices, masks, gathered KV rows, appended cache rows: bit exact
#  * fp32 attention output / stash: accumulation ORDER differs -> atol 2e-5, rtol 1e-5
#  * 16-bit attention output: the reference rounds the NORMALISED P to the model dtype before P.V, the kernel rounds the
#    un-normalised exp(s - running max) (decode_attn.hip header: the denominator is not known before the first V row)
#    -> a few output ulps: bf16 atol 1e-2 rtol 2e-2; f16 atol 2e-3 rtol 4e-3
#  * 16-bit stash: every reference rounding step is reproduced; a value may still land one ulp away
#    where fp32 accumulation order crosses the matmul->dtype rounding boundary; the following
#    /sqrt(d) -> dtype step stretches that to <= 2 ulp of the (smaller) quotient -> <2% of entries, <= 2 ulp.
#    f16 only: torch's CPU half GEMM does not accumulate purely in fp32 (observed 4e-4 absolute deviation on a
#    cancelling dot product of O(1) terms, reproduced by neither numpy, C nor the GPU) -> atol 6e-5 after /sqrt(d)
OUT_TOL = {"f32": dict(atol=2e-5, rtol=1e-5), "bf16": dict(atol=1e-2, rtol=2e-2), "f16": dict(atol=2e-3, rtol=4e-3)}
STASH_TOL = {"f32": dict(atol=2e-5, rtol=1e-5), "bf16": dict(atol=2e-6, rtol=2 ** -6), "f16": dict(atol=6e-5, rtol=2 ** -9)}


def dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TORCH_DT[dt]).cuda()


def host(t):
    return t.detach().float().cpu().numpy()


def golden(name):
    here = os.path.dirname(os.path.abspath(__file__))
    return np.load(os.path.join(here, "golden", name), allow_pickle=False)


def slab(x, cap):
    """[B,H,L,d] numpy -> device slab [B,H,cap,d] holding x in rows [0,L); rest poisoned with NaN."""
    B, H, L, d = x.shape
    t = torch.full((B, H, cap, d), float("nan"), dtype=x.dtype if isinstance(x, torch.Tensor) else torch.float32)
    t[:, :, :L] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if not isinstance(x, torch.Tensor) else x
    return t


def check_stash(got, want, dt, name=""):
    np.testing.assert_allclose(got, want, err_msg=f"stash {name}", **STASH_TOL[dt])
    if dt != "f32":
        frac = float(np.mean(got != want))
        assert frac < 0.02, (name, frac)


def attn_inputs(B, H, Hkv, d, P, ql, dt, seed):
    q = orc.synth_normal(seed, 0, (B, H, ql, d), dt)
    k = orc.synth_normal(seed, 1, (B, Hkv, ql, d), dt)
    v = orc.synth_normal(seed, 2, (B, Hkv, ql, d), dt)
    past = None
    if P > 0:
        past = (orc.synth_normal(seed, 3, (B, Hkv, P, d), dt), orc.synth_normal(seed, 4, (B, Hkv, P, d), dt))
    return q, k, v, past
