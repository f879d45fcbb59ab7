"""Transposing-read form of the flash kernel (no Vt pre-pass) against the Vt form, per shape (SPATTEN_PREFILL_VTR=0/1)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q, n in ((64, 2112), (256, 2304), (512, 2560), (1024, 2048), (1024, 4096), (2048, 2048), (2048, 8192), (4096, 4096)):
    r = []
    for v in ("0", "1"):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "probe_prefill_shape.py"), str(q), str(n)],
                             env=dict(os.environ, SPATTEN_PREFILL_VTR=v), capture_output=True, text=True).stdout
        r.append(out.strip().splitlines()[-1].split(":")[1].split("us")[0].strip())
    print(f"q={q} N={n}: Vt form {r[0]} us | transposing reads {r[1]} us")
