"""The peer-store all-gather (include/spatten.h: spatten_peer_*; SURVEY 8e "single-shot direct writes"): receive windows mapped
through hipIpc, one launch per all-gather, epoch flags in the windows.  What one GPU can check: the one-rank loopback (eager
and replayed from a HIP graph: the epoch advances on the device), and TWO PROCESSES on the same device exchanging real IPC
handles and each other's slices (the cross-process mapping, the flag protocol, the two window halves)."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_loopback_eager_and_graph_replayed():
    from spatten_amd.parallel import HeadParallel
    hp = HeadParallel(8, rank=0, world=1, gather_fn=lambda t, r: t)
    hp.init_peer_store(4096)
    try:
        for n in (8, 256, 2048):
            send = torch.arange(n, dtype=torch.int16, device="cuda") * 3
            recv = torch.zeros(n, dtype=torch.int16, device="cuda")
            hp.allgather_peer(send, recv)
            torch.cuda.synchronize()
            assert torch.equal(send, recv)
        # captured: the launch is its own epoch source, so one graph replays any number of times
        send = torch.zeros(512, dtype=torch.float32, device="cuda")
        recv = torch.zeros_like(send)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            hp.allgather_peer(send, recv)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                hp.allgather_peer(send, recv)
            for t in range(7):
                send.fill_(float(t + 1))
                g.replay()
                side.synchronize()
                assert float(recv.min()) == float(recv.max()) == float(t + 1)
        hp.peer_status()
        with pytest.raises(Exception):
            hp.allgather_peer(torch.zeros(4096, dtype=torch.float32, device="cuda"), torch.zeros(4096, dtype=torch.float32, device="cuda"))
    finally:
        hp.close_peer_store()


WORKER = r'''
import os, sys, time, torch
sys.path.insert(0, sys.argv[3])
from spatten_amd.parallel import HeadParallel
rank, d = int(sys.argv[1]), sys.argv[2]
torch.cuda.set_device(0)
def exchange(mine):
    open(os.path.join(d, f"h{rank}.tmp"), "wb").write(mine)
    os.rename(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}"))
    out = []
    for r in range(2):
        p = os.path.join(d, f"h{r}")
        t0 = time.time()
        while not os.path.exists(p):
            if time.time() - t0 > 60: raise SystemExit("peer handle never appeared")
            time.sleep(0.01)
        out.append(open(p, "rb").read())
    return out
hp = HeadParallel(8, rank=rank, world=2, gather_fn=lambda t, r: t)
hp.init_peer_store(8192, exchange_handles=exchange)
n = 1024
for step in range(40):                       # many epochs: both window halves, reuse after the peer moved on
    send = torch.full((n,), float(100 * rank + step), dtype=torch.float32, device="cuda")
    recv = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    hp.allgather_peer(send, recv)
    torch.cuda.synchronize()
    for r in range(2):
        blk = recv[r * n:(r + 1) * n]
        assert float(blk.min()) == float(blk.max()) == float(100 * r + step), (rank, step, r, float(blk.min()), float(blk.max()))
    if step % 7 == rank:
        time.sleep(0.02)                     # uneven pace between the two ranks
hp.peer_status()
# captured form
send = torch.zeros(n, dtype=torch.float32, device="cuda"); recv = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        hp.allgather_peer(send, recv)
    for step in range(10):
        send.fill_(float(1000 + 10 * rank + step))
        g.replay(); side.synchronize()
        for r in range(2):
            blk = recv[r * n:(r + 1) * n]
            assert float(blk.min()) == float(blk.max()) == float(1000 + 10 * r + step), (rank, step, r)
hp.peer_status()
hp.close_peer_store()
print("PEER_OK", rank)
'''


def test_two_processes_on_one_device_exchange_through_ipc_windows():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), d, root], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append((p.returncode, o, e))
    if any("not supported" in e.lower() or "ERR_UNSUPPORTED" in e or "status -2" in e for _, _, e in outs):
        pytest.skip("hipIpc memory handles are not available on this box: " + outs[0][2][-300:])
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f"PEER_OK {r}" in o, (rc, o[-500:], e[-2000:])
