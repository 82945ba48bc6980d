"""replays the captured two-workgroup window solve many times over alternating systems; reports every wrong replay (which one, how
many unknowns off, where).  GRAPH=0: the same sequence with eager launches."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dba-fusion_amd"))
import test_gpu_solve as T
from dbaf_amd import _lib
P, w = int(os.environ.get("P", 63)), int(os.environ.get("W", 8))
use_graph = os.environ.get("GRAPH", "1") != "0"
reps = int(os.environ.get("REPS", 600))
rng = np.random.default_rng(99)
S = T._SkylineSolver(P)
systems = [T._pose_system(rng, P, w) for _ in range(3)]
refs = [T._ref(H, b) for H, b, _ in systems]
dx, failed = S.solve(*systems[0])
print("eager first:", failed, S.fronts)
n, lay, ws = 6 * P, S.lay, S.ws
fp = torch.from_numpy(systems[0][2]).cuda()
Hd = [torch.from_numpy((np.tril(H) + np.triu(np.full_like(H, 1e300), 1)).reshape(-1)).cuda() for H, b, _ in systems]
bd = [torch.from_numpy(b).cuda() for H, b, _ in systems]
def launch():
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(S.lib.dba_ba_solve_skyline(*S.dims, 1e-4, 0.1, ctypes.c_void_p(fp.data_ptr()), ctypes.c_void_p(ws.data_ptr()), S.nbytes, stream), "solve")
if use_graph:
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            launch()
torch.cuda.synchronize()
nbad = 0
BADS = [int(x) for x in os.environ.get('BADS', '200,5,370,180,190').split(',') if x]
for rep in range(reps):
    k = rep % 3
    ws[lay.H:lay.H + 8 * n * n].view(torch.float64).copy_(Hd[k])
    bad = BADS and rep % 7 == 5
    if bad:
        ws[lay.H:lay.H + 8 * n * n].view(torch.float64)[(BADS[(rep // 7) % len(BADS)]) * (n + 1)] = -1.0
    ws[lay.b:lay.b + 8 * n].view(torch.float64).copy_(bd[k])
    ws[lay.dx:lay.dx + 4 * n].view(torch.float32).fill_(7.0)
    if os.environ.get("SYNC", "1") != "0":
        torch.cuda.synchronize()
    graph.replay() if use_graph else launch()
    torch.cuda.synchronize()
    dx = ws[lay.dx:lay.dx + 4 * n].view(torch.float32).cpu().numpy()
    meta = ws[lay.meta:lay.meta + 128].view(torch.int32).cpu().numpy()
    err = np.abs(dx - refs[k]); tol = 3e-7 * max(1.0, np.abs(refs[k]).max())
    if bad:
        if int(meta[1]) != 1 or dx.any():
            nbad += 1; print("rep", rep, "bad system not refused", meta[:8])
        continue
    if int(meta[1]) != 0 or (err > tol).any():
        nbad += 1
        idx = np.nonzero(err > tol)[0]
        print("rep", rep, "failed" , int(meta[1]), "off:", len(idx), "range", idx.min() if len(idx) else -1, idx.max() if len(idx) else -1,
              "max err", err.max(), "meta", meta[:8], meta[8:16], meta[24:28])
print("graph" if use_graph else "eager", "P", P, "w", w, "wrong:", nbad, "of", reps)
