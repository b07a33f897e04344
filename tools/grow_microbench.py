"""Development aid: time the stateless grow kernel (mht_gate_scan) on the headline-shape golden batch, replicated to
scale L (shows where the kernel leaves the launch-latency floor and what it reaches against the HBM roofline)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymht_amd.device import Context, NodeLayer, make_model
from pymht_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); import mht_oracle as orc
g = np.load(os.path.join(ROOT, 'tests/golden/g5_headline.npz'))
reps = [int(v) for v in sys.argv[1:]] or [1, 4, 16, 64, 256]
ctx = Context(0); dev = ctx.device
model = make_model(orc.model_Phi(2.5), orc.model_Q(2.5), orc.model_C(), orc.model_R(), 5.99, float(g['lambda_ex']), 0.9)
P0 = g['P_table'][g['Pidx'].astype(np.int64)]
z = g['z']; M = z.shape[0]
zd = torch.from_numpy(z).to(dev)
for rep in reps:
    x = np.tile(g['x'], (rep, 1)); P = np.tile(P0, (rep, 1, 1))
    n = x.shape[0]
    lin = NodeLayer.from_host(x, P, np.zeros(n), np.full(n, 0.9), np.zeros(n, np.uint8), dev)
    out = NodeLayer(n * 3, 2 * n, dev)
    cp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    def run():
        _lib.check(ctx.lib.mht_gate_scan(ctx.handle, C.byref(model), C.byref(lin.struct), None, n, zd.data_ptr(), M, C.byref(out.struct), cp.data_ptr(), None, None, None))
    for _ in range(5): run()
    ctx.synchronize()
    K = 50 if rep <= 16 else 10
    t0 = time.perf_counter()
    for _ in range(K): run()
    ctx.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / K
    G = int(cp[-1].item()) - n
    B = 280 * n + 48 * G + 8 * M
    print("ablate=%s L=%d M=%d G=%d  %.2f us per call (incl. 1 tiny memset)  %.1f GB/s algorithmic  %.2f Gpairs/s" % (
        os.environ.get('MHT_GROW_ABLATE', '0'), n, M, G, us, B / us * 1e-3, n * M / us * 1e-3))
    del lin, out, cp
