"""Development aid: time the stateless grow kernel (mht_gate_scan) on the headline-shape golden batch."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymht_amd.device import Context, NodeLayer, make_model
from pymht_amd import _lib
sys.path.insert(0, 'oracle'); import mht_oracle as orc
g = np.load('tests/golden/g5_headline.npz')
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # replicate the batch to scale L
P = g['P_table'][g['Pidx'].astype(np.int64)]
x = np.tile(g['x'], (rep, 1)); P = np.tile(P, (rep, 1, 1)); z = g['z']
n, M = x.shape[0], z.shape[0]
ctx = Context(0); dev = ctx.device
model = make_model(orc.model_Phi(2.5), orc.model_Q(2.5), orc.model_C(), orc.model_R(), 5.99, float(g['lambda_ex']), 0.9)
lin = NodeLayer.from_host(x, P, np.zeros(n), np.full(n, 0.9), np.zeros(n, np.uint8), dev)
zd = torch.from_numpy(z).to(dev)
out = NodeLayer(n * 3, 2 * n, dev)
cp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
def run():
    _lib.check(ctx.lib.mht_gate_scan(ctx.handle, C.byref(model), C.byref(lin.struct), None, n, zd.data_ptr(), M, C.byref(out.struct), cp.data_ptr(), None, None, None))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
e0.record()
for _ in range(K): run()
e1.record(); torch.cuda.synchronize()
G = int(cp[-1].item()) - n
print("ablate=%s L=%d M=%d G=%d  %.2f us per call (incl. 1 tiny memset)" % (os.environ.get('MHT_GROW_ABLATE', '0'), n, M, G, 1e3 * e0.elapsed_time(e1) / K))
