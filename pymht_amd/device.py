"""Device-memory plumbing: PyTorch-ROCm owns the HBM buffers, libmht_amd gets raw pointers.

`NodeLayer` is the Python handle of one `mht_nodes` structure-of-arrays layer (include/mht_amd.h)."""
import ctypes as C
import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("pymht_amd needs an AMD GPU (MI355X / gfx950): torch.cuda.is_available() is False. "
                           "There is no CPU fallback.")


class Context:
    """One mht_ctx bound to torch's current stream on `device`."""

    def __init__(self, device=0, nx=4):
        require_gpu()
        self.nx = int(nx)
        self.lib = _lib.load(nx=self.nx)      # libmht_amd.so, or the six-state build libmht_amd6.so
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        _lib.check(self.lib.mht_create(C.byref(h), device, C.c_void_p(self.stream.cuda_stream)))
        self.handle = h

    def synchronize(self):
        _lib.check(self.lib.mht_synchronize(self.handle))

    def close(self):
        if self.handle:
            self.lib.mht_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_model(A, Q, Cm, R, eta2, lambda_ex, default_pd):
    nx = int(np.asarray(A).shape[0])
    m = _lib.model_type(nx)()
    for name, arr, n in (("A", A, nx * nx), ("Q", Q, nx * nx), ("C", Cm, 2 * nx), ("R", R, 4)):
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        assert a.size == n, name
        getattr(m, name)[:] = a.tolist()
    m.eta2 = float(eta2)
    m.lambda_ex = float(lambda_ex)
    m.default_pd = float(default_pd)
    m.default_miss_nllr = float(-np.log(1 - default_pd))      # host libm, as pyTarget.py:326 evaluates it
    return m


class NodeLayer:
    def __init__(self, cap, cap_cov, device):
        self.cap, self.cap_cov, self.device = int(cap), int(cap_cov), device
        self.x = torch.zeros((4, self.cap), dtype=torch.float64, device=device)
        self.cnllr = torch.zeros(self.cap, dtype=torch.float64, device=device)
        self.pd = torch.zeros(self.cap, dtype=torch.float64, device=device)
        self.parent = torch.full((self.cap,), -1, dtype=torch.int32, device=device)
        self.meas = torch.zeros(self.cap, dtype=torch.int32, device=device)
        self.cov = torch.zeros(self.cap, dtype=torch.int32, device=device)
        self.flags = torch.zeros(self.cap, dtype=torch.uint8, device=device)
        self.P = torch.zeros((16, self.cap_cov), dtype=torch.float32, device=device)
        self.struct = _lib.MhtNodes(self.x.data_ptr(), self.cnllr.data_ptr(), self.pd.data_ptr(),
                                    self.parent.data_ptr(), self.meas.data_ptr(), self.cov.data_ptr(),
                                    self.flags.data_ptr(), self.P.data_ptr(), self.cap, self.cap_cov)

    @classmethod
    def from_host(cls, x, P, cnllr, pd, flags, device, cap=None):
        """x (n,4) f64|f32, P (n,4,4) f32: one covariance column per node."""
        n = x.shape[0]
        layer = cls(cap or max(n, 1), cap or max(n, 1), device)
        layer.x[:, :n] = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64).T)).to(device)
        layer.P[:, :n] = torch.from_numpy(np.ascontiguousarray(np.asarray(P, dtype=np.float32).reshape(n, 16).T)).to(device)
        layer.cnllr[:n] = torch.from_numpy(np.asarray(cnllr, dtype=np.float64)).to(device)
        layer.pd[:n] = torch.from_numpy(np.asarray(pd, dtype=np.float64)).to(device)
        layer.flags[:n] = torch.from_numpy(np.asarray(flags, dtype=np.uint8)).to(device)
        layer.cov[:n] = torch.arange(n, dtype=torch.int32, device=device)
        return layer


def process_leaf_nodes(ctx, model, x, P, cnllr, pd, flags, z):
    """Stateless use of seam (i) (`Tracker._processLeafNodes`, tracker.py:383): host arrays in, host arrays out.
    Returns a dict shaped like the oracle's process_leaves() result (CSR instead of lists)."""
    lib, dev = ctx.lib, ctx.device
    n, M = x.shape[0], z.shape[0]
    lin = NodeLayer.from_host(x, P, cnllr, pd, flags, dev)
    zd = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 2)).to(dev)
    cap = max(n * 4 + 64, 1)
    while True:
        out = NodeLayer(cap, max(2 * n, 1), dev)
        child_ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        inc = torch.zeros(cap, dtype=torch.float64, device=dev)
        used = torch.zeros(max((M + 63) // 64, 1), dtype=torch.int64, device=dev)
        total = C.c_int32(0)
        rc = lib.mht_gate_scan(ctx.handle, C.byref(model), C.byref(lin.struct), None, n, zd.data_ptr(), M,
                               C.byref(out.struct), child_ptr.data_ptr(), inc.data_ptr(), used.data_ptr(), C.byref(total))
        if rc == _lib.MHT_E_CAPACITY:
            cap = total.value + 64
            continue
        _lib.check(rc)
        break
    nc = total.value
    cp = child_ptr.cpu().numpy().astype(np.int64)
    xs = out.x[:, :nc].cpu().numpy().T.copy()
    meas = out.meas[:nc].cpu().numpy()
    cn = out.cnllr[:nc].cpu().numpy()
    incs = inc[:nc].cpu().numpy()
    Pall = out.P[:, :2 * n].cpu().numpy().T.reshape(-1, 4, 4)
    hit = meas > 0
    return dict(child_ptr=cp, x=xs, meas=meas, cnllr=cn, inc=incs, flags=out.flags[:nc].cpu().numpy(),
                parent=out.parent[:nc].cpu().numpy(), cov=out.cov[:nc].cpu().numpy(),
                x_bar=xs[cp[:-1]], P_bar=Pall[0::2], P_hat=Pall[1::2],
                row_ptr=cp - np.arange(n + 1), col_idx=(meas[hit] - 1).astype(np.int64), x_hat=xs[hit],
                nllr=incs[hit], used=used.cpu().numpy().view(np.uint64))


def process_leaf_nodes_x(ctx, A, Q, Cm, R, eta2, lambda_ex, x, P, pd, flags, z, ct_period=None):
    """The dimension-generic form of seam (i) (`mht_gate_scan_x`: nx = 4 or 6 states): the reference's kalman module
    (predict, precalc, z_tilde, NIS, gate, numpyFilter, nllr -- kalman.py:14-101) for n leaves x M measurements in one call.
    Host arrays in (x (n,nx) f64|f32, P (n,nx,nx) f32), host arrays out, shaped like the oracle's process_leaves()."""
    lib, dev = ctx.lib, ctx.device
    n, nx, M = x.shape[0], x.shape[1], z.shape[0]
    keep = [np.ascontiguousarray(np.asarray(m, dtype=np.float32).ravel()) for m in (A, Q, Cm, R)]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    # (ct_period: the constant-turn model of models/ct.py -- A is rebuilt per leaf from its turn rate, the A passed in is ignored)
    model = _lib.MhtModelX(nx, fp(keep[0]), fp(keep[1]), fp(keep[2]), fp(keep[3]), float(eta2), float(lambda_ex),
                           0 if ct_period is None else 1, 0.0 if ct_period is None else float(ct_period))
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    xd = t(np.asarray(x, dtype=np.float64).T, np.float64) if n else torch.zeros((nx, 0), dtype=torch.float64, device=dev)
    Pd = t(np.asarray(P, dtype=np.float32).reshape(n, nx * nx).T, np.float32) if n else torch.zeros((nx * nx, 0), dtype=torch.float32, device=dev)
    pdd, fd = t(np.asarray(pd, dtype=np.float64), np.float64), t(np.asarray(flags, dtype=np.uint8), np.uint8)
    zd = t(np.asarray(z, dtype=np.float32).reshape(-1, 2), np.float32)
    f64 = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
    f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    x_bar, P_bar, P_hat, S, S_inv, K = f64(nx, max(n, 1)), f32(nx * nx, max(n, 1)), f32(nx * nx, max(n, 1)), f32(4, max(n, 1)), f32(4, max(n, 1)), f32(2 * nx, max(n, 1))
    if n:      # (the arrays are [rows][n] exactly: allocate them at that width)
        x_bar, P_bar, P_hat, S, S_inv, K = f64(nx, n), f32(nx * nx, n), f32(nx * nx, n), f32(4, n), f32(4, n), f32(2 * nx, n)
    row_ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    cap = max(4 * n + 64, 64)
    while True:
        col_idx = torch.zeros(cap, dtype=torch.int32, device=dev)
        x_hat, nllr = f64(nx, cap), f64(cap)
        total = C.c_int32(0)
        rc = lib.mht_gate_scan_x(ctx.handle, C.byref(model), n, xd.data_ptr(), fd.data_ptr(), Pd.data_ptr(), pdd.data_ptr(), zd.data_ptr(), M,
                                 x_bar.data_ptr(), P_bar.data_ptr(), P_hat.data_ptr(), S.data_ptr(), S_inv.data_ptr(), K.data_ptr(),
                                 row_ptr.data_ptr(), col_idx.data_ptr(), x_hat.data_ptr(), nllr.data_ptr(), cap, C.byref(total))
        if rc == _lib.MHT_E_CAPACITY:
            cap = total.value + 64
            continue
        _lib.check(rc)
        break
    g = total.value
    T = lambda a, *shape: a.cpu().numpy().T.reshape((-1,) + shape).copy()
    return dict(x_bar=T(x_bar, nx)[:n], P_bar=T(P_bar, nx, nx)[:n], P_hat=T(P_hat, nx, nx)[:n], S=T(S, 2, 2)[:n], S_inv=T(S_inv, 2, 2)[:n],
                K=T(K, nx, 2)[:n], row_ptr=row_ptr.cpu().numpy().astype(np.int64), col_idx=col_idx[:g].cpu().numpy().astype(np.int64),
                x_hat=x_hat[:, :g].cpu().numpy().T.copy(), nllr=nllr[:g].cpu().numpy())


def fuse_radar_and_ais(ctx, model, eta2, lambda_ex, x, P, pd, flags, own, ais_list, leaf_time, scan_time, eta2_ais, lambda_ais, z):
    """The stateless AIS seam (`mht_fuse_ais`): Tracker.__fuseRadarAndAis (tracker.py:417-552) for n leaves of one time against
    the AIS messages and the radar measurements of a scan.  Host arrays in (x (n,4) float64 holding float32 values where flags
    says so, P (n,4,4) float32, own (n,) identity the leaf's track is bound to or 0), host arrays out: child_ptr (n+1), x (g,4),
    P (g,4,4) float64, radar (g,) 0-based index or -1, nllr (g,), mmsi (g,) -- the children of leaf i are child_ptr[i] .. child_ptr[i+1]-1
    in the reference's order."""
    from .ais import group_messages
    lib, dev = ctx.lib, ctx.device
    n, M = x.shape[0], z.shape[0]
    groups, nG, msgs, order = group_messages(ais_list, float(leaf_time), float(scan_time), model)
    m = make_model(model.Phi(scan_time - leaf_time), model.Q(scan_time - leaf_time), model.C_RADAR, model.R_RADAR(), eta2, lambda_ex, 0.8)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    xd = t(np.asarray(x, dtype=np.float64).reshape(n, 4).T, np.float64) if n else torch.zeros((4, 0), dtype=torch.float64, device=dev)
    # (P float64: `mht_fuse_ais_f64` -- the leaves flagged MHT_F_COV_F64 are fused from their float64 covariance, as the reference does with
    # a node whose P_0 is float64)
    f64 = np.asarray(P).dtype == np.float64
    Pd = t(np.asarray(P).reshape(n, 16), np.float64 if f64 else np.float32)
    fuse = lib.mht_fuse_ais_f64 if f64 else lib.mht_fuse_ais
    pdd, fd = t(np.asarray(pd, dtype=np.float64), np.float64), t(np.asarray(flags, dtype=np.uint8), np.uint8)
    od = t(np.asarray(own, dtype=np.int32), np.int32)
    zd = t(np.asarray(z, dtype=np.float32).reshape(-1, 2), np.float32)
    ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    cap = max(8 * n + 64, 64)
    while True:
        ox = torch.zeros((4, cap), dtype=torch.float64, device=dev)
        oP = torch.zeros((cap, 16), dtype=torch.float64, device=dev)
        orad, omsg = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
        onl = torch.zeros(cap, dtype=torch.float64, device=dev)
        total = C.c_int32(0)
        rc = fuse(ctx.handle, C.byref(m), n, xd.data_ptr(), fd.data_ptr(), Pd.data_ptr(), pdd.data_ptr(), od.data_ptr(),
                              C.byref(groups), nG, C.byref(msgs), len(order), float(eta2_ais), float(lambda_ais), zd.data_ptr(), M,
                              ptr.data_ptr(), ox.data_ptr(), oP.data_ptr(), orad.data_ptr(), onl.data_ptr(), omsg.data_ptr(), cap, C.byref(total))
        if rc == _lib.MHT_E_CAPACITY:
            cap = total.value + 64
            continue
        _lib.check(rc)
        break
    g = total.value
    mi = omsg[:g].cpu().numpy()
    return dict(child_ptr=ptr.cpu().numpy().astype(np.int64), x=ox[:, :g].cpu().numpy().T.copy(), P=oP[:g].cpu().numpy().reshape(-1, 4, 4),
                radar=orad[:g].cpu().numpy().astype(np.int64), nllr=onl[:g].cpu().numpy(),
                mmsi=np.array([ais_list[order[i]].mmsi for i in mi], dtype=np.int64))
