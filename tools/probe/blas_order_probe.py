"""Development probe (not product, not a test): which evaluation order does this host's NumPy/OpenBLAS use for the small float32
products on the initiator's path (m_of_n.py:186-188 F.dot(state); :303 K.dot(delta); :283 C.dot(pred); :205-207 similarity)?
Prints the runtime OpenBLAS core and, per operation, how many of N random cases each candidate order reproduces bit for bit."""
import ctypes, os, subprocess, tempfile
import numpy as np

d = tempfile.mkdtemp()
open(os.path.join(d, "fm.c"), "w").write("#include <math.h>\nfloat c_fmaf(float a,float b,float c){return fmaf(a,b,c);}\ndouble c_fma(double a,double b,double c){return fma(a,b,c);}\n")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-mfma", "-ffp-contract=off", os.path.join(d, "fm.c"), "-o", os.path.join(d, "fm.so"), "-lm"])
lib = ctypes.CDLL(os.path.join(d, "fm.so"))
lib.c_fmaf.restype = ctypes.c_float; lib.c_fmaf.argtypes = [ctypes.c_float] * 3
lib.c_fma.restype = ctypes.c_double; lib.c_fma.argtypes = [ctypes.c_double] * 3
f32 = np.float32
try:
    from threadpoolctl import threadpool_info
    print([(i.get("internal_api"), i.get("version"), i.get("architecture")) for i in threadpool_info()])
except Exception as e:
    print("threadpoolctl:", e)
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])


def fma(a, b, c): return f32(lib.c_fmaf(float(a), float(b), float(c)))


rng = np.random.default_rng(0)
def rnd(shape): return (rng.normal(size=shape) * rng.choice([1, 100, 0.01])).astype(f32)


def cands(a, x):
    K = len(a)
    p = [f32(a[i]) * f32(x[i]) for i in range(K)]
    out = {}
    acc = p[0]
    for k in range(1, K): acc = fma(a[k], x[k], acc)
    out["chain_fma"] = acc
    acc = p[0]
    for k in range(1, K): acc = acc + p[k]
    out["seq_nofma"] = acc
    if K == 4:
        out["hadd"] = (p[0] + p[1]) + (p[2] + p[3])
        out["(p0+p2)+(p1+p3)"] = (p[0] + p[2]) + (p[1] + p[3])
        out["fma(2,p0)+fma(3,p1)"] = fma(a[2], x[2], p[0]) + fma(a[3], x[3], p[1])
        out["fma(1,p0)+fma(3,p2)"] = fma(a[1], x[1], p[0]) + fma(a[3], x[3], p[2])
    s = 0.0
    for k in range(K): s += float(p[k])
    out["f64sum_f32prod"] = f32(s)
    return out


def run(name, fn, m, k, n, iters=400):
    score = {}
    for _ in range(iters):
        A = rnd((m, k)); B = rnd((k, n))
        Y = np.asarray(fn(A, B)).reshape(m, n)
        for i in range(m):
            for j in range(n):
                for q, v in cands(A[i], B[:, j]).items(): score[q] = score.get(q, 0) + int(v == Y[i, j])
    tot = iters * m * n
    print("%-34s" % name, {q: v for q, v in score.items() if v == tot} or score, "of", tot)


run("gemv F(4x4).dot(x)", lambda A, B: A.dot(B[:, 0]), 4, 4, 1)
run("gemv C(2x4).dot(x)", lambda A, B: A.dot(B[:, 0]), 2, 4, 1)
run("gemv K(4x2).dot(dz)", lambda A, B: A.dot(B[:, 0]), 4, 2, 1)
run("gemv d(4).dot(M 4x4)", lambda A, B: A[0].dot(B), 1, 4, 4)
run("sdot d(4).dot(d)", lambda A, B: A[0].dot(B[:, 0]), 1, 4, 1)
run("gemm 4x4.4x4", lambda A, B: A.dot(B), 4, 4, 4, 200)
run("gemm 2x4.4x4", lambda A, B: A.dot(B), 2, 4, 4, 200)
run("gemm 4x4.(4x2 T view)", lambda A, B: A.dot(np.ascontiguousarray(B.T).T), 4, 4, 2, 200)
run("gemm 4x2.2x2", lambda A, B: A.dot(B), 4, 2, 2, 200)
run("gemm 4x2.2x4", lambda A, B: A.dot(B), 4, 2, 4, 200)
run("matmul batched (n,4,4)@(4,4)", lambda A, B: np.matmul(np.array([A] * 5), B)[3], 4, 4, 4, 200)
# float64: A(4x4 f32 -> f64).dot(x.T).T  (kalman.py:60) and (n,2,4)... the state chain
score = {}
for _ in range(300):
    A = rnd((4, 4)).astype(np.float64); X = (rng.normal(size=(7, 4)) * 1000)
    Y = A.dot(X.T).T
    for r in range(7):
        for i in range(4):
            acc = A[i, 0] * X[r, 0]
            for k in range(1, 4): acc = lib.c_fma(A[i, k], X[r, k], acc)
            s = A[i, 0] * X[r, 0]
            for k in range(1, 4): s = s + A[i, k] * X[r, k]
            score["chain_fma"] = score.get("chain_fma", 0) + int(acc == Y[r, i])
            score["seq_nofma"] = score.get("seq_nofma", 0) + int(s == Y[r, i])
print("dgemm A.dot(x.T).T (n=7)", score, "of", 300 * 28)
# float32 log (kalman.py:19 uses np.log on float32)
x = np.abs(rnd(4096)) + f32(0.5)
print("f32 log vs rounded f64 log: differ in", int(np.sum(np.log(x) != np.log(x.astype(np.float64)).astype(f32))), "of 4096")
