"""Builds libmht_amd.so (HIP, gfx950) in-tree with hipcc.  `python -m pymht_amd.build` or build_library()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmht_amd.so")
LIB6 = os.path.join(HERE, "libmht_amd6.so")      # the same sources with -DMHT_NX=6: the six-state build (BASELINE config 5's state dimension)


def lib_path(nx=4):
    assert nx in (4, 6), "libmht_amd is built for 4 or 6 states"
    # (development: MHT_LIB_VARIANT=.name loads libmht_amd.so.name, an experiment build made by hand)
    return (LIB if nx == 4 else LIB6) + os.environ.get("MHT_LIB_VARIANT", "")
SOURCES = ["mht_api.hip", "mht_gate.hip", "mht_gatex.hip", "mht_fgrow.hip", "mht_cluster.hip", "mht_blp.hip", "mht_prune.hip", "mht_similar.hip", "mht_init.hip", "mht_ais.hip", "mht_forest.hip"]
# -ffp-contract=off is REQUIRED: mht_math.h spells out every fused multiply-add of the reference's
# BLAS evaluation order; letting the compiler contract anything else breaks bit-exact gating.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


def _stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mht_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=True, nx=4):
    """Compile the library (nx = 4: libmht_amd.so, nx = 6: libmht_amd6.so) if it is missing or older than its sources.  Safe under
    concurrent callers (one process per GPU all importing the package at once): an exclusive file lock serialises them, the second one
    finds a fresh library; the output is written to a temporary file and renamed, so a reader never sees a half-written .so."""
    LIB = lib_path(nx)
    if os.environ.get("MHT_LIB_VARIANT") and os.path.exists(LIB):
        return LIB      # an experiment build made by hand (other flags, other sources): never rebuilt from the tree's sources
    if not force and not _stale(LIB):
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(LIB):
                return LIB
            hipcc = os.environ.get("HIPCC", "hipcc")
            srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
            extra = os.environ.get("MHT_EXTRA_HIPCC_FLAGS", "").split()      # development: e.g. -DMHT_GROW_STAMPS (tools/grow_profile.py)
            # one object per translation unit (no relocatable device code: the units share headers only), compiled side by side and kept
            # under build/: a change to one .hip recompiles that unit, a change to a header or to the flags recompiles all
            odir = os.path.join(HERE, "build", "nx%d%s" % (nx, os.environ.get("MHT_LIB_VARIANT", "")))
            os.makedirs(odir, exist_ok=True)
            cflags = [f for f in FLAGS if f != "-shared"] + (["-DMHT_NX=%d" % nx] if nx != 4 else []) + extra
            stamp = os.path.join(odir, "flags.txt")
            old_flags = open(stamp).read() if os.path.exists(stamp) else None
            hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
            hdr_t = max(hdr_t, os.path.getmtime(os.path.join(HERE, "..", "include", "mht_amd.h")))
            todo = []
            for s in srcs:
                o = os.path.join(odir, os.path.basename(s)[:-4] + ".o")
                if force or old_flags != " ".join(cflags) or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
                    todo.append((s, o))
            import concurrent.futures as cf

            def compile_one(so):
                cmd = [hipcc] + cflags + ["-c", so[0], "-o", so[1]]
                if verbose:
                    print("[pymht_amd.build]", " ".join(cmd), file=sys.stderr)
                subprocess.check_call(cmd)
            with cf.ThreadPoolExecutor(max(1, min(len(todo), os.cpu_count() or 4))) as ex:
                list(ex.map(compile_one, todo))
            with open(stamp, "w") as fh:
                fh.write(" ".join(cflags))
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(odir, os.path.basename(s)[:-4] + ".o") for s in srcs] + ["-o", tmp]
            if verbose:
                print("[pymht_amd.build]", " ".join(cmd), file=sys.stderr)
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def build_all(force=False, verbose=True):
    """Both builds, side by side (two hipcc processes)."""
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(2) as ex:
        return [f.result() for f in [ex.submit(build_library, force, verbose, nx) for nx in (4, 6)]]


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
