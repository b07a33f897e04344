"""CPU (cross-compile only): register / scratch budget of the hot kernels.  Twice this round a harmless-looking change made
the compiler stop inlining a device function or hoist loop invariants, and the kernel silently started to spill or to pass
its argument struct through scratch memory (1.6 us slower per ILP cluster; 7 MB of extra HBM writes per grow launch).
The compiler's own resource report is checked here so that this cannot come back unnoticed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pymht_amd", "csrc")

# kernel -> (max scratch bytes per lane, max VGPRs).  grow_kernel must stay at <= 128 registers: 4 wavefronts per SIMD keep two
# of its 512-thread workgroups per CU resident, which the look-back of its tile prefix relies on.
BUDGET = {
    "mht_gate.hip": {"grow_kernel": (0, 128)},
    # blp_uf_kernel = the ILP launch of the replay / streamed path since round 4 (cluster tables derived in its prologue): one 155 KB workgroup per
    # CU, so all 256 registers are its to use -- what must not come back is scratch (spilled arguments in front of every workgroup)
    "mht_blp.hip": {"blp_kernelE": (32, 256), "blp_uf_kernel": (0, 256)},
    # cluster_init_kernel / post_scan_kernel / initiator_side_kernel carry the M-of-N initiator (1024 threads: 128 registers).  Their 288 bytes of
    # scratch are the frame of ONE cold call: the general 4 x 4 elimination of inv_small (dynamic pivoting = dynamically indexed arrays), which
    # the reference's models never reach (mht_init_dev.h).  Until round 6 it was inlined at every call site and the kernels spilled 46-59
    # registers around it; what must not come back is a REGISTER spill (third entry: 0).
    "mht_cluster.hip": {"cluster_kernel": (0, 128), "cluster_init_kernel": (288, 128, 0)},
    "mht_forest.hip": {"commit_kernel": (0, 128), "add_targets_kernel": (0, 128), "post_scan_kernelILb0": (288, 128, 0), "post_scan_kernelILb1": (288, 128, 0),
                       "initiator_side_kernel": (288, 128, 0)},
    "mht_init.hip": {"initiator_kernel": (288, 128, 0)},
    # (fgrow_ais_kernel: two workgroups per CU -- 256 registers; the float64 chain of a promoted target's float32 leaves runs inline in it: its pivoted 2x2 / dynamic row exchanges take 400 B of scratch per lane)
    "mht_fgrow.hip": {"fgrow_kernel": (0, 168), "fgrow_batch_kernel": (0, 128), "fgrow_adm_kernel": (0, 168), "fgrow_ais_kernel": (512, 256)},      # 3 / 4 workgroups per CU
}


# the six-state build (-DMHT_NX=6, libmht_amd6.so): BASELINE config 5's constant-turn forest.  fgrow_ct_kernel: three workgroups per CU
# (launch bounds; 147-150 registers measured), forest_ct_kernel (per-leaf Phi(T, w) + covariance chain in front of the grow launch): two
# wavefronts per SIMD at 189 registers -- neither may spill
BUDGET6 = {
    "mht_fgrow.hip": {"fgrow_ct_kernel": (0, 168), "fgrow_kernelILi2ELi128": (0, 168)},      # (the second: the one-sector grow kernel with the linear six-state stand-in, models/ca.py)
    "mht_ais.hip": {"forest_ct_kernel": (0, 200)},
}


def _report(src, tmp_path, extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from pymht_amd.build import FLAGS
    flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [hipcc] + flags + list(extra) + ["-c", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include"),
                                           os.path.join(CSRC, src), "-o", str(tmp_path / "o.o")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr
    found = {}
    for m in re.finditer(r"Function Name: (\S+)", text):
        seg = text[m.end():m.end() + 4000]
        nxt = seg.find("Function Name:")
        seg = seg if nxt < 0 else seg[:nxt]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", seg).group(1))
        vgpr = int(re.search(r"VGPRs: (\d+)", seg).group(1))
        found[m.group(1)] = (scratch, vgpr)
    return found


@pytest.mark.parametrize("src", sorted(BUDGET6))
def test_scratch_and_register_budget_six_state_build(src, tmp_path):
    found = _report(src, tmp_path, ["-DMHT_NX=6"])
    for kern, (max_scratch, max_vgpr) in BUDGET6[src].items():
        hits = [v for k, v in found.items() if kern in k]
        assert hits, "kernel %s not found in the compiler report of %s (-DMHT_NX=6)" % (kern, src)
        for scratch, vgpr in hits:
            assert scratch <= max_scratch, "%s (six-state build) uses %d B of scratch per lane (budget %d)" % (kern, scratch, max_scratch)
            assert vgpr <= max_vgpr, "%s (six-state build) needs %d VGPRs (budget %d)" % (kern, vgpr, max_vgpr)


@pytest.mark.parametrize("src", sorted(BUDGET))
def test_scratch_and_register_budget(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from pymht_amd.build import FLAGS
    flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [hipcc] + flags + ["-c", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include"),
                             os.path.join(CSRC, src), "-o", str(tmp_path / "o.o")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr
    found = {}
    for m in re.finditer(r"Function Name: (\S+)", text):
        seg = text[m.end():m.end() + 4000]
        nxt = seg.find("Function Name:")
        seg = seg if nxt < 0 else seg[:nxt]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", seg).group(1))
        vgpr = int(re.search(r"VGPRs: (\d+)", seg).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", seg).group(1))
        found[m.group(1)] = (scratch, vgpr, spill)
        if "initiator_side_kernel" in m.group(1):
            # the initiator runs NEXT to the ILP launch: its workgroup must fit the LDS a 155 KB ILP workgroup leaves of a CU (160 KB), or it waits for that launch to drain
            lds_b = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", seg).group(1))
            assert lds_b <= 8192, "initiator_side_kernel uses %d bytes of LDS (8 192 fit next to an ILP workgroup)" % lds_b
    for kern, budget in BUDGET[src].items():
        max_scratch, max_vgpr = budget[:2]
        hits = [v for k, v in found.items() if kern in k]
        assert hits, "kernel %s not found in the compiler report of %s" % (kern, src)
        scratch, vgpr, spill = hits[0]
        if len(budget) > 2:
            assert spill <= budget[2], "%s spills %d VGPRs (budget %d)" % (kern, spill, budget[2])
        assert scratch <= max_scratch, "%s uses %d B of scratch per lane (budget %d): a device function stopped being inlined or registers spill" % (kern, scratch, max_scratch)
        assert vgpr <= max_vgpr, "%s needs %d VGPRs (budget %d)" % (kern, vgpr, max_vgpr)
