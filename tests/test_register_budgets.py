"""Round 6: register budgets are part of the design.  A device function inherits the launch bound of the kernel that
calls it; the CCALayer's Jacobi routine keeps 64 float64 values per lane in registers, and compiled under
cca_train_kernel's 1024-thread bound (128 VGPRs) it went through 34 scratch round trips in every rotation round - the two
launches took 219 + 201 us instead of 113 + 96.  This test compiles the file for gfx950 (hipcc cross-compiles without a
GPU, ~20 s) and reads the register / scratch figures of the listing; tools/scratch_scan.sh does the same for every
kernel of the library."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _listing(tmp_path, name):
    out = str(tmp_path / (name + ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-S",
           "--cuda-device-only", "-Wno-comment", "-Wno-unused-result", "-Wno-unused-value",
           os.path.join(ROOT, "audio_sheet_retrieval_amd", "csrc", name + ".hip"), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def _functions(lines):
    """{mangled name: (vgprs, scratch bytes, scratch loads inside loops)}"""
    res, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_Z[\w$.]+):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        tail = lines[j:j + 80]
        num = lambda key: next((int(re.search(r"(\d+)", l.split(":")[1]).group(1)) for l in tail if key in l), 0)
        depth, inloop = 0, 0
        for l in lines[i:j]:
            if l.startswith(".LBB"):
                mm = re.search(r"Depth=(\d+)", l)
                depth = int(mm.group(1)) if mm else 0
            if "scratch_load" in l and depth >= 1:
                inloop += 1
        res[m.group(1)] = (num("; NumVgprs:"), num("; ScratchSize:"), inloop)
        i = j
    return res


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_the_jacobi_routine_is_compiled_with_the_registers_it_needs(tmp_path):
    fn = _functions(_listing(tmp_path, "cca_train_kernel"))
    eigh = [v for k, v in fn.items() if "eigh_spd" in k or "cca_eigh_kernel" in k]
    assert len(eigh) >= 1, sorted(fn)
    for vgprs, scratch, inloop in eigh:
        assert vgprs > 128, (vgprs, scratch)              # more than a 1024-thread bound allows: the bound is 256
        assert inloop == 0 and scratch <= 32, (vgprs, scratch, inloop)
    # the 32x32 algebra phases stay under their 1024-thread bound without reloads inside their loops
    train = [v for k, v in fn.items() if "cca_train_kernel" in k]
    assert train and all(v[0] <= 128 and v[2] == 0 for v in train), train


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_the_raw_winograd_builds_of_the_training_step_do_not_spill_into_their_loops(tmp_path):
    """The switched-off BatchNorm-backward epilogue of round 5 kept four per-channel constants live across the M-tile
    loops of every RAW (training) Winograd build: 68-136 bytes of scratch per lane, 0.2 ms of the batch-512 update.  It
    is compiled out by default (ASR_BNB_FUSE_BUILD); the RAW builds of conv3x3_wino and conv3x3_wino4s use no scratch at
    all, the two-waves-per-SIMD RAW builds of conv3x3_winog at most three reloads per M-tile, and the dominant inference
    kernels none that is waited for at once."""
    w = _functions(_listing(tmp_path, "conv_wino_kernels"))
    w4 = _functions(_listing(tmp_path, "conv_wino4_kernels"))
    raw_wino = {k: v for k, v in w.items() if "12conv3x3_winoI" in k and "ELb1ELi0ELi0EEEv" in k}     # RAW = true, PW = 0
    assert raw_wino, [k for k in w if "conv3x3_winoI" in k][:3]
    assert all(v[1] == 0 for v in raw_wino.values()), {k[:60]: v for k, v in raw_wino.items() if v[1]}
    raw_w4 = {k: v for k, v in w4.items() if "conv3x3_wino4sI" in k and "ELb1EEEv" in k}               # RAW = true
    assert raw_w4 and all(v[1] == 0 for v in raw_w4.values()), {k[:60]: v for k, v in raw_w4.items() if v[1]}
    raw_wg = {k: v for k, v in w.items() if "13conv3x3_winogI" in k and "ELb1EEEv" in k}
    assert raw_wg and all(v[1] <= 48 and v[2] <= 3 for v in raw_wg.values()), {k[:60]: v for k, v in raw_wg.items() if v[1]}
    inf_wg = {k: v for k, v in w.items() if "13conv3x3_winogI" in k and "ELb0EEEv" in k}
    assert inf_wg and all(v[1] <= 16 and v[2] <= 1 for v in inf_wg.values()), {k[:60]: v for k, v in inf_wg.items() if v[1]}
