"""Build libasr_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m audio_sheet_retrieval_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so
that it travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libasr_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-unused-result", "-Wno-unused-value"]
EXTRA_FLAGS = os.environ.get("ASR_EXTRA_HIPCC_FLAGS", "").split()      # timing experiments (tools/ablate_*.sh): -DASR_WINOG_ABL=...
FLAGS = BASE_FLAGS + ["-I", INCLUDE] + EXTRA_FLAGS


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inl"))]   # (.inl: included sources)
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hdrs


def source_hash():
    """sha256 over every file the library is built from (csrc/*.hip|.h|.inl, include/*.h), 12 hex digits.
    asr_version() reports it; _lib.load_library() refuses a .so built from other sources."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inl"))]
    files += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    for p in sorted(files):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fp:
            h.update(fp.read())
    # the compile flags are part of what the binary is: an ablation build (-DASR_WINOG_ABL=..., wrong results by
    # design) left behind by an interrupted tools/ablate_*.sh must not pass for the default build.  (The include path
    # is machine dependent and not hashed.)
    h.update(("\0flags\0" + " ".join(BASE_FLAGS + EXTRA_FLAGS)).encode())
    return h.hexdigest()[:12]


HASH_STAMP = os.path.join(OBJ_DIR, "source_hash.txt")


def _built_hash():
    try:
        with open(HASH_STAMP) as fp:
            return fp.read().strip()
    except OSError:
        return None


def needs_build():
    if not os.path.exists(LIB):
        return True
    if _built_hash() != source_hash():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + _deps() + [os.path.abspath(__file__)])


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ and link libasr_hip.so. Returns its path."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    dep_t = max(os.path.getmtime(p) for p in _deps() + [os.path.abspath(__file__)])
    shash = source_hash()
    flags_stamp = os.path.join(OBJ_DIR, "flags.txt")
    flags_now = " ".join(BASE_FLAGS + EXTRA_FLAGS)
    try:
        with open(flags_stamp) as fp:
            force = force or fp.read().strip() != flags_now        # objects built with other flags are not reusable
    except OSError:
        force = force or bool(EXTRA_FLAGS)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        is_version = os.path.basename(src) == "asr_version.hip"      # carries the hash: rebuilt whenever it changes
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), dep_t) and \
                not (is_version and _built_hash() != shash):
            return obj
        cmd = [HIPCC] + FLAGS + (['-DASR_SOURCE_HASH="%s"' % shash] if is_version else []) + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(HASH_STAMP, "w") as fp:
        fp.write(shash + "\n")
    with open(flags_stamp, "w") as fp:
        fp.write(flags_now + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
