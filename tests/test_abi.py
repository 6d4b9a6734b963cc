"""CPU: the C-ABI library builds, loads, and exports every symbol that
include/asr_hip.h declares; without a GPU the product path fails loudly
(no CPU fallback, no oracle import)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest


def _declared(repo_root):
    text = open(os.path.join(repo_root, "include", "asr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path(repo_root):
    from audio_sheet_retrieval_amd import build
    return build.build(force=False, verbose=False)


def test_header_symbols_are_exported(repo_root, lib_path):
    from audio_sheet_retrieval_amd import _lib
    declared = _declared(repo_root)
    assert len(declared) >= 20
    assert sorted(_lib.EXPORTS) == declared, "python binding list and header disagree"
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), "libasr_hip.so does not export %s" % name
    _lib.load_library()          # prototypes resolve
    assert b"gfx950" in lib_version(lib)


def lib_version(lib):
    lib.asr_version.restype = ctypes.c_char_p
    return lib.asr_version()


def test_code_object_targets_gfx950(lib_path):
    """the fat binary embedded in the .so carries a gfx950 code object"""
    blob = open(lib_path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_no_gpu_fails_loudly(lib_path):
    """Engine() must raise, not silently fall back, when no device exists."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from audio_sheet_retrieval_amd import _lib\n"
            "lib = _lib.load_library()\n"
            "import ctypes\n"
            "try:\n"
            "    _lib.Engine('mutopia_ccal_cont')\n"
            "    print('CREATED')\n"
            "except _lib.AsrError as e:\n"
            "    print('RAISED', e.code)\n"
            "print('oracle' in sys.modules)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    lines = out.stdout.strip().splitlines()
    assert lines and lines[0].startswith("RAISED"), out.stdout + out.stderr
    assert lines[-1] == "False", "product path imported the oracle"


def test_missing_library_is_an_import_error(tmp_path):
    from audio_sheet_retrieval_amd import _lib
    with pytest.raises(_lib.AsrLibraryError):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_config_struct_matches_header(repo_root):
    from audio_sheet_retrieval_amd import _lib
    text = open(os.path.join(repo_root, "include", "asr_hip.h")).read()
    body = text[text.index("typedef struct asr_config {"):text.index("} asr_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [f[0] for f in _lib.AsrConfig._fields_]
    assert ctypes.sizeof(_lib.AsrConfig) == 4 * len(names)
