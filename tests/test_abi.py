"""CPU tests of the drop-in boundary: libdispu_hip.so builds for gfx950, loads without a GPU, and exports
exactly the symbols include/dispu_hip.h declares; the ctypes table in dis-pu_amd/_lib.py covers them all.
No compute call is made here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("dispu_build", os.path.join(ROOT, "dis-pu_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.LIB) and not os.path.exists(mod.HIPCC):
        pytest.skip("no prebuilt library and no hipcc")
    return mod.build() if os.path.exists(mod.HIPCC) else mod.LIB


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dispu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dispu_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_exports(built_lib):
    syms = header_symbols()
    assert len(syms) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = sorted(set(re.findall(r" T (dispu_[a-z0-9_]+)", out)))
    assert exported == syms, (set(exported) ^ set(syms))


def test_library_loads_and_binding_is_complete(built_lib):
    import dispu_amd
    from dispu_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.lib()
    assert lib.dispu_version() == 5
    assert lib.dispu_fps_scratch_bytes(2, 1000, 10) == 0
    assert lib.dispu_fps_scratch_bytes(2, 30000, 10) == 2 * 30000 * 4
    assert lib.dispu_approx_match_scratch_bytes(3, 10, 20) == 3 * (20 * 30 + 2 * 10 + 20) * 4 + 4 * 4   # ratio vectors + chunk partials + stage counters
    assert lib.dispu_match_cost_scratch_bytes(2, 300, 200) == 2 * 2 * 13 * 4              # 16-partner tiles at this size
    assert lib.dispu_match_cost_grad_scratch_bytes(2, 300, 200) == 2 * 13 * 300 * 3 * 4
    assert isinstance(lib.dispu_error_string(1), bytes)


def test_only_gfx950_code_objects(built_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + built_lib],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    # fall back to strings when the bundler cannot parse a shared object
    text = out.stdout.decode(errors="replace")
    if "gfx" not in text:
        text = subprocess.run(["strings", built_lib], stdout=subprocess.PIPE).stdout.decode(errors="replace")
    archs = set(re.findall(r"gfx[0-9a-f]{3,4}", text))
    assert archs == {"gfx950"}, archs


def test_shims_validate_like_the_reference():
    """Shape errors carry the reference op's InvalidArgument text and fire before any device work."""
    import torch
    import dispu_amd.tf_sampling as S
    with pytest.raises(ValueError, match="must live on a ROCm device"):
        S.farthest_point_sample(4, torch.zeros(1, 8, 3))
    with pytest.raises(TypeError):
        S.farthest_point_sample(4, [[0, 0, 0]])
