"""CPU: the C-ABI library builds, loads, and exports every symbol include/lbm_hip.h declares; the
ctypes mirror of lbm_config matches the C layout; without a HIP device the product path fails
loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lbm_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from latticeboltzmann_amd import lb_hip_kernel as hk
    L = hk.load_library()
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(hk.SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", hk.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (lbm_[a-z_0-9]+)", out))
    assert set(declared) <= exported
    assert L.lbm_abi_version() == hk.ABI_VERSION == 3


def test_config_struct_layout_matches_c(tmp_path):
    from latticeboltzmann_amd import lb_hip_kernel as hk
    fields = [f[0] for f in hk.lbm_config._fields_]
    prog = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){',
             'printf("%zu\\n", sizeof(lbm_config));']
    lines += [f'printf("%zu\\n", offsetof(lbm_config, {f}));' for f in fields]
    lines += ['return 0;}']
    prog.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(prog)])
    vals = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert vals[0] == ctypes.sizeof(hk.lbm_config)
    assert vals[1:] == [getattr(hk.lbm_config, f).offset for f in fields]


def test_create_rejects_a_caller_built_against_another_abi():
    """sizeof(lbm_config) did not change when omega became a double, so struct_size cannot tell an
    ABI-1 caller apart; abi_version can (checked before any device is touched: runs without a GPU)."""
    from latticeboltzmann_amd import lb_hip_kernel as hk
    L = hk.load_library()
    cfg = hk.lbm_config()
    cfg.struct_size = ctypes.sizeof(hk.lbm_config)
    cfg.N, cfg.M, cfg.omega, cfg.inner_timesteps = 8, 8, 0.5, 1
    h = ctypes.c_void_p()
    for v in (0, 1, 2, 4):  # 0 = what a caller of ABI <= 2 leaves in the then-reserved field
        cfg.abi_version = v
        assert L.lbm_create(ctypes.byref(cfg), ctypes.byref(h)) == 1  # LBM_ERR_INVALID
        assert b"abi_version" in L.lbm_last_error() and not h.value
    cfg.abi_version = hk.ABI_VERSION
    cfg.struct_size -= 4
    assert L.lbm_create(ctypes.byref(cfg), ctypes.byref(h)) == 1 and b"struct_size" in L.lbm_last_error()


def test_header_is_plain_c():
    """The boundary is C: the header must compile as C99 with no HIP or torch types."""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert sorted(re.findall(r"#include\s*<([^>]+)>", code)) == ["stddef.h", "stdint.h"]
    assert not re.search(r"torch::|at::|hipStream_t|hipError_t", code)


def test_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from latticeboltzmann_amd import lb_hip_kernel as hk
    with pytest.raises(RuntimeError):
        hk.Solver(8, 8, 0.5)


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from latticeboltzmann_amd import lb_hip_kernel as hk
    monkeypatch.setattr(hk, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hk.load_library(str(tmp_path / "nope.so"))
