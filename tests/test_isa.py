"""The cache policy of the hot kernels' stores, read from the built library (no GPU needed).

`nt sc1` (non-temporal, write-through at agent scope) on the stores of the one-update kernels and of the
LDS kernel's last stage with fp32 storage is what took the headline kernel from 0.793 to 0.817 of the HBM
peak (profiles/r03_t1_layout.md section 0, profiles/r03_tb_hbm.md section 3d); written through, PARTIAL
lines are poison, so the LDS kernel with fp16 storage must keep plain stores.  hipcc has no builtin for
the policy (it is inline assembly, `st_stream` in csrc/lbm_device.h): this test is what notices if a
refactoring drops it."""
import os
import re
import shutil
import subprocess

import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "latticeboltzmann_amd", "liblbm_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    d = tmp_path_factory.mktemp("isa")
    so = shutil.copy(LIB, d)  # the code objects are extracted next to the file
    subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=d)
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(d, f)], check=True, capture_output=True,
                             text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                name = m.group(1)
                out.setdefault(name, [])
            elif name is not None and "global_store" in line:
                out[name].append(line.split("//")[0].strip())
    assert out, "no gfx950 code object in the library"
    return out


def _stores(kernels, pattern):
    hit = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert hit, f"no kernel matches {pattern}"
    return hit


@pytest.mark.parametrize("pattern,op", [
    (r"lbm::step_kernel<\d+, float>", "global_store_dword"),
    (r"lbm::step_kernel<\d+, __half>", "global_store_short"),
    (r"lbm::q21::step_kernel<", "global_store_dword"),
])
def test_one_update_kernels_store_write_through(kernels, pattern, op):
    for name, stores in _stores(kernels, pattern).items():
        assert len(stores) >= 9, name
        for s in stores:
            assert s.startswith(op) and s.endswith("nt sc1"), f"{name}: {s}"


def test_lds_kernel_streams_fp32_only(kernels):
    for name, stores in _stores(kernels, r"lbm::step_tb_kernel<.*float>").items():
        assert stores and all(s.endswith("nt sc1") for s in stores), name
    for name, stores in _stores(kernels, r"lbm::step_tb_kernel<.*__half>").items():
        assert stores and not any("sc1" in s or " nt" in s for s in stores), name
