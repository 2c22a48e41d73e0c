"""The MFMA GEMM kernels must fit their registers: no instantiation of gemm_f64_kernel may spill (DESIGN.md 3.1 says "no
spills"; round 3's build had one spilled VGPR in the K loop of the two ragged three-multiplication op = C instantiations).
Cross-compiles the file for gfx950 with the compiler's resource-usage remarks - no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_gemm_instantiation_spills(tmp_path):
    src = os.path.join(ROOT, "chase_amd", "csrc", "gemm_mfma_f64.hip")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "chase_amd", "csrc"), "-c", src, "-o",
                        str(tmp_path / "gemm.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", p.stderr)[1:]
    kernels = {}
    for b in blocks:
        name = b.split()[0]
        get = lambda key: int(re.search(re.escape(key) + r": (\d+)", b).group(1))
        kernels[name] = (get("VGPRs"), get("ScratchSize [bytes/lane]"), get("VGPRs Spill"), get("SGPRs Spill"),
                         get("Occupancy [waves/SIMD]"))
    gemms = {k: v for k, v in kernels.items() if "gemm_f64_kernel" in k}
    assert len(gemms) >= 32                                              # real / complex x op x tag x ragged x 3M x narrow
    bad = {k: v for k, v in gemms.items() if v[1] or v[2] or v[3]}
    assert not bad, bad
    assert all(v[0] <= 256 and v[4] >= 2 for v in gemms.values())        # two waves per SIMD = two workgroups per CU
    shutil.rmtree(tmp_path, ignore_errors=True)
