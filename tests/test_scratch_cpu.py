"""No scratch (private-memory) instruction inside a hot MFMA loop of the shipped library: a spill reload is a vector-memory operation whose
`s_waitcnt vmcnt(0)` also drains the LDS-DMA ring (the 256x256 implicit-GEMM convolutions paid two per K-tile until round 6; topk.hip one per round in
round 5).  tools/check_scratch.py disassembles the gfx950 code objects of motionrag_amd/libmrag_hip.so -- no GPU needed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "motionrag_amd", "libmrag_hip.so")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="llvm-objdump not installed")
def test_no_scratch_inside_hot_mfma_loops():
    from motionrag_amd import _lib
    _lib.build()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scratch.py"), LIB, "--json"], capture_output=True, text=True, timeout=600)
    rows = json.loads(res.stdout)
    assert rows, "the checker found no kernel with MFMAs"
    failing = [(r["kernel"], r["loops"]) for r in rows if r["status"] == "FAIL"]
    assert res.returncode == 0 and not failing, failing
    # the implicit-GEMM convolutions (CONV = 1 / 2 instantiations of gemm_bf16_kernel) and the persistent four-wave GEMM: nothing inside their K loops
    for r in rows:
        if "gemm_bf16_kernel<" in r["kernel"] or "gemm_w4_kernel<" in r["kernel"] or "topk_mfma_kernel<" in r["kernel"]:
            assert r["in_loop"] == 0, (r["kernel"], r["loops"])
