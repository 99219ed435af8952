"""The persistent four-wave GEMM (csrc/gemm_bf16.hip: gemm_w4_kernel) retires its LDS-DMA ring with COUNTED `s_waitcnt vmcnt(N)` written by hand: the count is only
right while hipcc puts no vector-memory instruction and no wait of its own into the K loop.  It did, twice, during development (a spill reload's `s_waitcnt vmcnt(0)`
parked in the loop header drained the ring once per K-tile: +33 % K-loop time, nothing wrong in the results), so the compiled ISA is checked here: no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_four_wave_gemm_k_loop_is_hand_scheduled_only(tmp_path):
    asm = tmp_path / "gemm.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(ROOT, "motionrag_amd", "csrc", "gemm_bf16.hip"), "-o", str(asm)],
                   check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN\S*gemm_w4_kernelILi(\d+)E[^:\s]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert {int(k[1]) for k in kernels} >= {0, 1, 3, 4, 7}, "every epilogue the DiT runs has a four-wave instantiation"
    for name, epi, body in kernels:
        lines = body.split("\n")
        mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32_bf16" in l]
        assert len(mfma) == 128, (epi, len(mfma))                     # ONE K-tile body (two 64-MFMA k-steps): one register allocation for the 256 accumulators
        in_asm, bad, dma, reads, barriers = False, [], 0, 0, 0
        for l in lines[mfma[0]:mfma[-1] + 1]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";"):
                continue
            op = t.split()[0]
            if in_asm:
                dma += op == "global_load_lds_dwordx4"
                reads += op == "ds_read_b128"
                barriers += op == "s_barrier"
            elif op.startswith(("scratch_", "global_", "buffer_", "flat_")) or (op == "s_waitcnt" and "vmcnt" in t):
                bad.append(t)                                           # compiler-made memory traffic or a compiler-made vmcnt wait inside the K loop
        assert not bad, (epi, bad)
        assert (dma, reads, barriers) == (16, 32, 2), (epi, dma, reads, barriers)
        assert "a[252:255]" in body                                    # the accumulators live in AGPRs
