#!/usr/bin/env python3
"""developer tool: shader clock / power while the GEMM (or attention) runs back to back -- is the chip power-throttled under MFMA load?"""
import subprocess, sys, threading, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionrag_amd import ops

def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power", "power", "Temperature (Sensor junction)", "mclk"))]
    return " | ".join(keep)

print("idle:", smi())
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if which == "gemm":
    x = torch.randn(35552, 3072, device="cuda").to(torch.bfloat16); w = (torch.randn(9216, 3072, device="cuda") * 0.02).to(torch.bfloat16)
    f = lambda: ops.linear(x, w)
    flop = 2 * 35552 * 3072 * 9216
elif which == "matmul":
    x = torch.randn(35552, 3072, device="cuda").to(torch.bfloat16); w = (torch.randn(9216, 3072, device="cuda") * 0.02).to(torch.bfloat16)
    f = lambda: torch.nn.functional.linear(x, w)
    flop = 2 * 35552 * 3072 * 9216
else:
    q = torch.randn(2, 17776, 48, 64, device="cuda").to(torch.bfloat16)
    f = lambda: ops.attention(q, q, q)
    flop = 4 * 2 * 48 * 17776 * 17776 * 64
f(); torch.cuda.synchronize()
stop = False
def poll():
    while not stop:
        time.sleep(1.0); print("load:", smi(), flush=True)
t = threading.Thread(target=poll); t.start()
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 300 if which != "attn" else 80
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{which}: {flop * n / (e0.elapsed_time(e1) * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
stop = True; t.join()
