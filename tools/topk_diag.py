"""developer tool: the fan-out top-k kernel's time split (run with MRAG_HIP_LIB=tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 for the in-kernel counters)"""
import ctypes, sys
sys.path.insert(0, "tools")
import torch, microbench as mb
from motionrag_amd import ops, _lib
L = _lib.lib()
stats = hasattr(L, "mrag_debug_topk_stats")
for N, Q in ((10000, 256), (1000000, 256), (1000000, 64)):
    db = torch.randn(N, 768, device="cuda"); q = torch.randn(Q, 768, device="cuda")
    ops.topk(db, q, 12, order="mfma"); torch.cuda.synchronize()
    if stats:
        buf = (ctypes.c_uint64 * 4)(); L.mrag_debug_topk_stats(buf, 1)
        ops.topk(db, q, 12, order="mfma"); torch.cuda.synchronize()
        L.mrag_debug_topk_stats(buf, 1)
        r, sel, tot, nb = (int(x) for x in buf)
        print(f"N={N} Q={Q}: {nb} row blocks, {r} rounds ({r / max(nb, 1):.2f} per block), selection {sel / max(nb,1):.0f} clock ticks per block ({100.0 * sel / max(tot, 1):.1f} % of kernel time), {sel / max(r, 1):.0f} per round")
    dt = mb.timeit(lambda: ops.topk(db, q, 12, order="mfma"), iters=5)
    print(f"N={N} Q={Q}: {dt*1e6:.1f} us  {2.0*N*Q*768/dt/1e12:.1f} TF")
