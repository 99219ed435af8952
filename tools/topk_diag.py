"""developer tool: the fan-out top-k kernel's time split (run with MRAG_HIP_LIB=tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 for the in-kernel counters)"""
import ctypes, sys
sys.path.insert(0, "tools")
import torch, microbench as mb
from motionrag_amd import ops, _lib
L = _lib.lib()
stats = hasattr(L, "mrag_debug_topk_stats")
for N, Q in ((10000, 256),) if hasattr(L, "mrag_debug_topk_dense_stats") and "--all" not in sys.argv else ((10000, 256), (1000000, 256), (1000000, 64)):
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

# the one-launch form (topk_dense_kernel) at BASELINE config #1's size: phase durations from in-kernel s_memtime stamps (100 MHz ticks -> us)
if hasattr(L, "mrag_debug_topk_dense_stats"):
    db = torch.randn(10000, 768, device="cuda"); q = torch.randn(256, 768, device="cuda")
    ops.topk(db, q, 12, order="mfma"); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 24)(); L.mrag_debug_topk_dense_stats(buf, 1)
    reps = 20
    for _ in range(reps):
        ops.topk(db, q, 12, order="mfma")
    torch.cuda.synchronize()
    L.mrag_debug_topk_dense_stats(buf, 1)
    b = [int(x) for x in buf]
    wg, nq = max(b[0], 1), max(b[8], 1)
    tick = 0.01
    print(f"one-launch form, {b[0] // reps} workgroups, {b[8] // reps} query finishes per call; mean us per workgroup: stream {b[1]/wg*tick:.1f}, scores + stores {b[2]/wg*tick:.1f}, arrive + wait {b[3]/wg*tick:.1f}")
    print(f"   slowest workgroup: stream {b[4]*tick:.1f} us, stores done at {b[5]*tick:.1f}, go seen at {b[6]*tick:.1f} (max over calls)")
    print(f"   finishing, mean us per query: bound {b[9]/nq*tick:.2f}, listing {b[10]/nq*tick:.2f}, score loads {b[11]/nq*tick:.2f}, ordering {b[12]/nq*tick:.2f}, second scoring + output {b[13]/nq*tick:.2f}; slowest {b[14]*tick:.1f}")
    print(f"   bound = minima loads {b[15]/nq*tick:.2f} + rank count {b[16]/nq*tick:.2f} + barrier {b[17]/nq*tick:.2f}")
