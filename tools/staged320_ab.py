"""developer tool: the 256x320 tile's LDS-staged epilogue against its direct 8-byte-store form (tuning bit MRAG_GEMM_TUNE_NO_STAGED) on the UNets' level-0
shapes, bit-equality checked.   python tools/staged320_ab.py"""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0):
    return (torch.randn(*s, generator=g) * sc).to(DEV, torch.bfloat16)
M = 258048
cases = []
x1280, w320_1280, b320, r320 = rnd(M, 1280), rnd(320, 1280, sc=0.03), rnd(320), rnd(M, 320)
cases.append(("linear  [258048 x 320 x 1280] + resid", lambda: ops.linear(x1280, w320_1280, b320, epilogue=ops.EPI_RESID, resid=r320)))
x640, w320_640 = rnd(M, 640), rnd(320, 640, sc=0.04)
cases.append(("linear  [258048 x 320 x 640]", lambda: ops.linear(x640, w320_640, b320)))
xc = rnd(28, 72, 128, 320)
wk3 = rnd(320, 9 * 320, sc=0.02)
rc = rnd(28, 72, 128, 320)
cases.append(("conv3x3 [258048 x 320 x 2880]", lambda: ops.conv_implicit(xc, wk3, b320, ops.CONV_3X3)))
cases.append(("conv3x3 [258048 x 320 x 2880] + resid", lambda: ops.conv_implicit(xc, wk3, b320, ops.CONV_3X3, resid=rc)))
xt = rnd(28, 9216, 320)
wkt = rnd(320, 3 * 320, sc=0.03)
rt = rnd(28, 9216, 320)
cases.append(("conv_t3 [258048 x 320 x 960]", lambda: ops.conv_implicit(xt, wkt, b320, ops.CONV_T3, frames=14)))
cases.append(("conv_t3 [258048 x 320 x 960] + resid", lambda: ops.conv_implicit(xt, wkt, b320, ops.CONV_T3, frames=14, resid=rt)))
x960, w960 = rnd(M, 320), rnd(960 + 320, 320, sc=0.05)
for name, fn in cases:
    outs, ts = {}, {}
    for tag, t in (("staged", 0), ("direct", 2)):
        ops.TUNING["gemm"] = t
        with ops.dispatched() as d:
            outs[tag] = fn()
        ts[tag] = mb.timeit(fn, iters=10, warm=2)
    ops.TUNING["gemm"] = 0
    print(f"{name:42s} staged {ts['staged']*1e6:7.1f} us  direct {ts['direct']*1e6:7.1f} us  {'bit-equal' if torch.equal(outs['staged'], outs['direct']) else 'DIFFERENT'}  [{','.join(d.counts)}]", flush=True)
