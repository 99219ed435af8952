#!/usr/bin/env python3
"""Kernel micro-benchmarks on the BASELINE shapes (developer tool; bench.py is the judged entry point).
usage: python tools/microbench.py [attn] [gemm] [topk] [norm]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402

DEV = "cuda"

# developer knobs: the library reads no environment; this TOOL maps its historical env names onto the explicit tuning fields
_E = os.environ
ops.TUNING["gemm"] = ((ops.GEMM_TUNE_NO_WIDE if _E.get("MRAG_GEMM_NO_WIDE") else 0) | (ops.GEMM_TUNE_NO_STAGED if _E.get("MRAG_GEMM_NO_STAGED") else 0)
                      | (ops.GEMM_TUNE_GEGLU_NO_STAGED if _E.get("MRAG_GEGLU_NO_STAGED") else 0) | (int(_E.get("MRAG_GEMM_CFG", "0")) << 4)
                      | (int(_E.get("MRAG_GEMM_GROUP_M", "0")) << 8))
ops.TUNING["attn"] = (ops.ATTN_TUNE_NO_TINY if _E.get("MRAG_ATTN_NO_TINY") else 0) | (ops.ATTN_TUNE_LEGACY if _E.get("MRAG_ATTN_LEGACY") else 0)
ops.TUNING["attn_no_split"] = _E.get("MRAG_ATTN_KV_SPLITS") == "0"
ops.TUNING["no_qkv_fuse"] = bool(_E.get("MRAG_NO_QKV_FUSE"))


def timeit(fn, iters=10, warm=2):
    """HIP-event time per call.  The cyclic garbage collector is off inside the timed region, as in the standard library's `timeit`: every launch allocates a
    ctypes argument struct, and in a process that also holds a 5.6 B-parameter model (bench.py's secondary workloads) a generation-2 collection stalls the host for
    milliseconds -- the launch-heavy UNet steps then measure the collector, not the GPU (SVD step: 125 ms standalone, 141-146 ms inside bench.py before this)."""
    import gc
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
    finally:
        if was:
            gc.enable()
    return e0.elapsed_time(e1) / iters * 1e-3


def attn():
    B, H, S = 2, 48, 17776
    qkv = torch.randn(B, S, 3, H, 64, device=DEV).to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device=DEV, dtype=torch.bfloat16)
    dt = timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out))
    fl = 4.0 * B * H * S * S * 64
    print(f"attn joint S={S}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s  ({fl/dt/2.5e15*100:.1f}% of 2.5 PF)")
    kv = torch.randn(B, 25, 2, H, 64, device=DEV).to(torch.bfloat16)
    dt = timeit(lambda: ops.attention(qkv[:, :, 0], kv[:, :, 0], kv[:, :, 1], out=out, resid=out, out_scale=1.0))
    by = 3.0 * B * S * H * 64 * 2
    print(f"attn motion (25 keys, fused residual): {dt*1e6:.1f} us  {by/dt/1e9:.0f} GB/s algorithmic (read Q, read+write O)")


def gemm():
    M = 2 * 17776
    for name, N, K, epi in (("qkv", 9216, 3072, ops.EPI_NONE), ("to_out/to_q_ip", 3072, 3072, ops.EPI_NONE), ("ff1 gelu", 12288, 3072, ops.EPI_GELU_TANH),
                            ("ff2", 3072, 12288, ops.EPI_NONE)):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        dt = timeit(lambda: ops.linear(x, w, b, out=out, epilogue=epi))
        fl = 2.0 * M * N * K
        print(f"gemm {name:16s} M={M} N={N} K={K}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/2.5e15*100:.1f}%)")


def ipfold():
    """folded motion-adapter branch at the BASELINE shape: scores GEMM [S, 3072] x [3072, 1536] per sample + finishing kernel"""
    B, S, H = 2, 17776, 48
    o = torch.randn(B, S, H * 64, device=DEV).to(torch.bfloat16)
    M = (torch.randn(B, H * 32, H * 64, device=DEV) * 0.02).to(torch.bfloat16)
    v = torch.randn(B, 25, H * 64, device=DEV).to(torch.bfloat16)
    sc = torch.empty(B, S, H * 32, device=DEV, dtype=torch.bfloat16)

    def gemms():
        for b in range(B):
            ops.linear(o[b], M[b], out=sc[b])
    dt = timeit(gemms)
    print(f"ipfold scores GEMMs: {dt*1e3:.3f} ms  {2.0*B*S*H*64*H*32/dt/1e12:.0f} TFLOP/s")
    dt = timeit(lambda: ops.ip_attn_folded_(sc, v, o, H, 25))
    print(f"ipfold finishing kernel: {dt*1e6:.1f} us  {(sc.numel()+2*o.numel())*2/dt/1e9:.0f} GB/s (scores read, hidden read+write)")
    wq = (torch.randn(H * 64, H * 64, device=DEV) * 0.02).to(torch.bfloat16)
    dt = timeit(lambda: ops.linear(o, wq))
    print(f"literal to_q_ip GEMM: {dt*1e3:.3f} ms")


def hbm5():
    """round 5's memory-bound kernels at their BASELINE shapes (tools/pmc_traffic_r5.sh profiles this target): the folded motion branch, the K = 320 linear with the
    weight in registers (+ residual), the narrow-row LayerNorm, the fan-out top-k"""
    B, S, H = 2, 17776, 48
    o = torch.randn(B, S, H * 64, device=DEV).to(torch.bfloat16)
    v = torch.randn(B, 25, H * 64, device=DEV).to(torch.bfloat16)
    sc = torch.randn(B, S, H * 32, device=DEV).to(torch.bfloat16)
    dt = timeit(lambda: ops.ip_attn_folded_(sc, v, o, H, 25))
    print(f"ip_attn_folded [2 x 17776 x 48 heads]: {dt*1e6:.1f} us  {(sc.numel()+2*o.numel())*2/dt/1e9:.0f} GB/s algorithmic")
    M = 258048
    x = torch.randn(M, 320, device=DEV).to(torch.bfloat16); w = (torch.randn(320, 320, device=DEV) * 0.05).to(torch.bfloat16)
    r = torch.randn(M, 320, device=DEV).to(torch.bfloat16); bias = torch.randn(320, device=DEV).to(torch.bfloat16)
    dt = timeit(lambda: ops.linear(x, w, bias, epilogue=ops.EPI_RESID, resid=r))
    print(f"gemm_k320 [258048 x 320 x 320] + resid: {dt*1e6:.1f} us  {3*M*320*2/dt/1e9:.0f} GB/s algorithmic")
    g = torch.ones(320, device=DEV, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    dt = timeit(lambda: ops.layernorm(x, g, g, 1e-5, out=y))
    print(f"layernorm_rows [258048 x 320]: {dt*1e6:.1f} us  {2*M*320*2/dt/1e9:.0f} GB/s algorithmic")
    db = torch.randn(1000000, 768, device=DEV); q = torch.randn(256, 768, device=DEV)
    dt = timeit(lambda: ops.topk(db, q, 12), iters=3)
    print(f"topk fan-out [10^6 x 768] x 256 queries: {dt*1e6:.1f} us  {db.numel()*4/dt/1e9:.0f} GB/s of table per pass, {2.0*1e6*256*768/dt/1e12:.1f} TFLOP/s fp32 MFMA")


def ceilings():
    """the box's practical ceilings: vendor-library bf16 GEMM (hipBLASLt through torch) and device-to-device copy bandwidth"""
    M, N, K = 35552, 9216, 3072
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16); w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    dt = timeit(lambda: torch.nn.functional.linear(x, w), iters=50, warm=10)
    print(f"hipBLASLt bf16 GEMM M={M} N={N} K={K}: {2.0*M*N*K/dt/1e12:.0f} TFLOP/s (sustained over 50 launches)")
    dt = timeit(lambda: ops.linear(x, w), iters=50, warm=10)
    print(f"this repo's GEMM, same shape:            {2.0*M*N*K/dt/1e12:.0f} TFLOP/s")
    a = torch.empty(1 << 30, dtype=torch.uint8, device=DEV); b = torch.empty_like(a)
    dt = timeit(lambda: b.copy_(a), iters=20, warm=3)
    print(f"device-to-device copy of 1 GiB: {2 * a.numel() / dt / 1e12:.2f} TB/s (read + write)")
    res = {"hipblaslt_bf16_gemm_tflops": None}
    return res


def gemm320():
    """UNet level-0 shapes (28 frames x 72 x 128 rows, widths that are multiples of 320)"""
    M = 28 * 9216
    for name, N, K, epi in (("proj/to_out", 320, 320, ops.EPI_NONE), ("ff2", 320, 1280, ops.EPI_RESID), ("qkv", 960, 320, ops.EPI_NONE),
                            ("level-1 to_out", 640, 640, ops.EPI_NONE), ("geglu proj", 2560, 320, ops.EPI_GEGLU)):
        Mm = M if N != 640 else M // 4
        x = torch.randn(Mm, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device=DEV).to(torch.bfloat16)
        out = torch.empty(Mm, N // 2 if epi == ops.EPI_GEGLU else N, device=DEV, dtype=torch.bfloat16)
        r = torch.randn(Mm, N, device=DEV).to(torch.bfloat16) if epi == ops.EPI_RESID else None
        dt = timeit(lambda: ops.linear(x, w, b, out=out, epilogue=epi, resid=r), iters=30, warm=5)
        fl = 2.0 * Mm * N * K
        print(f"gemm {name:16s} M={Mm} N={N} K={K}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s  ({(Mm*K+Mm*N*(2 if r is not None else 1))*2/dt/1e9:.0f} GB/s of activations)")
    x = torch.randn(28, 72, 128, 320, device=DEV).to(torch.bfloat16)
    wk = (torch.randn(320, 9 * 320, device=DEV) * 0.02).to(torch.bfloat16)
    b = torch.randn(320, device=DEV).to(torch.bfloat16)
    dt = timeit(lambda: ops.conv_implicit(x, wk, b, ops.CONV_3X3), iters=30, warm=5)
    print(f"conv3x3 320->320 on [28,72,128]: {dt*1e3:.3f} ms  {2.0*M*320*2880/dt/1e12:.1f} TFLOP/s")


def gemm_epi():
    """the DiT's fused-epilogue GEMMs: gate * out + resid (to_out / FF2) and QKV + qk-norm + RoPE"""
    B, S, D = 2, 17776, 3072
    x = torch.randn(B, S, D, device=DEV).to(torch.bfloat16)
    res = torch.randn(B, S, D, device=DEV).to(torch.bfloat16)
    gate = torch.randn(B, 2 * D, device=DEV).to(torch.bfloat16)
    w = (torch.randn(D, D, device=DEV) * 0.02).to(torch.bfloat16); bias = torch.randn(D, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(res)
    dt = timeit(lambda: ops.linear(x, w, bias, out=out, epilogue=ops.EPI_GATE_RESID, resid=res, gate0=gate[:, :D], gate1=gate[:, D:], rows_per_batch=S, split=226,
                                   gate_stride=gate.stride(0)))
    print(f"gemm to_out gate+resid N=3072 K=3072: {dt*1e3:.3f} ms  {2.0*B*S*D*D/dt/1e12:.0f} TFLOP/s")
    wq = (torch.randn(3 * D, D, device=DEV) * 0.02).to(torch.bfloat16); bq = torch.randn(3 * D, device=DEV).to(torch.bfloat16)
    g = torch.ones(64, device=DEV, dtype=torch.bfloat16); bb = torch.zeros(64, device=DEV, dtype=torch.bfloat16)
    cos = torch.rand(S - 226, 64, device=DEV); sin = torch.rand(S - 226, 64, device=DEV)
    dt = timeit(lambda: ops.qkv_linear_qknorm_rope(x, wq, bq, 48, g, bb, g, bb, cos, sin, 226, q_premul=0.18))
    print(f"gemm qkv + qknorm + rope N=9216 K=3072: {dt*1e3:.3f} ms  {2.0*B*S*3*D*D/dt/1e12:.0f} TFLOP/s")


def mfma_f32_sustained(seconds=0.5):
    """measured ceiling (not product path): what the fp32 matrix pipe sustains here -- libmrag_hip.so's register-resident v_mfma_f32_32x32x2_f32 loop,
    four waves per CU like the fan-out kernel, random operands, back-to-back launches for `seconds`"""
    import ctypes
    from motionrag_amd import _lib
    L = _lib.lib()
    operands = torch.randn(1 << 20, device=DEV)
    sink = torch.empty(256 * 256, dtype=torch.float32, device=DEV)
    iters = 4000
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    launch = lambda: _lib.check(L.mrag_probe_mfma_f32(st, ctypes.c_void_p(operands.data_ptr()), operands.numel() * 4, ctypes.c_void_p(sink.data_ptr()), iters),  # noqa: E731
                                "mrag_probe_mfma_f32")
    one = timeit(launch, iters=2, warm=1)
    dt = timeit(launch, iters=max(2, int(seconds / max(one, 1e-4))), warm=0)
    return L.mrag_probe_mfma_f32_flops(iters) / dt / 1e12


def topk(cases=((10000, 1), (10000, 256), (1000000, 1), (1000000, 64))):
    import time
    res = {}
    if any(Q >= 16 for _, Q in cases):
        res["fp32_mfma_sustained_tflops"] = round(mfma_f32_sustained(), 1)
        print(f"fp32 MFMA probe (v_mfma_f32_32x32x2_f32, 4 waves per CU, register-resident): {res['fp32_mfma_sustained_tflops']} TFLOP/s sustained of 157 nominal")
    for N, Q in cases:
        db = torch.randn(N, 768, device=DEV)
        q = torch.randn(Q, 768, device=DEV)
        small = N <= 100000                                                  # (a search of tens of microseconds: enough calls that the queue, not the first launches, is timed)
        dt = timeit(lambda: ops.topk(db, q, 12), iters=50 if small else 5, warm=5 if small else 2)
        print(f"topk N={N} Q={Q}: {dt*1e6:.1f} us  db stream {N*768*4/dt/1e9:.0f} GB/s  ({Q/dt:.0f} queries/s)")
        res[f"N{N}_Q{Q}"] = {"us": round(dt * 1e6, 1), "db_stream_GBps": round(N * 768 * 4 / dt / 1e9), "queries_per_s": round(Q / dt)}
        if Q >= 16:     # batches run the fan-out kernel (fp32 MFMA, one pass over the table per 256 queries); the 16-chain scan kernel beside it
            res[f"N{N}_Q{Q}"]["fp32_mfma_tflops"] = round(2.0 * N * Q * 768 / dt / 1e12, 1)
            dtc = timeit(lambda: ops.topk(db, q, 12, order="chain16"), iters=3)
            print(f"   scan kernel (order chain16): {dtc*1e6:.1f} us; fan-out kernel {2.0*N*Q*768/dt/1e12:.1f} TFLOP/s of fp32 MFMA (peak 157)")
            res[f"N{N}_Q{Q}"]["chain16_us"] = round(dtc * 1e6, 1)
            with ops.dispatched() as dsp:
                ops.topk(db, q, 12)
            res[f"N{N}_Q{Q}"]["launches"] = "+".join(f"{k}:{v}" for k, v in sorted(dsp.counts.items()))
            planb = ops.TopkPlan(db, Q, 12)                                      # the batch through a prepared plan: nothing allocated per call, host cost = one C-ABI call
            planb.queries.copy_(q)
            dtp = timeit(planb.run, iters=50 if small else 5, warm=5 if small else 2)
            assert torch.equal(planb.rows, ops.topk(db, q, 12)[0])
            print(f"   launches {res[f'N{N}_Q{Q}']['launches']}; prepared plan: {dtp*1e6:.1f} us per call")
            res[f"N{N}_Q{Q}"]["plan_us"] = round(dtp * 1e6, 1)
            del planb
        if Q <= 4:      # the interactive query through a prepared plan: one C-ABI call = one kernel launch, nothing allocated
            plan = ops.TopkPlan(db, Q, 12, graph=True)
            plan.queries.copy_(q)
            dtp = timeit(plan.run, iters=200, warm=20)                       # back-to-back launches: max(host call, kernel)
            dtg = timeit(plan.replay, iters=200, warm=20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                plan.run()
                torch.cuda.synchronize()                                     # one query at a time: launch + kernel + completion
            lat = (time.perf_counter() - t0) / 200
            assert torch.equal(plan.rows, ops.topk(db, q, 12)[0])
            print(f"   prepared plan: {dtp*1e6:.1f} us per call ({N*768*4/dtp/1e9:.0f} GB/s), HIP-graph replay {dtg*1e6:.1f} us, launch-to-completion latency {lat*1e6:.1f} us")
            res[f"N{N}_Q{Q}"].update({"plan_us": round(dtp * 1e6, 1), "plan_graph_us": round(dtg * 1e6, 1), "plan_latency_us": round(lat * 1e6, 1)})
    return res


def norm():
    B, S, D = 2, 17776, 3072
    x = torch.randn(B, S, D, device=DEV).to(torch.bfloat16)
    w = torch.ones(D, device=DEV, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    dt = timeit(lambda: ops.layernorm(x, w, w, 1e-5, out=y))
    print(f"layernorm [{B*S},{D}]: {dt*1e6:.1f} us  {2*x.numel()*2/dt/1e9:.0f} GB/s")
    qkv = torch.randn(B, S, 3 * 48 * 64, device=DEV).to(torch.bfloat16)
    g = torch.ones(64, device=DEV, dtype=torch.bfloat16)
    cs = torch.randn(S - 226, 64, device=DEV)
    dt = timeit(lambda: ops.qknorm_rope_(qkv, 48, g, g, g, g, cs, cs, 226, q_premul=0.18))
    print(f"qknorm_rope: {dt*1e6:.1f} us  {2*(qkv.numel()*2//3)*2/dt/1e9:.0f} GB/s")




def unet(precision="bf16", net=None):
    """DynamiCrafter-1024 UNet + CAMA tokens, one CFG denoise step at 16x576x1024 (x [2, 8, 16, 72, 128]), random-init weights;
    precision='fp8': spatial self-attention on the e4m3 MFMA path (BASELINE config #5)"""
    from motionrag_amd import workloads as W, dynamicrafter as dc
    net = W.dynamicrafter1024_unet(DEV) if net is None else net
    dc.set_attention_precision(net, precision)
    x, ts, ctx, fs = W.dynamicrafter1024_inputs(DEV)
    dt = timeit(lambda: net(x, ts, context=ctx, fs=fs), iters=5, warm=2)
    dc.set_attention_precision(net, "bf16")
    T = W.DC1024_STEP_TFLOP
    print(f"DynamiCrafter-1024 UNet CFG step (16x576x1024, attention {precision}): {dt*1e3:.1f} ms  {T/dt:.0f} TFLOP/s of {T} TFLOP algorithmic  -> {16/dt:.1f} frames/s")
    return {"ms_per_cfg_step": round(dt * 1e3, 1), "algorithmic_tflop": T, "tflops_per_s": round(T / dt), "frames_per_s": round(16 / dt, 1)}


def count_flops(fn):
    from motionrag_amd import workloads as W
    return W.count_flops(fn)


def svd():
    """SVD img2vid UNet + motion adapters, one CFG denoise step at 14x576x1024 (sample [2, 14, 8, 72, 128]), random-init weights"""
    from motionrag_amd import workloads as W
    net, names = W.svd_unet(DEV)
    step, _, _ = W.svd_step(net, DEV)
    Fr = 14
    fl = count_flops(step)
    dt = timeit(step, iters=5, warm=2)
    print(f"SVD UNet CFG step (14x576x1024, {len(names)} adapter sites): {dt*1e3:.1f} ms  {fl/dt/1e12:.0f} TFLOP/s of {fl/1e12:.1f} TFLOP algorithmic  -> {Fr/dt:.1f} frames/s")
    return {"ms_per_cfg_step": round(dt * 1e3, 1), "algorithmic_tflop": round(fl / 1e12, 1), "tflops_per_s": round(fl / dt / 1e12), "frames_per_s": round(Fr / dt, 1)}


def encoders(clips=10, frames=49, H=480, W=720):
    """the RAG-side overhead in front of the denoising loop (README.md:48 of the reference: "+3.6 s"): VideoMAE-B over the 9 retrieved
    clips + the target + the all-zero CFG clip, DINOv2-L over their first frames, then CAMA -- raw bf16 videos [1, 9, 49, 3, 480, 720] in, motion tokens out"""
    from motionrag_amd import cama, encoders as P
    torch.manual_seed(0)
    vm, dm = P.VideoMAEEmbedder(P.VideoMAEModel()).to(DEV, torch.bfloat16), P.DINOImageEmbedder(P.Dinov2Model()).to(DEV, torch.bfloat16)
    model = cama.build_cama(vm, dm).to(DEV, torch.bfloat16)
    vids = (torch.rand(1, clips, frames, 3, H, W, device=DEV) * 2 - 1).to(torch.bfloat16)
    batch = {"ref_videos": vids[:, :clips - 1], "video": vids[:, clips - 1]}
    flat = vids.reshape(clips, frames, 3, H, W)
    idx = P.uniform_frame_indices(frames, 16).to(DEV, torch.int32)
    t_pix = timeit(lambda: P.pixels_to_patch_rows(flat, resize=224, crop=224, mode="bilinear", patch=(2, 16, 16), frame_idx=idx), iters=10, warm=2)
    t_vm = timeit(lambda: vm(flat), iters=5, warm=2)
    t_dm = timeit(lambda: dm(flat[:, 0]), iters=5, warm=2)
    t_all = timeit(lambda: model.predict(batch, do_classifier_free_guidance=True), iters=5, warm=2)
    t_graph = None
    try:                                    # the whole RAG-side path as ONE HIP graph replay (pixels -> motion tokens); input copy into the static buffers included
        gp = cama.GraphedPredict(model, do_classifier_free_guidance=True)
        gp(batch)
        t_graph = timeit(lambda: gp(batch), iters=5, warm=1)
    except Exception as e:                  # capture is an optimisation: report, do not fail the bench
        print("HIP-graph capture of predict-from-pixels failed:", repr(e)[:200])
    # algorithmic bytes of the pixel kernel: the 16 sampled frames of every clip read once + the patch rows written once
    px_bytes = clips * 16 * 3 * H * W * 2 + clips * 8 * 196 * 1536 * 2
    S, D, L = 1568, 768, 12
    vm_flops = clips * (L * (24 * S * D * D + 4 * S * S * D) + 2 * S * 1536 * D)
    print(f"pixels -> patch rows ({clips} clips x 16 of {frames} frames, {H}x{W} -> 224): {t_pix*1e6:.0f} us  {px_bytes/t_pix/1e9:.0f} GB/s algorithmic")
    print(f"VideoMAE-B embedder, {clips} clips: {t_vm*1e3:.2f} ms  {vm_flops/t_vm/1e12:.0f} TFLOP/s;  DINOv2-L embedder, {clips} images: {t_dm*1e3:.2f} ms")
    print(f"CAMA predict from raw pixels (both encoders + 2 Resamplers + encoder, CFG): {t_all*1e3:.2f} ms per clip" + (f"; as one HIP graph {t_graph*1e3:.2f} ms" if t_graph else ""))
    return {"pixels_to_patch_rows_us": round(t_pix * 1e6), "pixels_GBps_algorithmic": round(px_bytes / t_pix / 1e9), "videomae_b_ms": round(t_vm * 1e3, 2),
            "videomae_b_tflops_per_s": round(vm_flops / t_vm / 1e12), "dinov2_l_ms": round(t_dm * 1e3, 2), "cama_predict_from_pixels_ms": round(t_all * 1e3, 2), "cama_predict_from_pixels_hip_graph_ms": round(t_graph * 1e3, 2) if t_graph else None,
            "clips": clips, "source": f"{frames}x{H}x{W} bf16"}


def vae(frames=16, h=72, w=128):
    """DynamiCrafter KL-VAE decode of one clip's final latents: z [1, 4, 16, 72, 128] -> video [1, 3, 16, 576, 1024], the shipped decoder
    (configs/dynamicrafter/MotionRAG_open.yml:245-259), random-init weights, all frames in one batch"""
    from motionrag_amd import dynamicrafter_vae as V
    torch.manual_seed(0)
    m = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[],
                             dropout=0.0), embed_dim=4).to(DEV, torch.bfloat16)
    z = (torch.randn(1, 4, frames, h, w, device=DEV) * 0.18215).to(torch.bfloat16)
    out = {}
    fl = count_flops(lambda: out.setdefault("y", V.decode_first_stage(m, z)))
    assert out["y"].shape == (1, 3, frames, 8 * h, 8 * w) and torch.isfinite(out["y"].float()).all()
    dt = timeit(lambda: V.decode_first_stage(m, z), iters=3, warm=1)
    print(f"KL-VAE decode {frames}x{8*h}x{8*w}: {dt*1e3:.1f} ms  {fl/dt/1e12:.0f} TFLOP/s of {fl/1e12:.1f} TFLOP (GEMM / conv launches)  -> {frames/dt:.0f} frames/s")
    return {"ms_per_clip": round(dt * 1e3, 1), "algorithmic_tflop": round(fl / 1e12, 1), "tflops_per_s": round(fl / dt / 1e12), "frames_per_s": round(frames / dt)}


def t5():
    """CogVideoX's prompt encoder: the T5-v1.1-XXL encoder (24 layers x 4096, 64 heads, d_ff 10240; 4.76 B parameters = 9.5 GB bf16) on prompt +
    negative prompt, 226 tokens each -- weight-bandwidth-bound: every weight is read once per call"""
    from motionrag_amd import t5 as T
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(DEV):
            m = T.T5EncoderModel()
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 4096 ** -0.5 if "SelfAttention.q" not in n else (4096 * 64) ** -0.5)
    ids = torch.randint(0, 32128, (2, 226), device=DEV)
    out = m(ids)[0]
    assert out.shape == (2, 226, 4096) and torch.isfinite(out.float()).all()
    dt = timeit(lambda: m(ids), iters=5, warm=2)
    nbytes = sum(p.numel() for n, p in m.named_parameters() if not n.startswith("shared")) * 2
    print(f"T5-v1.1-XXL encoder, 2 x 226 tokens: {dt*1e3:.2f} ms  weights {nbytes/1e9:.2f} GB -> {nbytes/dt/1e12:.2f} TB/s")
    return {"ms": round(dt * 1e3, 2), "weight_GB": round(nbytes / 1e9, 2), "weight_stream_TBps": round(nbytes / dt / 1e12, 2)}


def svd_vae(frames=14, h=72, w=128):
    """SVD's temporal-decoder VAE: decode one clip's latents [14, 4, 72, 128] -> [14, 3, 576, 1024] in one chunk (decode_chunk_size = num_frames), and
    encode the conditioning frame; the shipped configuration (97.7 M parameters), random-init weights"""
    from motionrag_amd import svd_vae as V
    torch.manual_seed(0)
    m = V.AutoencoderKLTemporalDecoder().to(DEV, torch.bfloat16)
    z = torch.randn(frames, 4, h, w, device=DEV).to(torch.bfloat16)
    out = {}
    fl = count_flops(lambda: out.setdefault("y", m.decode(z, num_frames=frames).sample))
    assert out["y"].shape == (frames, 3, 8 * h, 8 * w) and torch.isfinite(out["y"].float()).all()
    dt = timeit(lambda: m.decode(z, num_frames=frames), iters=3, warm=1)
    img = (torch.rand(1, 3, 8 * h, 8 * w, device=DEV) * 2 - 1).to(torch.bfloat16)
    de = timeit(lambda: m.encode(img), iters=5, warm=1)
    print(f"SVD temporal-decoder VAE: decode {frames}x{8*h}x{8*w} {dt*1e3:.1f} ms  {fl/dt/1e12:.0f} TFLOP/s of {fl/1e12:.1f} TFLOP -> {frames/dt:.0f} frames/s; encode one {8*h}x{8*w} frame {de*1e3:.2f} ms")
    return {"decode_ms_per_clip": round(dt * 1e3, 1), "algorithmic_tflop": round(fl / 1e12, 1), "tflops_per_s": round(fl / dt / 1e12), "frames_per_s": round(frames / dt),
            "encode_frame_ms": round(de * 1e3, 2)}


def text_embedder():
    """the retrieval row's text embedder: gte-base-en-v1.5's encoder (12 x 768, 136.8 M parameters, random-init) -- one 16-token query (the latency in
    front of every `text_search(text)`) and the builder's throughput on 32-token captions in unpadded batches of 1024"""
    from motionrag_amd.text_embedder import NewModel
    torch.manual_seed(0)
    m = NewModel().to(DEV, torch.bfloat16)
    q = torch.randint(1, 30528, (1, 16), device=DEV)
    caps = torch.randint(1, 30528, (1024, 32), device=DEV)
    assert torch.isfinite(m(q)[0].float()).all()
    tq = timeit(lambda: m(q), iters=20, warm=3)
    tb = timeit(lambda: m(caps), iters=5, warm=2)
    print(f"gte-base text embedder: one 16-token query {tq*1e3:.2f} ms; 1024 captions x 32 tokens {tb*1e3:.1f} ms -> {1024/tb:.0f} captions/s")
    return {"query_16_tokens_ms": round(tq * 1e3, 2), "captions_per_s_32_tokens": round(1024 / tb)}


def cogvideox_vae(frames=13, h=60, w=90, tiling=True):
    """CogVideoX-5B's 3-D causal VAE (215.6 M parameters, random-init): decode one clip's final latents [1, 16, 13, 60, 90] -> 49 frames of 480 x 720
    as the reference configures it (cogvideox/module.py:39-40: tiling + slicing -> nine 30 x 45 latent tiles of six frame batches each), the untiled decode
    beside it, and the tiled encode of the conditioning image"""
    from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX
    torch.manual_seed(0)
    m = AutoencoderKLCogVideoX().to(DEV, torch.bfloat16)
    z = torch.randn(1, 16, frames, h, w, device=DEV).to(torch.bfloat16)
    res = {}
    for name, tiled in (("tiled", True), ("tiled_one_stream", True), ("untiled", False)):
        if tiled and not tiling:
            continue
        m.enable_tiling() if tiled else m.disable_tiling()
        m.tile_streams = 1 if name == "tiled_one_stream" else int(os.environ.get("MRAG_VAE_STREAMS", "3"))
        out = {}
        fl = count_flops(lambda: out.setdefault("y", m.decode(z).sample))
        assert out["y"].shape == (1, 3, 1 + 4 * (frames - 1), 8 * h, 8 * w) and torch.isfinite(out["y"].float()).all()
        del out
        dt = timeit(lambda: m.decode(z), iters=2, warm=1)
        print(f"CogVideoX VAE decode ({name}) {frames} latent frames -> {1 + 4 * (frames - 1)}x{8*h}x{8*w}: {dt*1e3:.0f} ms  {fl/dt/1e12:.0f} TFLOP/s of {fl/1e12:.0f} TFLOP")
        res[f"decode_{name}_ms_per_clip"] = round(dt * 1e3)
        res[f"decode_{name}_tflop"] = round(fl / 1e12, 1)
        res[f"decode_{name}_tflops_per_s"] = round(fl / dt / 1e12)
    m.enable_tiling()
    m.tile_streams = 3
    img = (torch.rand(1, 3, 1, 8 * h, 8 * w, device=DEV) * 2 - 1).to(torch.bfloat16)
    de = timeit(lambda: m.encode(img), iters=3, warm=1)
    print(f"CogVideoX VAE encode of the conditioning image (tiled): {de*1e3:.1f} ms")
    res["encode_image_ms"] = round(de * 1e3, 1)
    return res


def r6():
    """round 6's items at the BASELINE shapes, each against the form it replaces (same process, back to back)"""
    import ctypes
    from motionrag_amd import _lib
    B, S, H, D = 2, 17776, 48, 3072
    # (1) AdaLN LayerNorm [35 552, 3 072]: layernorm_stream_kernel (unless the library was built with -DMRAG_LN_NO_STREAM: tools/build_variant.sh)
    x = torch.randn(B, S, D, device=DEV).to(torch.bfloat16)
    w = torch.ones(D, device=DEV, dtype=torch.bfloat16)
    md = (0.3 * torch.randn(B, 4, D, device=DEV)).to(torch.bfloat16)
    y = torch.empty_like(x)
    with ops.dispatched() as d:
        dt = timeit(lambda: ops.layernorm(x, w, w, 1e-5, out=y, shift0=md[:, 0], scale0=md[:, 1], shift1=md[:, 2], scale1=md[:, 3], rows_per_batch=S, split=226, mod_stride=md.stride(0)), iters=50)
    print(f"r6 layernorm + AdaLN [{B*S},{D}] ({'/'.join(d.counts)}): {dt*1e6:.1f} us  {2*x.numel()*2/dt/1e12:.2f} TB/s")
    # (2) the folded motion branch: scores as one launch with per-sample weights over packed 26-column blocks, vs round 5's two launches over 32-column blocks
    o = torch.randn(B, S, H * 64, device=DEV).to(torch.bfloat16)
    v = torch.randn(B, 25, H * 64, device=DEV).to(torch.bfloat16)
    M32 = (torch.randn(B, H * 32, D, device=DEV) * 0.02).to(torch.bfloat16)
    M26 = (torch.randn(B, 1280, D, device=DEV) * 0.02).to(torch.bfloat16)
    sc32 = torch.empty(B, S, H * 32, device=DEV, dtype=torch.bfloat16)

    def gemms32():
        for b in range(B):
            ops.linear(o[b], M32[b], out=sc32[b])
    dt32 = timeit(gemms32, iters=30)
    sc26 = torch.empty(B, S, 1280, device=DEV, dtype=torch.bfloat16)
    with ops.dispatched() as d:
        dt26 = timeit(lambda: ops.linear_per_sample(o, M26, out=sc26), iters=30)
    print(f"r6 score GEMM: two launches x [17776 x 1536 x 3072] {dt32*1e6:.1f} us -> one launch, per-sample weights [35552 x 1280 x 3072] ({'/'.join(d.counts)}) {dt26*1e6:.1f} us")
    f32 = timeit(lambda: ops.ip_attn_folded_(sc32, v, o, H, 25), iters=50)
    f26 = timeit(lambda: ops.ip_attn_folded_(sc26, v, o, H, 25, key_stride=26), iters=50)
    print(f"r6 ip_attn_folded: 32-column blocks {f32*1e6:.1f} us ({(sc32.numel()+2*o.numel())*2/f32/1e12:.2f} TB/s) -> 26-column blocks {f26*1e6:.1f} us ({(sc26.numel()+2*o.numel())*2/f26/1e12:.2f} TB/s)")
    del sc32, sc26, M32, M26
    # (3) FF1 + GELU [35 552 x 12 288 x 3 072]: 26 rounds + a 16-tile tail as a rectangle of 128x128 tiles vs a 27th round
    wf = (torch.randn(4 * D, D, device=DEV) * 0.02).to(torch.bfloat16)
    bfb = torch.zeros(4 * D, device=DEV, dtype=torch.bfloat16)
    hid = torch.empty(B * S, 4 * D, device=DEV, dtype=torch.bfloat16)
    x2 = x.view(B * S, D)
    for name, tune in (("tail rectangle", 1 << 19), ("27th round", 0), ("tail rectangle", 1 << 19), ("27th round", 0)):
        ops.TUNING["gemm"] = tune
        with ops.dispatched() as d:
            dt = timeit(lambda: ops.linear(x2, wf, bfb, out=hid, epilogue=ops.EPI_GELU_TANH), iters=20)
        print(f"r6 FF1 + GELU, {name} ({'/'.join(d.counts)}): {dt*1e3:.3f} ms  {2.0*B*S*D*4*D/dt/1e12:.0f} TFLOP/s")
    ops.TUNING["gemm"] = 0
    del wf, hid
    # (4) the stream-copy ceiling beside torch's copy
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV).random_(0, 255)
    dst = torch.empty_like(src)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dtc = timeit(lambda: L.mrag_probe_stream_copy(st, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(), 0), iters=20)
    dtt = timeit(lambda: dst.copy_(src), iters=20)
    print(f"r6 1 GiB copy: library stream copy {2*src.numel()/dtc/1e12:.2f} TB/s, torch copy_ {2*src.numel()/dtt/1e12:.2f} TB/s")
    del src, dst
    # (5) how much a partial last round of the attention costs: query rows chosen so that the launch is 11.0 / 11.5 / 12.0 rounds of 768 resident workgroups
    qkv = torch.randn(B, S, 3, H, 64, device=DEV).to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device=DEV, dtype=torch.bfloat16)
    for tiles in (88, 92, 96, 88, 92, 96):
        sq = tiles * 192
        qv = qkv[:, :sq, 0] if sq <= S else torch.randn(B, sq, H, 64, device=DEV).to(torch.bfloat16)
        ov = out[:, :sq] if sq <= S else torch.empty(B, sq, H * 64, device=DEV, dtype=torch.bfloat16)
        dt = timeit(lambda: ops.attention(qv, qkv[:, :, 1], qkv[:, :, 2], out=ov), iters=15)
        print(f"r6 attention, {tiles} query tiles x 96 (b, h) = {tiles * 96 / 768:.2f} rounds of 768 workgroups: {dt*1e3:.3f} ms = {dt*1e3/(tiles*96/768):.4f} ms per round-equivalent")


def copy_probe():
    """sweep of the stream-copy probe's forms (csrc/probe.hip) against torch's copy: the form that wins is the library's HBM ceiling (roofline.ceilings.stream_copy_TBps)"""
    import ctypes
    from motionrag_amd import _lib
    L = _lib.lib()
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV).random_(0, 255)
    dst = torch.empty_like(src)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = {0: "unroll 4, contiguous runs, nontemporal", 1: "unroll 4, contiguous runs", 2: "unroll 8, contiguous, nontemporal", 3: "unroll 4, grid-strided (first form)",
             4: "unroll 2, contiguous, nontemporal", 5: "unroll 1, nontemporal"}
    for rep in range(2):
        for form in range(6):
            for per_cu in (4, 8, 16, 32):
                v = form | (per_cu << 4)
                dt = timeit(lambda: L.mrag_probe_stream_copy(st, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(), v), iters=10)
                print(f"copy_probe form {form} ({names[form]}), {per_cu} workgroups per CU: {2*src.numel()/dt/1e12:.2f} TB/s")
        dtt = timeit(lambda: dst.copy_(src), iters=10)
        print(f"copy_probe torch copy_: {2*src.numel()/dtt/1e12:.2f} TB/s")


def topk_small():
    """the fan-out kernel at BASELINE config #1's table size (10 000 rows) x 256 queries, for a kernel trace (tools/prof.sh topk_small)"""
    db = torch.randn(10000, 768, device=DEV); q = torch.randn(256, 768, device=DEV)
    for order in ("mfma", "mfma_stream", "chain16"):
        dt = timeit(lambda: ops.topk(db, q, 12, order=order), iters=50, warm=5)
        print(f"topk 10000 x 256 order {order}: {dt*1e6:.1f} us")


def topk_sizes():
    """the fan-out form's two launch shapes over table sizes and batch widths (one launch: tables of one resident round; streaming: three launches)"""
    for N, Q in ((1000, 256), (4000, 256), (10000, 256), (10000, 64), (10000, 16), (20000, 256), (40000, 128), (65536, 64), (1000000, 256)):
        db = torch.randn(N, 768, device=DEV); q = torch.randn(Q, 768, device=DEV)
        line = f"topk N={N} Q={Q}:"
        for order in ("mfma", "mfma_stream"):
            with ops.dispatched() as d:
                ops.topk(db, q, 12, order=order)
            dt = timeit(lambda: ops.topk(db, q, 12, order=order), iters=30 if N < 100000 else 5, warm=5)
            line += f"  {order} {dt*1e6:.1f} us ({'+'.join(sorted(d.counts))})"
        print(line)
        del db, q


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "gemm", "topk", "norm"]
    for w in which:
        globals()[w]()
