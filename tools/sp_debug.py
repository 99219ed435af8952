#!/usr/bin/env python3
"""developer check: sequence-sharded DiT forward (two 'ranks' as threads on one GPU) against the unsharded forward at full width"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionrag_amd import ops
from motionrag_amd.cogvideox import CogVideoXTransformer3DModel, get_3d_rotary_pos_embed
from motionrag_amd.dist import SequenceParallel
from motionrag_amd.workloads import random_init_
DEV = "cuda"
torch.manual_seed(0)
old = torch.get_default_dtype(); torch.set_default_dtype(torch.bfloat16)
with torch.device(DEV):
    dit = CogVideoXTransformer3DModel(num_layers=int(os.environ.get("LAYERS", "1")), sample_frames=3)
    dit.install_motion_adapters(1024)
torch.set_default_dtype(old)
random_init_(dit); dit.patch_embed.pos_embedding.normal_(0, 0.02); dit.eval()
g = torch.Generator().manual_seed(1)
lat, img = (torch.randn(1, 3, 16, 60, 90, generator=g).to(DEV, torch.bfloat16) for _ in range(2))
text = torch.randn(2, 226, 4096, generator=g).to(DEV, torch.bfloat16)
ip = torch.randn(2, 25, 1024, generator=g).to(DEV, torch.bfloat16)
t = torch.tensor([481.0, 481.0], device=DEV)
cos, sin = (x.to(DEV) for x in get_3d_rotary_pos_embed(64, 3, 30, 45))
with torch.no_grad():
    want = dit(lat, text, t, image_rotary_emb=((cos, sin), ip), image_latents=img, batch=2)
    again = dit(lat, text, t, image_rotary_emb=((cos, sin), ip), image_latents=img, batch=2)
print("determinism:", torch.equal(want, again))
for nofuse in (False, True):
    ops.TUNING["no_qkv_fuse"] = nofuse
    world = 2
    slots, bar, outs, errs = [None] * world, threading.Barrier(world), [None] * world, []
    def gather_for(rank):
        def ag(x):
            torch.cuda.current_stream().synchronize(); slots[rank] = x.contiguous(); bar.wait()
            out = torch.cat(list(slots), dim=0); torch.cuda.current_stream().synchronize(); bar.wait(); return out
        return ag
    def run(rank):
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                outs[rank] = dit(lat, text, t, image_rotary_emb=((cos, sin), ip), image_latents=img, batch=2, sp=SequenceParallel(rank, world, all_gather=gather_for(rank)))
                torch.cuda.current_stream().synchronize()
        except Exception as e:
            errs.append(e); bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join(timeout=300) for x in th]
    print("no_qkv_fuse", nofuse, "errors", errs)
    for r in range(world):
        if outs[r] is not None:
            print(" rank", r, "rel err vs unsharded", ((outs[r].float() - want.float()).norm() / want.float().norm()).item())
ops.TUNING["no_qkv_fuse"] = False

# ---- where do the sharded and the unsharded forward part ways?  record block 0's attention input / output per run
import motionrag_amd.cogvideox as cvx
orig = cvx.joint_attention_core
rec = {}
def spy(attn, proc, x, text_len, rope, ip_, scale, sp=None):
    o = orig(attn, proc, x, text_len, rope, ip_, scale, sp=sp)
    key = ("sp", sp.rank) if sp is not None else ("full",)
    if key not in rec:
        rec[key] = (x.clone(), o.clone())
    return o
cvx.joint_attention_core = spy
with torch.no_grad():
    dit(lat, text, t, image_rotary_emb=((cos, sin), ip), image_latents=img, batch=2)
world = 2
slots, bar, outs, errs = [None] * world, threading.Barrier(world), [None] * world, []
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[x.start() for x in th]; [x.join(timeout=300) for x in th]
fx, fo = rec[("full",)]
S = fx.shape[1]
for r in range(world):
    sx, so = rec[("sp", r)]
    r0, r1 = r * S // world, (r + 1) * S // world
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    print(f"rank {r}: attention INPUT rel diff {rel(sx, fx[:, r0:r1]):.2e}; attention OUTPUT (incl. adapter) rel diff {rel(so, fo[:, r0:r1]):.2e}")
    d = (so.float() - fo[:, r0:r1].float()).norm(dim=-1) / fo[:, r0:r1].float().norm(dim=-1)      # per row
    worst = d[0].topk(5)
    print("   worst rows (local):", worst.indices.tolist(), [f"{v:.3f}" for v in worst.values.tolist()], " median row diff", d.median().item())
