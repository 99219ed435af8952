import sys, time, torch, gc
sys.path.insert(0, '.')
from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX
torch.manual_seed(0)
m = AutoencoderKLCogVideoX().to("cuda", torch.bfloat16); m.enable_tiling()
z = torch.randn(1, 16, 13, 60, 90, device="cuda").to(torch.bfloat16)
for s in (1, 3):
    m.tile_streams = s
    m.decode(z); torch.cuda.synchronize()
    for gcoff in (False, True):
        if gcoff: gc.disable()
        t0 = time.perf_counter(); y = m.decode(z); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"streams {s} gc_off {gcoff}: host issue {1e3*(t1-t0):.0f} ms, total {1e3*(t2-t0):.0f} ms")
        gc.enable()
