# developer probe: which Tensile kernel does hipBLASLt pick for the DiT GEMM shapes?  (rocprofv3 --kernel-trace --stats -- python3 tools/exp/hipblaslt_name.py)
import torch
M = 2 * 17776
for N, K in ((9216, 3072), (3072, 3072), (12288, 3072), (3072, 12288)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
