# round 6, GPU call F: tests around the GroupNorm / CAMA-test / packed-read changes, then UNet / VAE A/B of the streaming GroupNorm passes (gnold = -DMRAG_GN_OLD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6f
python -m pytest tests/test_gpu_round6.py tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_vae.py tests/test_gpu_svd.py tests/test_gpu_svd_vae.py tests/test_gpu_cogvideox_vae.py tests/test_gpu_fullwidth_golden.py -q > gpurun_out/r6f/pytest_sel.txt 2>&1; echo "rc=$?" >> gpurun_out/r6f/pytest_sel.txt
tail -12 gpurun_out/r6f/pytest_sel.txt | cut -c1-220
for r in 1 2; do for v in shipped gnold; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 900 python tools/microbench.py svd unet vae svd_vae 2>&1 | grep -E "CFG step|VAE" | sed "s/^/$v: /" | cut -c1-170
done; done > gpurun_out/r6f/gn_stream_ab.txt 2>&1
cat gpurun_out/r6f/gn_stream_ab.txt
