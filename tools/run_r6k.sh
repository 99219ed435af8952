# round 6, GPU call K: the tests touched by the last host-side change, rocprofv3 kernel statistics of the two UNet CFG steps, the three VAE decodes and CAMA on the final library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6k
python -m pytest tests/test_gpu_fullwidth_golden.py tests/test_gpu_attn_processor_golden.py tests/test_gpu_round6.py tests/test_gpu_multiproc.py -q > gpurun_out/r6k/pytest_sel.txt 2>&1; echo "rc=$?" >> gpurun_out/r6k/pytest_sel.txt
tail -4 gpurun_out/r6k/pytest_sel.txt | cut -c1-200
bash tools/prof.sh svd r6_svd_unet_step > gpurun_out/r6k/prof_svd.log 2>&1
bash tools/prof.sh unet r6_dc_unet_step > gpurun_out/r6k/prof_dc.log 2>&1
bash tools/prof.sh vae r6_dc_vae_decode > gpurun_out/r6k/prof_vae.log 2>&1
bash tools/prof.sh svd_vae r6_svd_vae_decode > gpurun_out/r6k/prof_svd_vae.log 2>&1
bash tools/prof_cama.sh r6_cama > gpurun_out/r6k/prof_cama.log 2>&1
tail -3 gpurun_out/r6k/prof_svd.log gpurun_out/r6k/prof_dc.log gpurun_out/r6k/prof_vae.log gpurun_out/r6k/prof_svd_vae.log gpurun_out/r6k/prof_cama.log | cut -c1-250
timeout 300 python tools/cama_prof.py 2>&1 | grep "CAMA predict" > gpurun_out/r6k/cama_prof.txt
cat gpurun_out/r6k/cama_prof.txt | cut -c1-300
