mkdir -p gpurun_out
for n in 0 3 6; do timeout 300 python tools/vae_queue_probe.py $n 2>&1 | tail -1; done > gpurun_out/r3h_vae_queue_probe.txt
for n in 0 6; do GPU_MAX_HW_QUEUES=8 timeout 300 python tools/vae_queue_probe.py $n 2>&1 | tail -1; done >> gpurun_out/r3h_vae_queue_probe.txt
cat gpurun_out/r3h_vae_queue_probe.txt
