mkdir -p gpurun_out/r5_t; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r5_t/tests.log; timeout 900 python bench.py > gpurun_out/r5_t/bench_unprofiled.json 2> gpurun_out/r5_t/bench_unprofiled.err; python - <<EOF2
import json
b=json.load(open("gpurun_out/r5_t/bench_unprofiled.json"))
print(b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], b["cama_hip_graph_ms"])
sw=b["secondary_workloads"]
print(sw["svd_unet_14x576x1024_cfg_step"], sw["dynamicrafter1024_unet_16x576x1024_cfg_step"], sw["dynamicrafter1024_unet_16x576x1024_cfg_step_fp8_attention"])
print(sw["retrieval_top12_768d"])
EOF2
