# round-5 final evidence: smoke(), the full GPU suite, the judged command under rocprofv3 (tools/run_final.sh) and unprofiled on the same box
mkdir -p gpurun_out/r5_final
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r5_final/smoke.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -9 | tee gpurun_out/r5_final/tests.log
bash tools/run_final.sh r5_final > gpurun_out/r5_final/run_final.log 2>&1
timeout 900 python bench.py > gpurun_out/r5_final/bench_unprofiled.json 2> gpurun_out/r5_final/bench_unprofiled.err
python - <<EOF2
import json
for f in ("bench.json", "bench_unprofiled.json"):
    b=json.load(open("gpurun_out/r5_final/"+f))
    sw=b["secondary_workloads"]
    print(f, b["ms_per_step"], b["roofline"]["frac"], b["roofline"].get("frac_of_sustained"), b["roofline"]["avg_launch_ms"], b["cama_ms"], b["cama_hip_graph_ms"],
          sw["svd_unet_14x576x1024_cfg_step"]["ms_per_cfg_step"], sw["dynamicrafter1024_unet_16x576x1024_cfg_step"]["ms_per_cfg_step"],
          sw["retrieval_top12_768d"]["N10000_Q256"]["us"], sw["retrieval_top12_768d"]["N1000000_Q256"]["us"], sw["retrieval_text_embedder_gte_base"], sw["t5_xxl_prompt_encoder_2x226"]["ms"])
EOF2
