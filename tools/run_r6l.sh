# round 6, GPU call L: smoke() and the full GPU suite on the final commit (host-side changes since call G: attn_processor.packed_score_layout, named tuning constants, tests)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6l
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6l/smoke.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r6l/tests.log
python -c "from motionrag_amd._lib import source_hash, lib; print('source hash', source_hash(), lib().mrag_source_hash().decode())" | tee gpurun_out/r6l/hash.log
