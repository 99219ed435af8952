# round 6, GPU call Z: what makes the first load after the grid wait slow: phase stamps with the dense score stores as shipped / left out / as plain cached stores
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6z
for v in topk_stats topk_stats_nod1 topk_stats_nod2; do
  echo "== $v"; MRAG_HIP_LIB=$PWD/tools/lib_$v.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6z/topk_diag_store_forms.txt 2>&1
cat gpurun_out/r6z/topk_diag_store_forms.txt
