// api.hip -- library identity entry points of libmrag_hip.so and the diagnostic launch counters.
#include "../../include/mrag_hip.h"
#include "common.h"

#ifndef MRAG_SOURCE_HASH
#define MRAG_SOURCE_HASH "unstamped"
#endif

extern "C" int mrag_abi_version(void) { return 10; }
extern "C" const char* mrag_target_arch(void) { return "gfx950"; }
// "MRAG_SOURCE_HASH=<hex>" is also findable in the file's bytes, so the build can read a binary's stamp without loading it
static const char k_source_stamp[] = "MRAG_SOURCE_HASH=" MRAG_SOURCE_HASH;
extern "C" const char* mrag_source_hash(void) { return k_source_stamp + 17; }

// one slot per enum mrag_kernel_id; bumped by MRAG_COUNT at every launch site (common.h), read only by mrag_dispatch_counts
unsigned long long mrag_dispatch_table[MRAG_K_COUNT];

extern "C" int mrag_dispatch_counts(uint64_t* out_host, int32_t n) {
  if (out_host)
    for (int i = 0; i < n && i < (int)MRAG_K_COUNT; ++i) out_host[i] = __atomic_load_n(&mrag_dispatch_table[i], __ATOMIC_RELAXED);
  return (int)MRAG_K_COUNT;
}

extern "C" const char* mrag_dispatch_name(int32_t id) {
  static const char* const names[MRAG_K_COUNT] = {
      "GEMM_W4", "GEMM_W4_QKNORM_ROPE", "GEMM_W4_GEGLU", "GEMM_256x256", "GEMM_256x320", "GEMM_256x128", "GEMM_128x128", "GEMM_STREAMK_TAIL", "GEMM_N320K320", "GEMM_192x256",
      "CONV3_W4", "CONV3_256x256", "CONV3_256x320", "CONV3_256x128", "CONV3_128x128", "CONV3_192x256",
      "CONVT_W4", "CONVT_256x256", "CONVT_256x320", "CONVT_128x128", "CONVT_192x256", "CONVT_256x128",
      "ATTN16", "ATTN16_KSPLIT", "ATTN_FLASH", "ATTN_FLASH_KSPLIT", "ATTN_COMBINE", "ATTN_TINY", "ATTN_SMALL", "ATTN_FP8", "IP_ATTN_FOLDED",
      "LAYERNORM", "LAYERNORM_ROWS", "QKNORM_ROPE", "GN_STATS", "GN_FOLD", "GN_APPLY", "GN_APPLY_MOD", "LAYERNORM_STREAM", "GN_STATS_FOLD",
      "TOPK_SCAN", "TOPK_SCAN_FUSED_MERGE", "TOPK_MERGE", "TOPK_MFMA", "GEMM_W4_TAIL_RECT", "GEMM_W4_BATCHED_W", "GEMM_SKINNY_LNA", "TOPK_DENSE", "TOPK_DENSE_FINISH", "GEMM_SKINNY"};
  static_assert(sizeof(names) / sizeof(names[0]) == MRAG_K_COUNT, "one name per enum mrag_kernel_id");
  return (id >= 0 && id < (int)MRAG_K_COUNT) ? names[id] : nullptr;
}
