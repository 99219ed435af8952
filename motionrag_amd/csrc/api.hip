// api.hip -- library identity entry points of libmrag_hip.so.
#include "../../include/mrag_hip.h"
extern "C" int mrag_abi_version(void) { return 8; }
extern "C" const char* mrag_target_arch(void) { return "gfx950"; }
