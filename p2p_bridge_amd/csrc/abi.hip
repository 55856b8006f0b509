// abi.hip -- library identification for the C ABI (include/p2pb_hip.h)
#include "common.h"

extern "C" int p2pb_version(void) { return 1; }
extern "C" const char *p2pb_target_arch(void) { return "gfx950"; }
