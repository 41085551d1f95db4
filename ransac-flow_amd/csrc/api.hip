// api.hip -- library identification.
#include "common.h"
extern "C" const char* rfx_version(void) { return "rfx 0.3.0 gfx950"; }
extern "C" int rfx_abi_version(void) { return RFX_ABI_VERSION; }
