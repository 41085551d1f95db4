// api.hip -- library identification.
#include "common.h"
extern "C" const char* rfx_version(void) { return "rfx 0.1.0 gfx950"; }
