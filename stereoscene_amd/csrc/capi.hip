// Library identification entry points of libssbev_hip.so.
#include "common.h"

extern "C" {
int ssbev_version(void) { return 100; /* 0.1.0 */ }
const char* ssbev_build_arch(void) { return "gfx950"; }
}
