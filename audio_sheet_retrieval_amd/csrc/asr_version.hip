// asr_version(): library version + the hash of the sources this binary was built from (build.py passes
// -DASR_SOURCE_HASH; _lib.load_library compares it with the sources next to the .so, so a stale build is detected).
#include "../../include/asr_hip.h"

#ifndef ASR_SOURCE_HASH
#define ASR_SOURCE_HASH "unknown"
#endif

extern "C" const char *asr_version(void) { return "asr_hip 0.2 (gfx950) src:" ASR_SOURCE_HASH; }
