// adi_stamp.hip -- identity of the build: the first 16 hex digits of the SHA-256 over the library's sources and compile
// flags, computed by adi_thermal_fields_amd/build.py and compiled in here (-DADI_SOURCE_STAMP).  It does not depend on
// where the tree was checked out (the objects are built with -ffile-prefix-map, so __FILE__ in the error strings is
// relative too): profiles/pmc_traffic.json carries the stamp of the library the counter passes ran, and bench.py quotes
// those bytes only for a library that reports the same stamp.
#include "../../include/adi_hip.h"

#ifndef ADI_SOURCE_STAMP
#define ADI_SOURCE_STAMP "unstamped"
#endif

extern "C" const char *adi_build_stamp(void) { return ADI_SOURCE_STAMP; }
