#include "common.h"
extern "C" int as_abi_version(void) { return 7; }
