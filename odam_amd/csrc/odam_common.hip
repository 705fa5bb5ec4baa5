// odam_common.hip -- ABI-level helpers shared by all entry points.
#include "odam_err.h"
extern "C" const char* odam_last_error(void) { return g_odam_err; }
