// Library-wide C ABI entry points (version, error reporting).
#include "hl_common.h"

extern "C" {
int hl_version(void) { return 100; }  // 0.1.0
const char *hl_last_error(void) { return hl::err_buf(); }
}
