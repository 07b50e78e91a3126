// Version and error reporting of the C ABI.
#include "common.h"

#include <cstring>

namespace lbm {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace lbm

extern "C" int lbm_version(void) { return LBM_ABI_VERSION; }
extern "C" const char* lbm_last_error(void) { return lbm::last_error_buf(); }
