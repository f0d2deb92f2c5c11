// Error reporting and version for libnvq.
#include "common.h"

namespace nvq {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nvq

extern "C" {
int nvq_version(void) { return 100; }
const char* nvq_last_error(void) { return nvq::g_err; }
}
