#include <string>
#include "../../include/mm2amd.h"
namespace mm2amd {
static thread_local std::string g_last_error;
void capi_set_error(const std::string &msg) { g_last_error = msg; }
int capi_fail(int code, const std::string &msg) { g_last_error = msg; return code; }
}
extern "C" {
const char *mm2amd_last_error(void) { return mm2amd::g_last_error.c_str(); }
int mm2amd_version(void) { return 1; }
}
