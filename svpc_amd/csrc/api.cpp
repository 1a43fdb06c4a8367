// Error-string plumbing of the C-ABI (include/svpc_hip.h: svpc_last_error).
#include <string.h>
#include <stdio.h>
static thread_local char g_err[512] = "";
extern "C" void svpc_set_error(const char* msg) { strncpy(g_err, msg, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* svpc_last_error(void) { return g_err; }
extern "C" int svpc_abi_version(void) { return 1; }
