// Error-string plumbing of the C-ABI (include/svpc_hip.h: svpc_last_error).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <mutex>
static thread_local char g_err[512] = "";
extern "C" void svpc_set_error(const char* msg) { strncpy(g_err, msg, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* svpc_last_error(void) { return g_err; }
extern "C" int svpc_abi_version(void) { return 2; }

// Raise a kernel's dynamic-LDS limit to the 160 KiB of a gfx950 CU, once per kernel symbol for the life of the process: the
// attribute call must not recur on later launches (it is not capturable into a hipGraph).  One table for every source file;
// large enough for every template instantiation of the library (the per-file tables of 8/16 entries could overflow).
extern "C" int svpc_raise_lds_once(const void* fn, const char* who) {
    static const void* done[1024];
    static int n_done = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_done; ++i) if (done[i] == fn) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: cannot raise the dynamic LDS limit (%s)", who, hipGetErrorString(e));
        svpc_set_error(buf);
        return (int)e;
    }
    if (n_done < 1024) done[n_done++] = fn;
    return 0;
}
