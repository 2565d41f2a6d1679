// Last-error string for the C ABI (include/roitr_hip.h: roitr_last_error).
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void roitr_set_error(const char* msg, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg ? msg : "?", file ? file : "?", line);
}

extern "C" const char* roitr_last_error(void) { return g_err; }
extern "C" int roitr_abi_version(void) { return 1; }
