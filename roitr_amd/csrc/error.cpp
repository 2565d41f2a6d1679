// Last-error string for the C ABI (include/roitr_hip.h: roitr_last_error).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";

void roitr_set_error(const char* msg, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg ? msg : "?", file ? file : "?", line);
}

extern "C" const char* roitr_last_error(void) { return g_err; }
// 2: RoitrForwardIO::inputs_ready and RoitrGemm::a_cat_idx (round 4), RoitrLocalAttnFold::ldqt and the RoitrLocalTd operator (round 5)
// 3: RoitrGemm::batch_live, the compacted patch layout (RoitrPatch / RoitrOT / RoitrFine ::pair_off, ::slots, RoitrFine::pair_starts,
//    roitr_patch_offsets) and RoitrForwardIO::patch_slots / patch_offsets / pair_starts (round 6)
// 4: RoitrLocalBlock::wq_h / wcat_h / wout_h (round 6: bf16 matrix operands in the fused block of the bf16 operand mode)
// A client built against an older version passes shorter structs: it must check this number.
extern "C" int roitr_abi_version(void) { return 4; }

// Dynamic-LDS limit of a kernel, raised once per (kernel, device) and checked: hipFuncSetAttribute applies to the CURRENT device
// only, so a process that drives several devices needs it on each of them (one process per GPU is the normal case, but nothing
// here depends on it).  Returns ROITR_OK (0) or ROITR_ERR_HIP (2) with the error string set.
int roitr_grant_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(mu);
        int& g = granted[std::make_pair(kernel, dev)];
        if (bytes <= g) return 0;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) { g = bytes; return 0; }
    }
    roitr_set_error(hipGetErrorString(e), __FILE__, __LINE__);
    return 2;
}
