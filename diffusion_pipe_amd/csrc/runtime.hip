// runtime.hip -- error reporting and device probing for libdpipe_hip.so.
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace dpipe {
static thread_local char g_last_error[512] = "";

void set_last_error(const char* msg) {
    strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}
// Called right after a kernel launch: reports launch-configuration errors (not asynchronous faults).
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DPIPE_OK;
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
// Process-wide kernel-selection options (dpipe_set_option): A/B timing and tests of the fallback kernels.  -1 = unset: the environment variable of the
// same meaning, if any, then the built-in default (see include/dpipe_hip.h for the names).
static int g_options[DPIPE_OPTION_COUNT] = {-1, -1, -1, -1, -1, -1};
static const char* const g_option_env[DPIPE_OPTION_COUNT] = {"DPIPE_ATTN_FWD_DMA", "DPIPE_ATTN_BWD_DMA", "DPIPE_ATTN_DQ8", "DPIPE_ATTN_DKV_SPLIT",
                                                                "DPIPE_GEMM_SHALLOW", "DPIPE_GEMM_BIG_TILES"};
int option(int id, int dflt) {
    if (id < 0 || id >= DPIPE_OPTION_COUNT) return dflt;
    if (g_options[id] >= 0) return g_options[id];
    const char* e = getenv(g_option_env[id]);
    return e ? atoi(e) : dflt;
}
// DEBUG ONLY (never set by the product; results are garbage): DPIPE_DEBUG_ABLATE = bit mask of kernel classes whose launches are SKIPPED, DPIPE_DEBUG_GEMM_KDIV = d
// shortens every plain GEMM's K loop d-fold.  tools/run_gpu.sh `ablate` times the four-lane step with one class removed / halved at a time: the step's sensitivity to
// each class under concurrency, which no serialising profiler can measure.
bool ablated(int cls) {
    static const int mask = [] { const char* e = getenv("DPIPE_DEBUG_ABLATE"); return e ? atoi(e) : 0; }();
    return (mask & cls) != 0;
}
// progress marks: *mark = *gen, published to the whole device; the waiting side sleeps on the word (wall_clock64: the 100 MHz constant counter)
__global__ void mark_post_kernel(unsigned* mark, const unsigned* gen) {
    __hip_atomic_store(mark, *gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void mark_wait_kernel(const unsigned* mark, unsigned value, unsigned* err, long long timeout_ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((int)(__hip_atomic_load(mark, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (timeout_ticks > 0 && (long long)wall_clock64() - t0 > timeout_ticks) {
            if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
int ablate_gemm_kdiv() {
    static const int d = [] { const char* e = getenv("DPIPE_DEBUG_GEMM_KDIV"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    return d;
}
}  // namespace dpipe

extern "C" {
int dpipe_version(void) { return DPIPE_ABI_VERSION; }
int dpipe_set_option(int id, int value) {
    if (id < 0 || id >= DPIPE_OPTION_COUNT) { dpipe::set_last_error("dpipe_set_option: unknown option"); return DPIPE_ERR_ARG; }
    dpipe::g_options[id] = value;
    return DPIPE_OK;
}
int dpipe_get_option(int id) { return dpipe::option(id, -1); }
const char* dpipe_last_error(void) { return dpipe::g_last_error; }
// ---- progress marks (include/dpipe_hip.h C5): a device word written by a kernel NODE of a captured graph, waited for by a kernel on another stream
int dpipe_mark_post(void* mark, const void* gen, void* stream) {
    if (!mark || !gen) { dpipe::set_last_error("dpipe_mark_post: null"); return DPIPE_ERR_ARG; }
    dpipe::mark_post_kernel<<<1, 1, 0, reinterpret_cast<hipStream_t>(stream)>>>(reinterpret_cast<unsigned*>(mark), reinterpret_cast<const unsigned*>(gen));
    return dpipe::check_launch("dpipe_mark_post");
}
int dpipe_mark_wait(const void* mark, unsigned value, void* err, int timeout_ms, void* stream) {
    if (!mark || timeout_ms < 0) { dpipe::set_last_error("dpipe_mark_wait: bad argument"); return DPIPE_ERR_ARG; }
    dpipe::mark_wait_kernel<<<1, 1, 0, reinterpret_cast<hipStream_t>(stream)>>>(reinterpret_cast<const unsigned*>(mark), value, reinterpret_cast<unsigned*>(err),
                                                                           (long long)timeout_ms * 100000LL);
    return dpipe::check_launch("dpipe_mark_wait");
}
int dpipe_device_info(int dev, int* cu_count, char* arch_name, int arch_name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { dpipe::set_last_error(hipGetErrorString(e)); return (int)e; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch_name && arch_name_len > 0) { strncpy(arch_name, prop.gcnArchName, arch_name_len - 1); arch_name[arch_name_len - 1] = 0; }
    return DPIPE_OK;
}
}
