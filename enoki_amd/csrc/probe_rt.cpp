// Runtime shim of libenoki-hip-probe.so (measurement scaffolding, NOT part of the product library).
//
// csrc/probe.hip is written against the internal helpers of ek_internal.h (ctx(), ensure_init(), fail(), ...).
// The product library keeps those hidden; this file provides private copies on top of the PUBLIC C ABI
// (ek_hip_init / ek_hip_stream), so that the probes launch on the product's stream without the product exporting
// anything for them.  tools/probe_*.py load the probe library through enoki_amd.capi.probe_lib().
#include "ek_internal.h"

#include <cstdlib>

namespace ek {

static Context g_probe_ctx;
static thread_local char g_probe_error[512] = "";

Context &ctx() { return g_probe_ctx; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_probe_error, sizeof(g_probe_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t err, const char *what, const char *file, int line) {
    return fail(EK_ERR_HIP, "%s failed: %s (%s:%d)", what, hipGetErrorString(err), file, line);
}

int ensure_init() {
    if (ek_hip_device() < 0)
        if (int rc = ek_hip_init(-1)) return fail(rc, "%s", ek_hip_last_error());
    Context &c = g_probe_ctx;
    c.initialized = true;
    c.device = ek_hip_device();
    c.stream = (hipStream_t) ek_hip_stream();        // the product's stream: probes interleave with product kernels
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) == hipSuccess) c.num_cu = prop.multiProcessorCount;
    return EK_OK;
}

int reduce_scratch(size_t, void **) { return fail(EK_ERR_UNSUPPORTED, "probe library: no reduction scratch"); }

void profile_mark(const char *, size_t, size_t) { }

} // namespace ek

extern "C" EK_API const char *ek_hip_probe_last_error(void) { return ek::g_probe_error; }
