// libenoki-hip.so runtime: device/stream context, error reporting, caching allocator, copies.
//
// Replaces the non-JIT half of the reference's src/cuda/jit.cu (Context 149-262, caching
// allocator 1683-1896) and src/cuda/common.cu (memcpy wrappers 104-122, error policy 268-286).
// There is no trace/variable table: arrays are plain device buffers owned by HIPArray<T>.
#include "ek_internal.h"

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace ek {

static thread_local std::string t_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_last_error = buf;
    if (ctx().log_level >= 1)
        fprintf(stderr, "enoki-hip: error: %s\n", buf);
    return code;
}

int hip_fail(hipError_t err, const char *what, const char *file, int line) {
    return fail(EK_ERR_HIP, "%s failed: %s (%s:%d)", what, hipGetErrorString(err), file, line);
}

// Context and allocator are intentionally leaked: arrays owned by other shared objects (the global tape,
// Python modules) may be released during static destruction, after this library's statics would be gone.
Context &ctx() {
    static Context *c = new Context();
    return *c;
}

// ------------------------------------------------------------------------------------------------
//  Caching allocator.  Size classes: powers of two up to 1 MiB, multiples of 1 MiB above (a
//  64 Mi-element f32 array is exactly 256 MiB either way).  Freed blocks go to a per-class free
//  list and are reused in stream order (single stream => no event bookkeeping needed); on
//  out-of-memory the cache is released and the allocation retried once (jit.cu:1716-1723).
// ------------------------------------------------------------------------------------------------
// Blocks that a captured step graph (ek_hip_graph_*) touches form the graph's private POOL: the graph has their
// addresses baked in, so until the graph is destroyed they are only ever handed out again to allocations made while
// capturing that same graph -- never to eager code, whose data a replay would overwrite.
struct GraphPool {
    std::unordered_map<size_t, std::vector<void *>> free_lists;
    std::vector<void *> blocks;                                   // every block that ever entered the pool
};

struct Allocator {
    std::mutex mutex;
    std::unordered_map<size_t, std::vector<void *>> free_lists;   // class size -> blocks
    std::unordered_map<void *, size_t> live;                      // block -> class size
    std::unordered_map<void *, GraphPool *> pool_of;              // blocks owned by a graph pool (live or free)
    GraphPool *capture_pool = nullptr;                            // non-null between ek_hip_graph_begin / _end
    size_t live_bytes = 0, cached_bytes = 0, watermark = 0, n_malloc = 0, n_reuse = 0;

    static size_t round_size(size_t bytes) {
        if (bytes <= 256) return 256;
        if (bytes <= (size_t(1) << 20)) {
            size_t s = 256;
            while (s < bytes) s <<= 1;
            return s;
        }
        const size_t mib = size_t(1) << 20;
        return (bytes + mib - 1) / mib * mib;
    }

    hipError_t trim_locked() {
        for (auto &kv : free_lists)
            for (void *p : kv.second) {
                hipError_t e = hipFree(p);
                if (e != hipSuccess) return e;
            }
        free_lists.clear();
        cached_bytes = 0;
        return hipSuccess;
    }
};

static Allocator &alloc() {
    static Allocator *a = new Allocator();
    return *a;
}

int ensure_init() {
    if (ctx().initialized) return EK_OK;
    return ek_hip_init(-1);
}

int reduce_scratch(size_t bytes, void **out) {
    Context &c = ctx();
    if (c.reduce_scratch_bytes < bytes) {
        if (c.reduce_scratch) {
            // the old scratch may still be in use by enqueued kernels: free is stream-ordered
            // through the caching allocator (same stream), so handing it back is safe.
            ek_hip_free(c.reduce_scratch);
            c.reduce_scratch = nullptr;
            c.reduce_scratch_bytes = 0;
        }
        size_t want = bytes < 65536 ? 65536 : bytes;
        int rc = ek_hip_malloc(want, &c.reduce_scratch);
        if (rc) return rc;
        c.reduce_scratch_bytes = want;
    }
    *out = c.reduce_scratch;
    return EK_OK;
}

// ------------------------------------------------------------------------------------------------
//  Launch profiling: one hipEvent after every kernel launch; the time between consecutive events is
//  attributed to the later launch (launches are back to back on one stream, so this is the kernel's
//  duration plus the inter-kernel gap).  Used by bench.py for the `roofline` object.
// ------------------------------------------------------------------------------------------------
struct ProfileRecord { const char *name; size_t n, bytes; hipEvent_t event; };
static std::vector<ProfileRecord> g_profile;
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t g_profile_start = nullptr;

static hipEvent_t pool_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}

void profile_mark(const char *name, size_t n, size_t bytes) {
    hipEvent_t e = pool_event();
    if (!e) return;
    (void) hipEventRecord(e, ctx().stream);
    g_profile.push_back(ProfileRecord{ name, n, bytes, e });
}

// roctx through dlopen (see ek_internal.h)
static void (*g_roctx_push)(const char *) = nullptr;
static void (*g_roctx_pop)() = nullptr;
static int g_roctx_state = 0;      // 0: not looked at, 1: active, -1: off

static void roctx_load() {
    g_roctx_state = -1;
    const char *e = getenv("ENOKI_HIP_ROCTX");
    if (!e || e[0] == '0') return;
    void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    g_roctx_push = (void (*)(const char *)) dlsym(lib, "roctxRangePushA");
    g_roctx_pop = (void (*)()) dlsym(lib, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) g_roctx_state = 1;
}

void roctx_push(const char *name) {
    if (g_roctx_state == 0) roctx_load();
    if (g_roctx_state == 1) g_roctx_push(name);
}

void roctx_pop() {
    if (g_roctx_state == 1) g_roctx_pop();
}

} // namespace ek

using namespace ek;

extern "C" {

int ek_hip_init(int device) {
    Context &c = ctx();
    if (c.initialized && (device < 0 || device == c.device)) return EK_OK;
    int count = 0;
    EK_HIP_CHECK(hipGetDeviceCount(&count));
    if (count == 0) return fail(EK_ERR_HIP, "ek_hip_init(): no HIP device visible");
    if (device < 0) {
        // honour LOCAL_RANK so that one process per GPU needs no extra plumbing
        const char *lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) % count : 0;
    }
    if (device >= count) return fail(EK_ERR_INVALID, "ek_hip_init(): device %d out of range (%d visible)", device, count);
    if (c.initialized) {
        // Switching devices: blocks of the old device must not be handed to kernels of the new one.  The free lists are
        // keyed by size only, so the cache (and the reduction scratch, which lives in it) is released on the old device
        // first; arrays that are still alive would dangle, so the switch is refused while any exist.
        EK_HIP_CHECK(hipStreamSynchronize(c.stream));
        release_meta_ring();
        if (c.reduce_scratch) {
            ek_hip_free(c.reduce_scratch);
            c.reduce_scratch = nullptr;
            c.reduce_scratch_bytes = 0;
        }
        {
            Allocator &a = alloc();
            std::lock_guard<std::mutex> guard(a.mutex);
            if (!a.live.empty())
                return fail(EK_ERR_INVALID, "ek_hip_init(): cannot switch from device %d to %d while %zu allocations are alive",
                            c.device, device, a.live.size());
            EK_HIP_CHECK(a.trim_locked());
        }
        if (c.owns_stream) EK_HIP_CHECK(hipStreamDestroy(c.stream));
        c.stream = nullptr;
    }
    EK_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EK_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c.device = device;
    c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    EK_HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    c.owns_stream = true;
    c.initialized = true;
    // A process that returns from main() with kernels still in flight lets the HIP runtime tear its queues down underneath
    // them (seen once as an abort in an HSA completion thread after a test binary had printed its last result).  Registered
    // after the runtime's own exit handlers, so it runs before them; a stream the caller supplied is the caller's to drain.
    static bool exit_hook = false;
    if (!exit_hook) {
        exit_hook = true;
        atexit([]() {
            Context &x = ctx();
            if (x.initialized && x.owns_stream && x.stream) (void) hipStreamSynchronize(x.stream);
        });
    }
    if (const char *lv = getenv("ENOKI_HIP_LOG")) c.log_level = (uint32_t) atoi(lv);
    if (const char *dv = getenv("ENOKI_HIP_DETERMINISTIC")) c.tuning.deterministic = atoi(dv) != 0;
    if (const char *bo = getenv("ENOKI_HIP_BUCKET_ORDERED")) c.tuning.bucket_ordered = atoi(bo) != 0;
    if (const char *ea = getenv("ENOKI_HIP_EARLY_ADJOINT")) c.tuning.early_adjoint = atoi(ea) != 0;
    if (const char *xb = getenv("ENOKI_HIP_XCD_BALANCE")) c.tuning.xcd_balance = atoi(xb) == 2 ? 2 : atoi(xb) != 0;
    if (const char *gr = getenv("ENOKI_HIP_GATHER_RECORDS")) { int v = atoi(gr); if (v >= 0 && v <= 2) c.tuning.gather_records = v; }
    if (c.log_level >= 1)
        fprintf(stderr, "enoki-hip: device %d (%s, %d CUs, %.1f GiB)\n", device, prop.name, c.num_cu,
                (double) prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return EK_OK;
}

int ek_hip_device(void) { return ctx().initialized ? ctx().device : -1; }

int ek_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

void *ek_hip_stream(void) { return (void *) ctx().stream; }

int ek_hip_set_stream(void *s) {
    int rc = ensure_init();
    if (rc) return rc;
    Context &c = ctx();
    EK_HIP_CHECK(hipStreamSynchronize(c.stream));   // cached blocks may be reused on the new stream
    if (c.owns_stream) EK_HIP_CHECK(hipStreamDestroy(c.stream));
    c.stream = (hipStream_t) s;
    c.owns_stream = false;
    return EK_OK;
}

int ek_hip_sync(void) {
    int rc = ensure_init();
    if (rc) return rc;
    if (int busy = refuse_while_capturing("ek_hip_sync()")) return busy;
    EK_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    return EK_OK;
}

const char *ek_hip_last_error(void) { return t_last_error.c_str(); }

int ek_hip_malloc(size_t bytes, void **out) {
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_malloc(): null output pointer");
    int rc = ensure_init();
    if (rc) return rc;
    Allocator &a = alloc();
    size_t cls = Allocator::round_size(bytes);
    std::lock_guard<std::mutex> guard(a.mutex);
    void *ptr = nullptr;
    GraphPool *pool = a.capture_pool;
    if (pool) {
        auto pit = pool->free_lists.find(cls);
        if (pit != pool->free_lists.end() && !pit->second.empty()) {
            ptr = pit->second.back();
            pit->second.pop_back();
            a.n_reuse++;
        }
    }
    auto it = a.free_lists.find(cls);
    if (ptr) {
        /* reused inside the graph's pool */
    } else if (it != a.free_lists.end() && !it->second.empty()) {
        ptr = it->second.back();
        it->second.pop_back();
        a.cached_bytes -= cls;
        a.n_reuse++;
    } else {
        hipError_t e = hipMalloc(&ptr, cls);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            // trimming needs the stream drained -- which cannot be done (and would invalidate the capture) while a step graph is
            // being captured: fail cleanly instead
            if (int busy = refuse_while_capturing("ek_hip_malloc(): out of device memory, and trimming the cache")) return busy;
            hipError_t e2 = hipStreamSynchronize(ctx().stream);
            if (e2 == hipSuccess) e2 = a.trim_locked();
            if (e2 == hipSuccess) e = hipMalloc(&ptr, cls);
            if (e != hipSuccess) {
                (void) hipGetLastError();
                return fail(EK_ERR_OOM, "ek_hip_malloc(): out of memory allocating %zu bytes (%zu live, %zu cached)",
                            cls, a.live_bytes, a.cached_bytes);
            }
        }
        a.n_malloc++;
    }
    if (pool && !a.pool_of.count(ptr)) {
        a.pool_of[ptr] = pool;
        pool->blocks.push_back(ptr);
    }
    a.live[ptr] = cls;
    a.live_bytes += cls;
    if (a.live_bytes > a.watermark) a.watermark = a.live_bytes;
    *out = ptr;
    return EK_OK;
}

int ek_hip_free(void *ptr) {
    if (!ptr) return EK_OK;
    Allocator &a = alloc();
    std::lock_guard<std::mutex> guard(a.mutex);
    auto it = a.live.find(ptr);
    if (it == a.live.end())
        return fail(EK_ERR_INVALID, "ek_hip_free(): pointer %p was not allocated by ek_hip_malloc()", ptr);
    size_t cls = it->second;
    a.live.erase(it);
    a.live_bytes -= cls;
    auto pit = a.pool_of.find(ptr);
    if (pit != a.pool_of.end()) {
        pit->second->free_lists[cls].push_back(ptr);     // stays reserved for its graph
    } else {
        a.free_lists[cls].push_back(ptr);
        a.cached_bytes += cls;
    }
    return EK_OK;
}

int ek_hip_malloc_trim(void) {
    int rc = ensure_init();
    if (rc) return rc;
    Allocator &a = alloc();
    if (int busy = refuse_while_capturing("ek_hip_malloc_trim()")) return busy;
    EK_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    std::lock_guard<std::mutex> guard(a.mutex);
    EK_HIP_CHECK(a.trim_locked());
    return EK_OK;
}

int ek_hip_host_malloc(size_t bytes, void **out) {
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_host_malloc(): null output pointer");
    int rc = ensure_init();
    if (rc) return rc;
    EK_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return EK_OK;
}

int ek_hip_host_free(void *ptr) {
    if (!ptr) return EK_OK;
    EK_HIP_CHECK(hipHostFree(ptr));
    return EK_OK;
}

int ek_hip_mem_get_info(size_t *free_bytes, size_t *total_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    size_t f = 0, t = 0;
    EK_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return EK_OK;
}

int ek_hip_memcpy_to_device(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!bytes) return EK_OK;
    if (int busy = refuse_while_capturing("ek_hip_memcpy_to_device()")) return busy;
    EK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream));
    EK_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    return EK_OK;
}

int ek_hip_memcpy_to_host(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!bytes) return EK_OK;
    if (alloc().capture_pool)
        return fail(EK_ERR_INVALID, "ek_hip_memcpy_to_host(): device -> host reads (coeff, count, any, all, to_host) cannot be "
                                    "part of a captured step graph");
    EK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
    EK_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    return EK_OK;
}

int ek_hip_memcpy_device(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!bytes) return EK_OK;
    EK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream));
    return EK_OK;
}

int ek_hip_memset(void *dst, int byte, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!bytes) return EK_OK;
    EK_HIP_CHECK(hipMemsetAsync(dst, byte, bytes, ctx().stream));
    return EK_OK;
}

char *ek_hip_whos(void) {
    Allocator &a = alloc();
    std::lock_guard<std::mutex> guard(a.mutex);
    char buf[512];
    snprintf(buf, sizeof(buf),
             "\n  enoki-hip allocator (device %d)\n"
             "  ===============================\n"
             "  live blocks      : %zu\n"
             "  live bytes       : %zu\n"
             "  cached bytes     : %zu\n"
             "  max. live bytes  : %zu\n"
             "  hipMalloc calls  : %zu\n"
             "  cache hits       : %zu\n"
             "  kernel launches  : %llu\n",
             ctx().device, a.live.size(), a.live_bytes, a.cached_bytes, a.watermark, a.n_malloc, a.n_reuse,
             (unsigned long long) ctx().launches);
    return strdup(buf);
}

void ek_hip_set_log_level(uint32_t level) { ctx().log_level = level; }
uint32_t ek_hip_log_level(void) { return ctx().log_level; }
uint64_t ek_hip_launch_count(void) { return ctx().launches; }

// ---------------------------------------------------------------------------------------------
//  Step graphs: capture the launches of one step once, replay them without host work
// ---------------------------------------------------------------------------------------------
struct ek_hip_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    GraphPool pool;
    void *reduce_scratch = nullptr;       // the graph's private reduction scratch
    size_t reduce_scratch_bytes = 0;
    uint64_t launches = 0;                // kernel launches captured
};

static ek_hip_graph *g_capturing = nullptr;

} // extern "C"
int ek::refuse_while_capturing_quiet() { return g_capturing ? EK_ERR_UNSUPPORTED : EK_OK; }

int ek::refuse_while_capturing(const char *what) {
    if (!g_capturing) return EK_OK;
    return fail(EK_ERR_INVALID, "%s: the host would have to wait for the device, which cannot be part of a captured step graph "
                                "(end the capture and run this step eagerly)", what);
}
extern "C" {
static void *g_stashed_scratch = nullptr;
static size_t g_stashed_scratch_bytes = 0;
static uint64_t g_capture_launch_base = 0;

int ek_hip_graph_begin(void) {
    if (int rc = ensure_init()) return rc;
    Context &c = ctx();
    if (g_capturing) return fail(EK_ERR_INVALID, "ek_hip_graph_begin(): a capture is already in progress");
    if (c.profiling) return fail(EK_ERR_INVALID, "ek_hip_graph_begin(): stop ek_hip_profile_* first (events are not captured)");
    ek_hip_graph *g = new ek_hip_graph();
    {
        Allocator &a = alloc();
        std::lock_guard<std::mutex> guard(a.mutex);
        a.capture_pool = &g->pool;
    }
    // the shared reduction scratch may be re-allocated by later eager code: the graph gets its own
    g_stashed_scratch = c.reduce_scratch; g_stashed_scratch_bytes = c.reduce_scratch_bytes;
    c.reduce_scratch = nullptr; c.reduce_scratch_bytes = 0;
    g_capture_launch_base = c.launches;
    hipError_t e = hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        { Allocator &a = alloc(); std::lock_guard<std::mutex> guard(a.mutex); a.capture_pool = nullptr; }
        c.reduce_scratch = g_stashed_scratch; c.reduce_scratch_bytes = g_stashed_scratch_bytes;
        delete g;
        (void) hipGetLastError();          // (do not leave the failure behind as a sticky launch error)
        return hip_fail(e, "hipStreamBeginCapture", __FILE__, __LINE__);
    }
    g_capturing = g;
    return EK_OK;
}

static void graph_release_pool(ek_hip_graph *g) {
    Allocator &a = alloc();
    std::lock_guard<std::mutex> guard(a.mutex);
    for (auto &kv : g->pool.free_lists)
        for (void *p : kv.second) {
            a.free_lists[kv.first].push_back(p);
            a.cached_bytes += kv.first;
        }
    g->pool.free_lists.clear();
    for (void *p : g->pool.blocks) a.pool_of.erase(p);      // blocks that are still alive become ordinary blocks
    g->pool.blocks.clear();
}

int ek_hip_graph_end(ek_hip_graph **out) {
    Context &c = ctx();
    if (!g_capturing) return fail(EK_ERR_INVALID, "ek_hip_graph_end(): no capture in progress");
    ek_hip_graph *g = g_capturing;
    g_capturing = nullptr;
    hipError_t e = hipStreamEndCapture(c.stream, &g->graph);
    { Allocator &a = alloc(); std::lock_guard<std::mutex> guard(a.mutex); a.capture_pool = nullptr; }
    g->reduce_scratch = c.reduce_scratch; g->reduce_scratch_bytes = c.reduce_scratch_bytes;
    c.reduce_scratch = g_stashed_scratch; c.reduce_scratch_bytes = g_stashed_scratch_bytes;
    g->launches = c.launches - g_capture_launch_base;
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess || !out) {
        (void) hipGetLastError();
        ek_hip_graph_destroy(g);
        if (!out) return fail(EK_ERR_INVALID, "ek_hip_graph_end(): null output pointer");
        return hip_fail(e, "hipStreamEndCapture / hipGraphInstantiate", __FILE__, __LINE__);
    }
    *out = g;
    return EK_OK;
}

int ek_hip_graph_launch(ek_hip_graph *g) {
    if (!g || !g->exec) return fail(EK_ERR_INVALID, "ek_hip_graph_launch(): invalid graph");
    Context &c = ctx();
    EK_HIP_CHECK(hipGraphLaunch(g->exec, c.stream));
    c.launches += g->launches;
    return EK_OK;
}

uint64_t ek_hip_graph_launch_count(const ek_hip_graph *g) { return g ? g->launches : 0; }

int ek_hip_graph_destroy(ek_hip_graph *g) {
    if (!g) return EK_OK;
    // a replay of this graph may still be running: wait for it -- unless ANOTHER graph is being captured on the stream right
    // now, where a synchronisation would invalidate that capture
    if (int busy = refuse_while_capturing("ek_hip_graph_destroy()")) return busy;
    if (ctx().initialized) (void) hipStreamSynchronize(ctx().stream);
    if (g->exec) (void) hipGraphExecDestroy(g->exec);
    if (g->graph) (void) hipGraphDestroy(g->graph);
    if (g->reduce_scratch) ek_hip_free(g->reduce_scratch);
    graph_release_pool(g);
    delete g;
    return EK_OK;
}

void **ek_hip_binding_slot(void) {
    static void *slot = nullptr;
    return &slot;
}

int ek_hip_note_launch(const char *name, size_t n, size_t bytes) {
    if (int rc = ensure_init()) return rc;
    if (!name) return fail(EK_ERR_INVALID, "ek_hip_note_launch(): null name");
    EK_LAUNCH_CHECK(name, n, bytes);
    return EK_OK;
}

int ek_hip_profile_begin(void) {
    int rc = ensure_init();
    if (rc) return rc;
    Context &c = ctx();
    for (auto &r : g_profile) g_event_pool.push_back(r.event);
    g_profile.clear();
    if (!g_profile_start) EK_HIP_CHECK(hipEventCreate(&g_profile_start));
    EK_HIP_CHECK(hipEventRecord(g_profile_start, c.stream));
    c.profiling = true;
    return EK_OK;
}

char *ek_hip_profile_end(void) {
    Context &c = ctx();
    c.profiling = false;
    if (c.initialized) (void) hipStreamSynchronize(c.stream);
    struct Agg { size_t launches = 0, bytes = 0, n = 0; double ms = 0; };
    std::vector<std::pair<std::string, Agg>> aggs;
    hipEvent_t prev = g_profile_start;
    for (auto &r : g_profile) {
        float ms = 0.f;
        if (prev && r.event) (void) hipEventElapsedTime(&ms, prev, r.event);
        prev = r.event;
        Agg *a = nullptr;
        // launches of the same kernel over different sizes / operand shapes are different roofline points
        std::string key = std::string(r.name) + "/" + std::to_string(r.n) + "/" + std::to_string(r.bytes);
        for (auto &kv : aggs) if (kv.first == key) { a = &kv.second; break; }
        if (!a) { aggs.emplace_back(key, Agg()); a = &aggs.back().second; }
        a->launches++; a->bytes += r.bytes; a->n += r.n; a->ms += ms;
    }
    for (auto &r : g_profile) g_event_pool.push_back(r.event);
    g_profile.clear();
    std::string out = "[";
    char buf[512];
    for (size_t i = 0; i < aggs.size(); ++i) {
        const Agg &a = aggs[i].second;
        snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %zu, \"total_ms\": %.6f, \"bytes\": %zu, \"elements\": %zu}",
                 i ? ", " : "", aggs[i].first.substr(0, aggs[i].first.find('/')).c_str(), a.launches, a.ms, a.bytes, a.n);
        out += buf;
    }
    out += "]";
    return strdup(out.c_str());
}

int ek_hip_set_tuning(const char *key, int value) {
    if (!key) return fail(EK_ERR_INVALID, "ek_hip_set_tuning(): null key");
    Tuning &t = ctx().tuning;
    if (!strcmp(key, "blocks_per_cu") && value > 0) t.blocks_per_cu = value;
    else if (!strcmp(key, "reduce_blocks_per_cu") && value > 0) t.reduce_blocks_per_cu = value;
    else if (!strcmp(key, "scatter_add_binned") && (value == 0 || value == 1)) t.scatter_add_binned = value;
    else if (!strcmp(key, "deterministic") && (value == 0 || value == 1)) t.deterministic = value;
    else if (!strcmp(key, "gather_records") && value >= 0 && value <= 2) t.gather_records = value;
    else if (!strcmp(key, "bucket_ordered") && (value == 0 || value == 1)) t.bucket_ordered = value;
    else if (!strcmp(key, "early_adjoint") && (value == 0 || value == 1)) t.early_adjoint = value;
    else if (!strcmp(key, "xcd_balance") && value >= 0 && value <= 2) t.xcd_balance = value;
    else return fail(EK_ERR_INVALID, "ek_hip_set_tuning(): unknown key/value %s=%d", key, value);
    return EK_OK;
}

} // extern "C"
