// Index-range sharding for callers WITHOUT python / torch: one process per GPU, RCCL collectives on the library stream.
//
// The reference has no multi-device layer; BASELINE's north star asks for arrays sharded over the 8 GPUs of a node
// with horizontal results and table gradients finished by RCCL over xGMI.  enoki_amd/dist.py does that on top of
// torch.distributed; this file is the same exchange step at the C ABI: rank 0 makes a unique id, the caller ships its 128 bytes
// to the other ranks (a file, MPI, an environment variable -- bootstrap is the caller's business, as with ncclGetUniqueId), every
// rank calls ek_hip_dist_init, and the collectives run on the stream that the kernels run on (ordered with them, no host wait).
// librccl.so is loaded on first use: the library has no link-time dependency on it, and a world of ONE rank never loads it.
#include "ek_internal.h"

#include <dlfcn.h>
#include <link.h>

#include <cstring>
#include <string>

namespace ek {

struct Id128 { char bytes[128]; };            // ncclUniqueId (passed by value)

namespace {
struct Rccl {
    void *lib = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*comm_init_rank)(void **, int, Id128, int) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*reduce_scatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    const char *(*error_string)(int) = nullptr;
};
} // namespace

static Rccl g_rccl;
static void *g_comm = nullptr;
static int g_rank = 0, g_world = 1;

// ONE copy of RCCL per process.  A python caller has torch's own librccl.so mapped already (torch/lib/librccl.so, soname
// librccl.so.1, the copy `torch.distributed` talks to); mapping /opt/rocm/lib/librccl.so next to it would give the process two
// RCCL runtimes with separate bootstrap state.  So: (1) ENOKI_HIP_RCCL_PATH when set, (2) whatever copy is ALREADY mapped --
// found by soname without loading anything, then by walking the mapped objects --, (3) only then the loader's search path.
static int find_mapped_rccl(struct dl_phdr_info *info, size_t, void *out) {
    const char *name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char *base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) != 0) return 0;
    *static_cast<std::string *>(out) = name;
    return 1;
}

static std::string g_rccl_path = "(not loaded)";

static int load_rccl() {
    if (g_rccl.lib) return EK_OK;
    void *lib = nullptr;
    if (const char *e = getenv("ENOKI_HIP_RCCL_PATH")) {
        lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist: ENOKI_HIP_RCCL_PATH=%s cannot be loaded (%s)", e, dlerror());
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!lib) {
        std::string mapped;
        dl_iterate_phdr(find_mapped_rccl, &mapped);
        if (!mapped.empty()) lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_GLOBAL);
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist: librccl.so cannot be loaded (%s)", dlerror());
    Rccl r;
    r.lib = lib;
    r.get_unique_id = (decltype(r.get_unique_id)) dlsym(lib, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank)) dlsym(lib, "ncclCommInitRank");
    r.all_reduce = (decltype(r.all_reduce)) dlsym(lib, "ncclAllReduce");
    r.reduce_scatter = (decltype(r.reduce_scatter)) dlsym(lib, "ncclReduceScatter");
    r.all_gather = (decltype(r.all_gather)) dlsym(lib, "ncclAllGather");
    r.comm_destroy = (decltype(r.comm_destroy)) dlsym(lib, "ncclCommDestroy");
    r.error_string = (decltype(r.error_string)) dlsym(lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.reduce_scatter || !r.all_gather || !r.comm_destroy)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist: librccl.so lacks an entry point");
    g_rccl = r;
    Dl_info where;
    if (dladdr((void *) r.all_reduce, &where) && where.dli_fname) g_rccl_path = where.dli_fname;
    return EK_OK;
}

static int nccl_fail(int rc, const char *what) {
    return fail(EK_ERR_HIP, "%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
}

static int nccl_type(int type) {
    switch (type) {
        case EK_BOOL: return 1;       // ncclUint8
        case EK_I32: return 2;
        case EK_U32: return 3;
        case EK_I64: return 4;
        case EK_U64: return 5;
        case EK_F32: return 7;
        case EK_F64: return 8;
        default: return -1;
    }
}

static int nccl_op(int reduce_op) {
    switch (reduce_op) {
        case EK_HSUM: return 0;       // ncclSum
        case EK_HPROD: return 1;
        case EK_HMAX: return 2;
        case EK_HMIN: return 3;
        default: return -1;
    }
}

} // namespace ek

using namespace ek;

extern "C" {

int ek_hip_dist_unique_id(void *id128) {
    if (int rc = ensure_init()) return rc;
    if (!id128) return fail(EK_ERR_INVALID, "ek_hip_dist_unique_id(): null pointer");
    if (int rc = load_rccl()) return rc;
    if (int rc = g_rccl.get_unique_id(id128)) return nccl_fail(rc, "ncclGetUniqueId");
    return EK_OK;
}

int ek_hip_dist_init(int rank, int world, const void *id128) {
    if (int rc = ensure_init()) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(EK_ERR_INVALID, "ek_hip_dist_init(): rank %d of %d", rank, world);
    if (g_comm) return fail(EK_ERR_INVALID, "ek_hip_dist_init(): already initialised (ek_hip_dist_finalize first)");
    g_rank = rank;
    g_world = world;
    if (world == 1 && !id128) return EK_OK;                  // a world of one: every collective is local
    if (!id128) return fail(EK_ERR_INVALID, "ek_hip_dist_init(): null unique id");
    if (int rc = load_rccl()) return rc;
    // the communicator belongs to the device of the library's context, whatever device the calling thread has current
    EK_HIP_CHECK(hipSetDevice(ctx().device));
    Id128 id;
    memcpy(&id, id128, sizeof(id));
    if (int rc = g_rccl.comm_init_rank(&g_comm, world, id, rank)) { g_rank = 0; g_world = 1; return nccl_fail(rc, "ncclCommInitRank"); }
    return EK_OK;
}

int ek_hip_dist_world(int *rank, int *world) {
    if (rank) *rank = g_rank;
    if (world) *world = g_world;
    return EK_OK;
}

int ek_hip_dist_shard_range(size_t n, int rank, int world, size_t *begin, size_t *end) {
    if (world < 1 || rank < 0 || rank >= world || !begin || !end) return fail(EK_ERR_INVALID, "ek_hip_dist_shard_range(): bad arguments");
    // rank r owns [r n / P, (r + 1) n / P): the same partition for every size-n array, so vertical operations stay local
    *begin = (size_t) ((unsigned __int128) n * (unsigned) rank / (unsigned) world);
    *end = (size_t) ((unsigned __int128) n * (unsigned) (rank + 1) / (unsigned) world);
    return EK_OK;
}

int ek_hip_dist_all_reduce(int type, int reduce_op, void *buf, size_t n) {
    if (int rc = ensure_init()) return rc;
    const int t = nccl_type(type), op = nccl_op(reduce_op);
    if (t < 0 || op < 0 || (!buf && n)) return fail(EK_ERR_INVALID, "ek_hip_dist_all_reduce(): bad arguments");
    if (!g_comm) return g_world == 1 ? EK_OK : fail(EK_ERR_INVALID, "ek_hip_dist_all_reduce(): ek_hip_dist_init has not been called");
    if (n == 0) return EK_OK;
    if (refuse_while_capturing_quiet() != EK_OK)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist_all_reduce(): RCCL collectives are not recorded into step graphs (end the capture and run this step eagerly)");
    if (int rc = g_rccl.all_reduce(buf, buf, n, t, op, g_comm, ctx().stream)) return nccl_fail(rc, "ncclAllReduce");
    note_launch("dist_all_reduce", n, 2 * n * type_size(type));
    return EK_OK;
}

int ek_hip_dist_reduce_scatter(int type, int reduce_op, void *recv, const void *send, size_t recv_count) {
    if (int rc = ensure_init()) return rc;
    const int t = nccl_type(type), op = nccl_op(reduce_op);
    if (!recv || !send || t < 0 || op < 0) return fail(EK_ERR_INVALID, "ek_hip_dist_reduce_scatter(): bad arguments");
    if (!g_comm) {
        if (g_world != 1) return fail(EK_ERR_INVALID, "ek_hip_dist_reduce_scatter(): ek_hip_dist_init has not been called");
        return (recv == send || recv_count == 0) ? EK_OK : ek_hip_memcpy_device(recv, send, recv_count * type_size(type));
    }
    if (recv_count == 0) return EK_OK;
    if (refuse_while_capturing_quiet() != EK_OK)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist_reduce_scatter(): RCCL collectives are not recorded into step graphs (end the capture and run this step eagerly)");
    if (int rc = g_rccl.reduce_scatter(send, recv, recv_count, t, op, g_comm, ctx().stream)) return nccl_fail(rc, "ncclReduceScatter");
    note_launch("dist_reduce_scatter", recv_count * g_world, (size_t) (g_world + 1) * recv_count * type_size(type));
    return EK_OK;
}

int ek_hip_dist_all_gather(int type, void *recv, const void *send, size_t send_count) {
    if (int rc = ensure_init()) return rc;
    const int t = nccl_type(type);
    if (!recv || !send || t < 0) return fail(EK_ERR_INVALID, "ek_hip_dist_all_gather(): bad arguments");
    if (!g_comm) {
        if (g_world != 1) return fail(EK_ERR_INVALID, "ek_hip_dist_all_gather(): ek_hip_dist_init has not been called");
        return (recv == send || send_count == 0) ? EK_OK : ek_hip_memcpy_device(recv, send, send_count * type_size(type));
    }
    if (send_count == 0) return EK_OK;
    if (refuse_while_capturing_quiet() != EK_OK)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_dist_all_gather(): RCCL collectives are not recorded into step graphs (end the capture and run this step eagerly)");
    if (int rc = g_rccl.all_gather(send, recv, send_count, t, g_comm, ctx().stream)) return nccl_fail(rc, "ncclAllGather");
    note_launch("dist_all_gather", send_count * g_world, (size_t) (g_world + 1) * send_count * type_size(type));
    return EK_OK;
}

const char *ek_hip_dist_rccl_path(void) { return g_rccl_path.c_str(); }

int ek_hip_dist_finalize(void) {
    if (g_comm) {
        (void) hipStreamSynchronize(ctx().stream);
        g_rccl.comm_destroy(g_comm);
        g_comm = nullptr;
    }
    g_rank = 0;
    g_world = 1;
    return EK_OK;
}

} // extern "C"
