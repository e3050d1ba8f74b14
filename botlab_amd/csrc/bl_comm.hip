// bl_comm.hip -- the particle shards' exchange as ONE RCCL call enqueued from C++ on the filter's own stream.
//
// Per moved update every rank all-gathers the 16-byte exchange record in place.
// torch.distributed can do it (botlab_amd/sharded.py keeps that form as the fallback), but its all_gather runs on the
// process group's own stream -- two event hops around it on ours -- behind ~45 us of Python and dispatcher time per call.
// Here the collective goes onto the ctx stream directly, between k_mcl_main and the finish kernels, with no host work
// beyond the call itself.  RCCL is the library torch has already loaded (its path is handed over; nothing is linked at
// build time), reached through dlopen'd entry points of the stable NCCL 2.x C API; the communicator is this library's
// own, created from a unique id that rank 0 makes and torch.distributed broadcasts (rendezvous only).
#include <dlfcn.h>
#include <string.h>

#include "bl_internal.h"

namespace {
struct nccl_unique_id { char internal[128]; };           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
typedef int (*fn_get_unique_id)(nccl_unique_id*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, nccl_unique_id, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef const char* (*fn_error_string)(int);
const int kNcclFloat32 = 7;                               // ncclDataType_t (rccl.h)

struct rccl_api {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
};
rccl_api g_rccl;

int rccl_load(const char* path)
{
    if (g_rccl.lib) return BL_OK;
    void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { bl_set_error("dlopen(%s) failed: %s", path, dlerror()); return BL_ERR_STATE; }
    rccl_api a;
    a.lib = lib;
    a.get_unique_id = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
    a.comm_init_rank = (fn_comm_init_rank)dlsym(lib, "ncclCommInitRank");
    a.all_gather = (fn_all_gather)dlsym(lib, "ncclAllGather");
    a.comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
    a.error_string = (fn_error_string)dlsym(lib, "ncclGetErrorString");
    if (!a.get_unique_id || !a.comm_init_rank || !a.all_gather || !a.comm_destroy) {
        bl_set_error("%s does not export the NCCL entry points", path);
        return BL_ERR_STATE;
    }
    g_rccl = a;
    return BL_OK;
}

int rccl_fail(const char* what, int rc)
{
    bl_set_error("%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
    return BL_ERR_HIP;
}
}  // namespace

struct bl_comm {
    bl_ctx* ctx;
    nccl_comm_t comm;
    int rank, world;
};

// dlopen + symbol lookup only (no RCCL call): lets every rank agree that the library is usable before any of them enters the
// collective bl_comm_create.
extern "C" int bl_comm_load(const char* rccl_path)
{
    BL_CHECK_ARG(rccl_path != nullptr);
    return rccl_load(rccl_path);
}

extern "C" int bl_comm_unique_id(const char* rccl_path, char* out_id /* 128 bytes */)
{
    BL_CHECK_ARG(rccl_path != nullptr && out_id != nullptr);
    int rc = rccl_load(rccl_path);
    if (rc) return rc;
    nccl_unique_id id;
    memset(&id, 0, sizeof(id));
    int r = g_rccl.get_unique_id(&id);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(out_id, id.internal, sizeof(id.internal));
    return BL_OK;
}

// Collective over the ranks: every rank calls it with the same id.
extern "C" int bl_comm_create(bl_ctx* ctx, const char* rccl_path, const char* id128, int rank, int world, bl_comm** out)
{
    BL_CHECK_ARG(ctx != nullptr && rccl_path != nullptr && id128 != nullptr && out != nullptr && world >= 1 && rank >= 0 && rank < world);
    int rc = rccl_load(rccl_path);
    if (rc) return rc;
    BL_HIP(hipSetDevice(ctx->device));
    nccl_unique_id id;
    memcpy(id.internal, id128, sizeof(id.internal));
    nccl_comm_t comm = nullptr;
    int r = g_rccl.comm_init_rank(&comm, world, id, rank);
    if (r != 0) return rccl_fail("ncclCommInitRank", r);
    bl_comm* c = new bl_comm();
    c->ctx = ctx; c->comm = comm; c->rank = rank; c->world = world;
    *out = c;
    return BL_OK;
}

extern "C" void bl_comm_destroy(bl_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    delete c;
}

// In-place all-gather of `rec` (world x per_rank_floats floats; this rank's slice already sits at its offset) on the ctx stream.
extern "C" int bl_comm_all_gather_inplace(bl_comm* c, void* rec, size_t per_rank_floats)
{
    BL_CHECK_ARG(c != nullptr && rec != nullptr && per_rank_floats > 0);
    BL_HIP(hipSetDevice(c->ctx->device));
    const char* mine = (const char*)rec + (size_t)c->rank * per_rank_floats * sizeof(float);
    int r = g_rccl.all_gather(mine, rec, per_rank_floats, kNcclFloat32, c->comm, c->ctx->stream);
    if (r != 0) return rccl_fail("ncclAllGather", r);
    return BL_OK;
}
