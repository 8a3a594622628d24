// comm.hip -- the one collective of the path, inside the library: an all-gather of partial MSM sums over RCCL (xGMI), for the
// one-process-per-GPU deployment of BASELINE config 4 without torch.distributed.
//
// A large MSM shards by point range: rank r holds g[o_r, o_r + n_r) as its own kh_srs_t and reduces its slice of the scalars
// with the full single-GPU pipeline; what has to cross GPUs is ONE point per rank and MSM (72 bytes).  RCCL has no elliptic-curve reduction
// operator, so "all-reduce" = ncclAllGather of the partial points + a local fold (kh_points_sum) on every rank -- latency-bound, one collective
// per batch of MSMs.  (Several handles in ONE process need no collective at all: kh_msm_sharded.)  librccl is resolved with dlopen at
// kh_comm_init, so the library has no link-time dependency on it and single-GPU users never load it -- and no BUILD-time dependency either:
// the five entry points and the few types they take are declared here (RCCL keeps NCCL's stable C ABI: rccl.h, "ncclUniqueId",
// "ncclDataType_t", "ncclResult_t"), so a ROCm install without the RCCL headers still builds the library.
#include <dlfcn.h>

#include "common.hpp"
#include "msm.hpp"

using namespace kh;

namespace {
// ---- the slice of the NCCL / RCCL C ABI this file uses
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;                       // any other value is an error; the text comes from ncclGetErrorString
typedef enum { ncclUint64 = 5 } ncclDataType_t;                      // (ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ...)

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl R;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (R.lib) return &R;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (R.lib) break;
    }
    if (!R.lib) { set_error("librccl.so not found (%s)", dlerror()); return nullptr; }
    R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.lib, "ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.lib, "ncclCommInitRank");
    R.AllGather = (decltype(R.AllGather))dlsym(R.lib, "ncclAllGather");
    R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.lib, "ncclCommDestroy");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.AllGather || !R.CommDestroy || !R.GetErrorString) {
        set_error("librccl.so lacks an expected symbol"); dlclose(R.lib); R.lib = nullptr; return nullptr;
    }
    return &R;
}
#define KH_NCCL(R, expr)                                                                      \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) { set_error("%s failed: %s", #expr, (R)->GetErrorString(r_)); return KH_E_DEVICE; } \
    } while (0)
static_assert(KH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the C ABI as a byte string of RCCL's size");
}  // namespace

struct kh_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = -1;
    hipStream_t stream = nullptr;
    DevBuf send, recv;
    std::mutex mu;
};

extern "C" {

int kh_comm_unique_id(uint8_t id[128]) {
    KH_REQUIRE(id, "kh_comm_unique_id: null argument");
    Rccl* R = rccl(); if (!R) return KH_E_NOTFOUND;
    ncclUniqueId u;
    KH_NCCL(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, KH_COMM_ID_BYTES);
    return KH_OK;
}

int kh_comm_init(int world_size, int rank, const uint8_t id[128], kh_comm_t** out) {
    KH_REQUIRE(id && out && world_size >= 1 && rank >= 0 && rank < world_size, "kh_comm_init: bad argument (world %d, rank %d)", world_size, rank);
    int rc = ensure_init(); if (rc) return rc;
    Rccl* R = rccl(); if (!R) return KH_E_NOTFOUND;
    kh_comm* c = new (std::nothrow) kh_comm();
    KH_REQUIRE(c, "out of memory");
    c->world = world_size; c->rank = rank; c->device = ctx().device;
    ncclUniqueId u; memcpy(u.internal, id, KH_COMM_ID_BYTES);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return KH_E_DEVICE; }
    ncclResult_t r = R->CommInitRank(&c->comm, world_size, u, rank);    // collective: returns when every rank has called it
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", R->GetErrorString(r)); (void)hipStreamDestroy(c->stream); delete c; return KH_E_DEVICE; }
    *out = c;
    return KH_OK;
}

void kh_comm_free(kh_comm_t* c) {
    if (!c) return;
    DeviceScope scope(c->device);
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int kh_comm_world_size(const kh_comm_t* c) { return c ? c->world : 0; }
int kh_comm_rank(const kh_comm_t* c) { return c ? c->rank : -1; }

int kh_comm_allgather_points(kh_comm_t* c, const uint64_t* xy, const uint8_t* inf, size_t k, uint64_t* out_xy, uint8_t* out_inf) {
    KH_REQUIRE(c && (k == 0 || (xy && out_xy && out_inf)), "kh_comm_allgather_points: null argument");
    if (k == 0) return KH_OK;
    DeviceScope scope(c->device);
    int rc = ensure_init(); if (rc) return rc;
    Rccl* R = rccl(); if (!R) return KH_E_NOTFOUND;
    std::lock_guard<std::mutex> lk(c->mu);
    // one record per point: x | y | infinity flag, nine 64-bit words
    const size_t words = 9 * k;
    std::vector<uint64_t> h(words), all(words * (size_t)c->world);
    for (size_t i = 0; i < k; i++) { memcpy(&h[9 * i], xy + 8 * i, 64); h[9 * i + 8] = inf && inf[i] ? 1 : 0; }
    if ((rc = c->send.reserve(words * 8))) return rc;
    if ((rc = c->recv.reserve(words * 8 * (size_t)c->world))) return rc;
    KH_HIP(hipMemcpyAsync(c->send.p, h.data(), words * 8, hipMemcpyHostToDevice, c->stream));
    KH_NCCL(R, R->AllGather(c->send.p, c->recv.p, words, ncclUint64, c->comm, c->stream));
    KH_HIP(hipMemcpyAsync(all.data(), c->recv.p, words * 8 * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    KH_HIP(hipStreamSynchronize(c->stream));
    for (size_t j = 0; j < (size_t)c->world * k; j++) { memcpy(out_xy + 8 * j, &all[9 * j], 64); out_inf[j] = all[9 * j + 8] ? 1 : 0; }
    return KH_OK;
}

int kh_msm_allreduce(kh_comm_t* c, kh_srs_t* shard, const uint64_t* scalars, size_t n, int scalars_are_montgomery, uint64_t out_xy[8], uint8_t* out_is_inf) {
    KH_REQUIRE(c && shard && out_xy && out_is_inf && (n == 0 || scalars), "kh_msm_allreduce: null argument");
    uint64_t part[8]; uint8_t pinf = 1;
    memset(part, 0, sizeof(part));
    int rc;
    if (n) { if ((rc = kh_msm(shard, KH_BASIS_G, 0, 0, scalars, n, scalars_are_montgomery, part, &pinf))) return rc; }
    std::vector<uint64_t> all(8 * (size_t)c->world); std::vector<uint8_t> ainf(c->world);
    if ((rc = kh_comm_allgather_points(c, part, &pinf, 1, all.data(), ainf.data()))) return rc;
    return kh_points_sum(kh_srs_curve(shard), all.data(), ainf.data(), (size_t)c->world, out_xy, out_is_inf);
}

}  // extern "C"
