// Multi-GPU transport of libdarsia_hip: neighbour send/recv of halo rows between row slabs, one process per GPU.
//
// The reference is single-process, so nothing here mirrors reference code.  What the exchange has to
// respect are the data dependences of one outer iteration: the Jacobi neighbour sum reads u one row above and below
// (utils/linear_solvers/jacobi.py:43-54), the right-hand side takes a one-sided difference of ell (b0 - d0) across
// the row above (restoration/split_bregman_tvd.py:84-89) and the split update needs D+ u_new of the row below (:144).
//
// Two transports sit behind the same entry points:
//   * RCCL (kind 0): grouped ncclSend / ncclRecv of the gathered halo rows -- one message per direction -- on the stream
//     the caller names (dxh_tvd.hip alternates a slab context between two streams); on MI355X these go peer to peer
//     over xGMI.  The communicator's own stream serves the small all-reduces.  RCCL is loaded with dlopen on first use,
//     so single-GPU users never pay for (or depend on) librccl.
//   * host callbacks (kind 1): the rows are staged through pinned host memory and handed to a caller-provided
//     function that moves bytes between ranks (tests and rehearsals on boxes with fewer GPUs than ranks: RCCL refuses
//     two ranks on one device).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "dxh_comm.h"

namespace dxh {

// ---- the part of the RCCL API used here: types and prototypes from rccl.h, the functions themselves through dlsym ----
namespace rccl {
typedef ncclUniqueId UniqueId;
typedef ncclComm_t CommT;
constexpr ncclResult_t kSuccess = ncclSuccess;
constexpr ncclDataType_t kChar = ncclChar, kFloat64 = ncclFloat64;
constexpr ncclRedOp_t kSum = ncclSum, kMax = ncclMax;
struct Api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
static Api g_api;
static std::mutex g_mu;

static int load() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_api.handle) return DXH_OK;
    // RCCL must sit on the SAME HIP runtime as this library (streams, events and pointers cross between them).  A process
    // can hold two ROCm stacks -- e.g. PyTorch's bundled one next to /opt/rocm -- and a bare dlopen("librccl.so.1")
    // returns whichever copy was loaded first.  So: find the file our own HIP symbols come from and take the librccl
    // beside it, by absolute path, bound to its own dependency chain (RTLD_DEEPBIND).
    std::string beside1, beside2;
    {
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                beside1 = dir + "/librccl.so.1";
                beside2 = dir + "/librccl.so";
            }
        }
    }
    const char* names[] = {getenv("DXH_RCCL_LIB"), beside1.c_str(), beside2.c_str(), "librccl.so.1",
                           "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (h) break;
    }
    if (!h) {
        set_error("cannot load librccl (%s): multi-GPU runs need RCCL", dlerror());
        return DXH_EHIP;
    }
    Api a;
    a.handle = h;
#define DXH_SYM(field, name)                                             \
    *(void**)(&a.field) = dlsym(h, name);                                \
    if (!a.field) {                                                      \
        set_error("librccl lacks %s", name);                             \
        dlclose(h);                                                      \
        return DXH_EHIP;                                                 \
    }
    DXH_SYM(GetUniqueId, "ncclGetUniqueId")
    DXH_SYM(CommInitRank, "ncclCommInitRank")
    DXH_SYM(CommDestroy, "ncclCommDestroy")
    DXH_SYM(GetErrorString, "ncclGetErrorString")
    DXH_SYM(GroupStart, "ncclGroupStart")
    DXH_SYM(GroupEnd, "ncclGroupEnd")
    DXH_SYM(Send, "ncclSend")
    DXH_SYM(Recv, "ncclRecv")
    DXH_SYM(AllReduce, "ncclAllReduce")
    DXH_SYM(GetVersion, "ncclGetVersion")
#undef DXH_SYM
    g_api = a;
    return DXH_OK;
}
}  // namespace rccl

#define DXH_NCCL(call)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                                    \
        if (r_ != rccl::kSuccess) {                                                                           \
            ::dxh::set_error("%s failed: %s (%s:%d)", #call, rccl::g_api.GetErrorString(r_), __FILE__, __LINE__); \
            return DXH_EHIP;                                                                                  \
        }                                                                                                     \
    } while (0)

int comm_exchange(Comm* cm, const P2POp* ops, int n, hipStream_t stream) {
    if (n <= 0) return DXH_OK;
    cm->exchanges += 1;
    for (int i = 0; i < n; ++i)
        if (ops[i].is_send) cm->bytes_sent += (int64_t)ops[i].bytes;
    if (cm->kind == 0) {
        DXH_NCCL(rccl::g_api.GroupStart());
        for (int i = 0; i < n; ++i) {
            ncclResult_t r = ops[i].is_send
                        ? rccl::g_api.Send(ops[i].dev, ops[i].bytes, rccl::kChar, ops[i].peer, (rccl::CommT)cm->nccl, stream)
                        : rccl::g_api.Recv(ops[i].dev, ops[i].bytes, rccl::kChar, ops[i].peer, (rccl::CommT)cm->nccl, stream);
            if (r != rccl::kSuccess) {
                (void)rccl::g_api.GroupEnd();
                set_error("ncclSend/ncclRecv failed: %s", rccl::g_api.GetErrorString(r));
                return DXH_EHIP;
            }
        }
        DXH_NCCL(rccl::g_api.GroupEnd());
        return DXH_OK;
    }
    // host-staged: device -> pinned host, caller's transport, pinned host -> device
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (ops[i].bytes + 255) / 256 * 256;
    if (cm->h_stage_bytes < total) {
        if (cm->h_stage) (void)hipHostFree(cm->h_stage);
        cm->h_stage = nullptr;
        cm->h_stage_bytes = 0;
        DXH_HIP(hipHostMalloc((void**)&cm->h_stage, total, hipHostMallocDefault));
        cm->h_stage_bytes = total;
    }
    std::vector<int> peers(n), is_send(n);
    std::vector<void*> bufs(n);
    std::vector<size_t> bytes(n);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        peers[i] = ops[i].peer;
        is_send[i] = ops[i].is_send;
        bufs[i] = cm->h_stage + off;
        bytes[i] = ops[i].bytes;
        if (ops[i].is_send) DXH_HIP(hipMemcpyAsync(bufs[i], ops[i].dev, ops[i].bytes, hipMemcpyDeviceToHost, stream));
        off += (ops[i].bytes + 255) / 256 * 256;
    }
    DXH_HIP(hipStreamSynchronize(stream));
    int rc = cm->xfn(cm->user, n, peers.data(), is_send.data(), bufs.data(), bytes.data());
    if (rc != 0) {
        set_error("the host transport callback failed with code %d", rc);
        return DXH_EHIP;
    }
    for (int i = 0; i < n; ++i)
        if (!ops[i].is_send) DXH_HIP(hipMemcpyAsync(ops[i].dev, bufs[i], ops[i].bytes, hipMemcpyHostToDevice, stream));
    // the staging buffer is reused by the next exchange: the copies must have left it
    DXH_HIP(hipStreamSynchronize(stream));
    return DXH_OK;
}

int comm_allreduce(Comm* cm, double* vals, int n, int op) {
    DXH_REQUIRE(n >= 1 && n <= 16, "all-reduce of 1..16 values");
    DXH_REQUIRE(op == 0 || op == 1, "op: 0 = sum, 1 = max");
    if (cm->kind == 1) {
        if (cm->world == 1) return DXH_OK;
        int rc = cm->afn(cm->user, vals, n, op);
        if (rc != 0) {
            set_error("the host all-reduce callback failed with code %d", rc);
            return DXH_EHIP;
        }
        return DXH_OK;
    }
    DXH_HIP(hipMemcpyAsync(cm->d_red, vals, sizeof(double) * n, hipMemcpyHostToDevice, cm->stream));
    DXH_NCCL(rccl::g_api.AllReduce(cm->d_red, cm->d_red, (size_t)n, rccl::kFloat64, op == 0 ? rccl::kSum : rccl::kMax,
                                   (rccl::CommT)cm->nccl, cm->stream));
    DXH_HIP(hipMemcpyAsync(vals, cm->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, cm->stream));
    DXH_HIP(hipStreamSynchronize(cm->stream));
    return DXH_OK;
}

static int comm_common_init(Comm* cm) {
    DXH_HIP(hipSetDevice(cm->device));
    // the exchange should start the moment the interface rows are ready: highest priority the device offers
    int lo = 0, hi = 0;
    DXH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    DXH_HIP(hipStreamCreateWithPriority(&cm->stream, hipStreamNonBlocking, hi));
    DXH_HIP(hipMalloc((void**)&cm->d_red, sizeof(double) * 16));
    return DXH_OK;
}

static void comm_free(Comm* cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->nccl) (void)rccl::g_api.CommDestroy((rccl::CommT)cm->nccl);
    if (cm->d_red) (void)hipFree(cm->d_red);
    if (cm->h_stage) (void)hipHostFree(cm->h_stage);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
}

void comm_release(Comm* cm) {
    if (!cm) return;
    if (--cm->refs <= 0 && cm->dead) comm_free(cm);
}

}  // namespace dxh

using namespace dxh;

extern "C" {

int dxh_comm_unique_id(void* id) {
    DXH_REQUIRE(id, "null argument");
    int rc = rccl::load();
    if (rc) return rc;
    rccl::UniqueId u;
    DXH_NCCL(rccl::g_api.GetUniqueId(&u));
    static_assert(sizeof(u) == DXH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof u);
    return DXH_OK;
}

int dxh_comm_create_rccl(dxh_comm_t* out, int device, const void* id, int rank, int world) {
    DXH_REQUIRE(out && id, "null argument");
    DXH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    int rc = rccl::load();
    if (rc) return rc;
    Comm* cm = new (std::nothrow) Comm();
    if (!cm) { set_error("out of host memory"); return DXH_ENOMEM; }
    cm->kind = 0;
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    rc = comm_common_init(cm);
    if (rc) { comm_free(cm); return rc; }
    rccl::UniqueId u;
    memcpy(&u, id, sizeof u);
    rccl::CommT nc = nullptr;
    ncclResult_t r = rccl::g_api.CommInitRank(&nc, world, u, rank);
    cm->nccl = nc;
    if (r != rccl::kSuccess) {
        set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, rccl::g_api.GetErrorString(r));
        cm->nccl = nullptr;
        comm_free(cm);
        return DXH_EHIP;
    }
    *out = reinterpret_cast<dxh_comm_t>(cm);
    return DXH_OK;
}

int dxh_comm_create_host(dxh_comm_t* out, int device, int rank, int world, dxh_comm_exchange_fn exchange,
                         dxh_comm_allreduce_fn allreduce, void* user) {
    DXH_REQUIRE(out && exchange && allreduce, "null argument");
    DXH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    Comm* cm = new (std::nothrow) Comm();
    if (!cm) { set_error("out of host memory"); return DXH_ENOMEM; }
    cm->kind = 1;
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    cm->xfn = exchange;
    cm->afn = allreduce;
    cm->user = user;
    int rc = comm_common_init(cm);
    if (rc) { comm_free(cm); return rc; }
    *out = reinterpret_cast<dxh_comm_t>(cm);
    return DXH_OK;
}

int dxh_comm_destroy(dxh_comm_t h) {
    Comm* cm = reinterpret_cast<Comm*>(h);
    if (!cm) return DXH_OK;
    if (cm->refs > 0) {
        cm->dead = true;  // contexts still hold it (their streams wait on ours): the last dxh_tvd_destroy frees it
        return DXH_OK;
    }
    comm_free(cm);
    return DXH_OK;
}

int dxh_comm_allreduce(dxh_comm_t h, double* vals, int n, int op) {
    Comm* cm = reinterpret_cast<Comm*>(h);
    DXH_REQUIRE(cm && vals, "null argument");
    DXH_HIP(hipSetDevice(cm->device));
    return comm_allreduce(cm, vals, n, op);
}

int dxh_comm_barrier(dxh_comm_t h) {
    double v = 0.0;
    return dxh_comm_allreduce(h, &v, 1, 0);
}

int dxh_comm_info(dxh_comm_t h, int* rank, int* world, int* kind, int64_t* exchanges, int64_t* bytes_sent) {
    Comm* cm = reinterpret_cast<Comm*>(h);
    DXH_REQUIRE(cm, "null argument");
    if (rank) *rank = cm->rank;
    if (world) *world = cm->world;
    if (kind) *kind = cm->kind;
    if (exchanges) *exchanges = cm->exchanges;
    if (bytes_sent) *bytes_sent = cm->bytes_sent;
    return DXH_OK;
}

}  // extern "C"
