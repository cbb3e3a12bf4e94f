// Internal interface of the multi-GPU transport (dxh_comm.hip): neighbour send/recv of device row blocks
// and tiny host-value all-reduces, over RCCL (xGMI) or over a host-staged callback transport.
#pragma once

#include "dxh_common.h"

namespace dxh {

struct P2POp {
    int peer;       // rank
    void* dev;      // device pointer (contiguous block)
    size_t bytes;
    int is_send;    // 1: send dev -> peer, 0: receive from peer into dev
};

struct Comm {
    int kind = 0;   // 0: RCCL, 1: host-staged callbacks
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;  // the comm stream (owned)
    void* nccl = nullptr;          // ncclComm_t
    dxh_comm_exchange_fn xfn = nullptr;
    dxh_comm_allreduce_fn afn = nullptr;
    void* user = nullptr;
    double* d_red = nullptr;       // 16 doubles of device scratch for all-reduces
    char* h_stage = nullptr;       // pinned staging of the host transport
    size_t h_stage_bytes = 0;
    int refs = 0;                  // contexts attached (dxh_tvd_attach_comm); destruction waits for the last one
    bool dead = false;             // dxh_comm_destroy was called while contexts were still attached
    int64_t exchanges = 0;         // statistics
    int64_t bytes_sent = 0;
};

// Enqueue all ops on `stream` as ONE group (they progress concurrently).  RCCL: asynchronous; host transport: blocks the
// calling thread until the transfer is done (rehearsal / tests only).
int comm_exchange(Comm* cm, const P2POp* ops, int n, hipStream_t stream);
// Blocking all-reduce of n <= 16 host doubles over all ranks; op 0 = sum, 1 = max.
int comm_allreduce(Comm* cm, double* vals, int n, int op);
// A context lets go of its communicator; frees it if dxh_comm_destroy has been called meanwhile.
void comm_release(Comm* cm);

}  // namespace dxh
