// Native communicator (new layer N1): the PS traffic of one rank as direct RCCL calls on the caller's HIP stream.
//
// Replaces external X2 -- the TF1 gRPC `tf.train.Server` mesh on port 2222 that the reference's headless Services
// address (/root/reference/handlers/train_handler.py:31-35,84,99-100).  One process per GPU; a communicator is created
// from a 128-byte ncclUniqueId that rank 0 publishes through the job's rendezvous store (the reference's rendezvous of
// record is its Redis; here: the TCPStore on rank 0's Service address).
//
// Why not torch.distributed for the data path: every c10d collective allocates a Work object, records two events and
// feeds a watchdog thread that polls them -- measured on MI355X (scripts/comm_host_cost.py, profiles/r3_comm_host_cost.txt):
// 10.9 / 13.4 / 16.0 / 17.6 us of host time per reduce_scatter / all_gather / all_to_all / scalar all_reduce against
// 8.0 / 8.1 / 11.2 / 4.6 us here (a step issues 10-30 of them from autograd hooks, on a host that is already 65-75 %
// busy enqueueing kernels), and event queries from the watchdog are illegal while a hipGraph capture is open (the c10d
// path sleeps 0.5 s before capturing to let it drain and must capture in thread-local mode).  Here a collective is ONE
// call that enqueues RCCL's kernel on the stream it is given: stream-ordered, no host object, graph-capturable as is.
//
// PS mapping:
//   push  = dtf_rccl_reduce_scatter, or the DIRECT form dtf_rccl_all_to_all (slice j straight to rank j: all 7 xGMI
//           links of the MI355X mesh at once; the W-way sum is fused into the K1 / K2 shard-apply kernel)
//   pull  = dtf_rccl_all_gather
//   embedding ids / rows / row gradients = dtf_rccl_all_to_all (fixed capacity) / dtf_rccl_all_to_all_v (exact route)
//   "few PS" topology (detail.ps < workers) = dtf_rccl_reduce / dtf_rccl_broadcast
//   global step, grad norm = dtf_rccl_all_reduce
//
// C ABI (ctypes, parallel/rccl_native.py).  Every call returns 0 or a negative error; dtf_rccl_last_error() has the text.
// Built with:  hipcc -O2 -fPIC -shared rccl_comm.cpp -o libdtf_rccl.so -lrccl   (links the RCCL the process already
// loaded: torch's bundled librccl.so.1 when torch was imported first).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace {

std::mutex g_err_mu;
std::string g_err;

int fail(const char* what, const char* detail, int code) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = std::string(what) + ": " + (detail ? detail : "?");
    return code;
}

#define NCCL_TRY(call)                                                             \
    do {                                                                           \
        ncclResult_t r_ = (call);                                                  \
        if (r_ != ncclSuccess) return fail(#call, ncclGetErrorString(r_), -100 - (int)r_); \
    } while (0)

#define HIP_TRY(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) return fail(#call, hipGetErrorString(e_), -1000 - (int)e_); \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    uint64_t calls = 0;   // collectives issued (diagnostics / tests)
    uint64_t bytes = 0;   // payload bytes handed to RCCL by this rank
};

// dtype codes shared with ops/hip.py (DTF_F32 = 0, DTF_BF16 = 1) + integer types for ids / counters
bool to_nccl(int dt, ncclDataType_t* out, size_t* esize) {
    switch (dt) {
        case 0: *out = ncclFloat32; *esize = 4; return true;
        case 1: *out = ncclBfloat16; *esize = 2; return true;
        case 2: *out = ncclInt64; *esize = 8; return true;
        case 3: *out = ncclInt32; *esize = 4; return true;
        case 4: *out = ncclFloat64; *esize = 8; return true;
        case 5: *out = ncclUint8; *esize = 1; return true;
        default: return false;
    }
}

bool to_op(int op, ncclRedOp_t* out) {
    switch (op) {
        case 0: *out = ncclSum; return true;
        case 1: *out = ncclMax; return true;
        case 2: *out = ncclMin; return true;
        case 3: *out = ncclProd; return true;
        default: return false;
    }
}

}  // namespace

extern "C" {

const char* dtf_rccl_last_error() {
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

int dtf_rccl_abi_version() { return 1; }

int dtf_rccl_version(int* out) {
    NCCL_TRY(ncclGetVersion(out));
    return 0;
}

int dtf_rccl_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

// rank 0: create the id every rank needs for dtf_rccl_init (128 bytes, published through the job store)
int dtf_rccl_unique_id(void* out) {
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(out, &id, sizeof(id));
    return 0;
}

int dtf_rccl_init(void** handle, const void* id_bytes, int nranks, int rank, int device) {
    if (!handle || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return fail("dtf_rccl_init", "bad arguments", -2);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    Comm* c = new Comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank", ncclGetErrorString(r), -100 - (int)r);
    }
    *handle = c;
    return 0;
}

int dtf_rccl_info(void* h, int* rank, int* nranks, int* device, uint64_t* calls, uint64_t* bytes) {
    Comm* c = (Comm*)h;
    if (!c) return fail("dtf_rccl_info", "null communicator", -2);
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (device) *device = c->device;
    if (calls) *calls = c->calls;
    if (bytes) *bytes = c->bytes;
    return 0;
}

// Orderly teardown: every rank has finished its collectives (callers barrier first).
int dtf_rccl_destroy(void* h) {
    Comm* c = (Comm*)h;
    if (!c) return 0;
    ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return fail("ncclCommDestroy", ncclGetErrorString(r), -100 - (int)r);
    return 0;
}

// Error-path teardown: peers may be gone and outstanding collectives will never finish.
int dtf_rccl_abort(void* h) {
    Comm* c = (Comm*)h;
    if (!c) return 0;
    ncclResult_t r = ncclCommAbort(c->comm);
    delete c;
    if (r != ncclSuccess) return fail("ncclCommAbort", ncclGetErrorString(r), -100 - (int)r);
    return 0;
}

// Asynchronous error state of the communicator (a dead peer shows up here, not as a hang): 0 = healthy.
int dtf_rccl_async_error(void* h) {
    Comm* c = (Comm*)h;
    if (!c) return fail("dtf_rccl_async_error", "null communicator", -2);
    ncclResult_t st = ncclSuccess;
    NCCL_TRY(ncclCommGetAsyncError(c->comm, &st));
    if (st != ncclSuccess) return fail("communicator", ncclGetErrorString(st), -100 - (int)st);
    return 0;
}

#define COMM_ARGS(h, dt)                                                  \
    Comm* c = (Comm*)(h);                                                 \
    if (!c) return fail(__func__, "null communicator", -2);               \
    ncclDataType_t nt; size_t es;                                         \
    if (!to_nccl((dt), &nt, &es)) return fail(__func__, "dtype code", -3); \
    hipStream_t st = (hipStream_t)stream;

// push: recv[count] = sum over ranks of send[rank * count .. (rank + 1) * count)
int dtf_rccl_reduce_scatter(void* h, const void* send, void* recv, int64_t count, int dt, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclReduceScatter(send, recv, (size_t)count, nt, ncclSum, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * c->nranks * es;
    return 0;
}

// pull: recv[r * count ..] = rank r's send[count]
int dtf_rccl_all_gather(void* h, const void* send, void* recv, int64_t count, int dt, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclAllGather(send, recv, (size_t)count, nt, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * es;
    return 0;
}

// direct push / fixed-capacity embedding exchange: recv[r * count ..] = rank r's send[me * count ..]
int dtf_rccl_all_to_all(void* h, const void* send, void* recv, int64_t count, int dt, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclAllToAll(send, recv, (size_t)count, nt, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * c->nranks * es;
    return 0;
}

// exact embedding route: element counts / displacements per peer (host arrays of nranks entries)
int dtf_rccl_all_to_all_v(void* h, const void* send, const int64_t* send_counts, const int64_t* send_displs, void* recv,
                          const int64_t* recv_counts, const int64_t* recv_displs, int dt, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclGroupStart());
    uint64_t total = 0;
    for (int p = 0; p < c->nranks; ++p) {
        if (send_counts[p] > 0) {
            ncclResult_t r = ncclSend((const char*)send + (size_t)send_displs[p] * es, (size_t)send_counts[p], nt, p, c->comm, st);
            if (r != ncclSuccess) { ncclGroupEnd(); return fail("ncclSend", ncclGetErrorString(r), -100 - (int)r); }
            total += (uint64_t)send_counts[p] * es;
        }
        if (recv_counts[p] > 0) {
            ncclResult_t r = ncclRecv((char*)recv + (size_t)recv_displs[p] * es, (size_t)recv_counts[p], nt, p, c->comm, st);
            if (r != ncclSuccess) { ncclGroupEnd(); return fail("ncclRecv", ncclGetErrorString(r), -100 - (int)r); }
        }
    }
    NCCL_TRY(ncclGroupEnd());
    c->calls++; c->bytes += total;
    return 0;
}

int dtf_rccl_all_reduce(void* h, const void* send, void* recv, int64_t count, int dt, int op, void* stream) {
    COMM_ARGS(h, dt);
    ncclRedOp_t ro;
    if (!to_op(op, &ro)) return fail(__func__, "reduce op code", -3);
    NCCL_TRY(ncclAllReduce(send, recv, (size_t)count, nt, ro, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * es;
    return 0;
}

int dtf_rccl_broadcast(void* h, const void* send, void* recv, int64_t count, int dt, int root, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclBroadcast(send, recv, (size_t)count, nt, root, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * es;
    return 0;
}

int dtf_rccl_reduce(void* h, const void* send, void* recv, int64_t count, int dt, int root, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclReduce(send, recv, (size_t)count, nt, ncclSum, root, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * es;
    return 0;
}

int dtf_rccl_send(void* h, const void* buf, int64_t count, int dt, int peer, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclSend(buf, (size_t)count, nt, peer, c->comm, st));
    c->calls++; c->bytes += (uint64_t)count * es;
    return 0;
}

int dtf_rccl_recv(void* h, void* buf, int64_t count, int dt, int peer, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclRecv(buf, (size_t)count, nt, peer, c->comm, st));
    c->calls++;
    return 0;
}

// One bucket of the non-clipped sync step issued as ONE native call: push (reduce-scatter of the packed gradient
// bucket) and, after the caller's shard-apply kernel, pull.  Kept as two entry points because K1 / K2 sit between
// them on the same stream; grouping them here would not remove a launch.  What IS grouped: the pull of bucket b and
// the push of bucket b + 1 when both are ready (ncclGroup = one RCCL kernel, both directions of every link busy).
int dtf_rccl_pull_and_push(void* h, const void* pull_send, void* pull_recv, int64_t pull_count, const void* push_send,
                           void* push_recv, int64_t push_count, int dt, void* stream) {
    COMM_ARGS(h, dt);
    NCCL_TRY(ncclGroupStart());
    ncclResult_t r1 = ncclAllGather(pull_send, pull_recv, (size_t)pull_count, nt, c->comm, st);
    ncclResult_t r2 = ncclReduceScatter(push_send, push_recv, (size_t)push_count, nt, ncclSum, c->comm, st);
    ncclResult_t r3 = ncclGroupEnd();
    if (r1 != ncclSuccess) return fail("ncclAllGather", ncclGetErrorString(r1), -100 - (int)r1);
    if (r2 != ncclSuccess) return fail("ncclReduceScatter", ncclGetErrorString(r2), -100 - (int)r2);
    if (r3 != ncclSuccess) return fail("ncclGroupEnd", ncclGetErrorString(r3), -100 - (int)r3);
    c->calls++; c->bytes += (uint64_t)(pull_count + push_count * c->nranks) * es;
    return 0;
}

}  // extern "C"
