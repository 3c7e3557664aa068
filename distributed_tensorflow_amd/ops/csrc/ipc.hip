// Cross-process device memory for the one-sided asynchronous PS (parallel/async_ps.py): every rank
// allocates the shard it hosts with hipMalloc, exports an IPC handle, and maps every peer's shard
// into its own address space (dmabuf IPC; over xGMI when the peer is another GPU).  A worker then
// runs the K1/K2 shard-apply kernels on ITS device against the OWNER's memory: no collective, no
// rendezvous, no barrier between workers -- TF1's asynchronous parameter-server update.
#include "dtf_common.h"
#include <string.h>

extern "C" {

int dtf_ipc_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

int dtf_ipc_alloc(void** ptr, int64_t bytes) {
    DTF_CLEAR_ERRORS();
    if (bytes <= 0) return -51;
    hipError_t e = hipMalloc(ptr, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ptr, 0, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}

int dtf_ipc_free(void* ptr) {
    DTF_CLEAR_ERRORS();
    return (int)hipFree(ptr);
}

int dtf_ipc_get_handle(void* ptr, void* handle_out) {
    DTF_CLEAR_ERRORS();
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int dtf_ipc_open(const void* handle_in, void** ptr) {
    DTF_CLEAR_ERRORS();
    hipIpcMemHandle_t h;
    memcpy(&h, handle_in, sizeof(h));
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

int dtf_ipc_close(void* ptr) {
    DTF_CLEAR_ERRORS();
    return (int)hipIpcCloseMemHandle(ptr);
}

}  // extern "C"
