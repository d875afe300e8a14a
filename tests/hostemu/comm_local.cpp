// Stand-in for csrc/comm_rccl.hip in the host emulation (TEST INFRASTRUCTURE): the transport interface of plonk_internal.hpp for a
// world of ONE rank — an all-to-all or all-gather with yourself is a copy.  Larger worlds are refused: collectives between
// processes are RCCL's job and are covered by tests/test_gpu_multirank.py on a GPU box and by the gloo tests on CPU.
#include <string.h>

#include "plonk_internal.hpp"

struct PlonkComm { int rank, world; };

int comm_unique_id(void* out128) {
    memset(out128, 0, 128);
    memcpy(out128, "hostemu", 8);
    return PLONK_OK;
}
int comm_create(PlonkComm** out, const void* id128, int rank, int world, int device) {
    (void)id128; (void)device;
    if (world != 1 || rank != 0) return plonk_fail(PLONK_ERR_ARG, "host emulation: only a communicator of world size 1 exists (got rank %d of %d)", rank, world);
    *out = new PlonkComm{0, 1};
    return PLONK_OK;
}
void comm_destroy(PlonkComm* c) { delete c; }
int comm_rank(const PlonkComm* c) { return c->rank; }
int comm_world(const PlonkComm* c) { return c->world; }
int comm_rccl_version() { return 0; }
int comm_alltoall(PlonkComm*, const void* send, void* recv, size_t bytes_per_peer, hipStream_t) {
    if (bytes_per_peer) memmove(recv, send, bytes_per_peer);
    return PLONK_OK;
}
int comm_allgather(PlonkComm*, const void* send, void* recv, size_t bytes, hipStream_t) {
    if (bytes) memmove(recv, send, bytes);
    return PLONK_OK;
}
int comm_allgather_host(PlonkComm*, const void* in, size_t bytes, void* out, hipStream_t) {
    if (bytes) memmove(out, in, bytes);
    return PLONK_OK;
}
