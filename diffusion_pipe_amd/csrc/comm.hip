// comm.hip -- stage-to-stage point-to-point over RCCL (xGMI links between neighbouring MI355X): the C-ABI form of the activation /
// gradient exchange the reference's schedule issues per micro-batch (utils/patches.py:126-160 SendActivation / RecvActivation / SendGrad /
// RecvGrad -> DeepSpeed p2p.send / p2p.recv -> torch.distributed -> NCCL).  One 2-rank communicator per neighbour pair; a tuple crossing a
// stage boundary goes out as ONE grouped RCCL operation (dpipe_group_start .. dpipe_group_end) on the caller's communication stream, and
// receives land directly in caller-owned buffers (the stage graph's static inputs).  Bytes are moved untyped (ncclInt8): bound = the
// xGMI link (~153 GB/s per direction per neighbour), algorithmic bytes = the payload.
// RCCL is resolved at run time (dlopen / dlsym) so that libdpipe_hip.so carries no link-time dependency on it: PyTorch-ROCm brings its own
// librccl.so and a process must hold ONE copy (RTLD_DEFAULT finds that one first).
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

using namespace dpipe;

namespace {

struct Rccl {
    int (*GetUniqueId)(void*);
    int (*CommInitRank)(void**, int, char[128] /* ncclUniqueId by value */, int);
    int (*CommDestroy)(void*);
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t);
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
    bool ok;
};

struct UniqueId { char internal[128]; };       // layout of ncclUniqueId (rccl.h NCCL_UNIQUE_ID_BYTES)
typedef int (*init_rank_fn)(void**, int, UniqueId, int);

Rccl* rccl() {
    static Rccl r{};
    static bool tried = false;
    if (tried) return r.ok ? &r : nullptr;
    tried = true;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclSend")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { set_last_error("dpipe_comm: librccl.so not found"); return nullptr; }
    }
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) { set_last_error("dpipe_comm: RCCL symbol " name " missing"); return nullptr; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    r.ok = true;
    return &r;
}

int fail(Rccl* r, const char* what, int rc) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", what, r->GetErrorString(rc));
    set_last_error(msg);
    return 1000 + rc;
}

}  // namespace

extern "C" {

int dpipe_comm_unique_id(void* id128) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    if (!id128) { set_last_error("dpipe_comm_unique_id: null"); return DPIPE_ERR_ARG; }
    const int rc = r->GetUniqueId(id128);
    return rc ? fail(r, "ncclGetUniqueId", rc) : DPIPE_OK;
}

int dpipe_comm_init(void** comm, int world, int rank, const void* id128) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) { set_last_error("dpipe_comm_init: bad argument"); return DPIPE_ERR_ARG; }
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    const int rc = reinterpret_cast<init_rank_fn>(r->CommInitRank)(comm, world, id, rank);
    return rc ? fail(r, "ncclCommInitRank", rc) : DPIPE_OK;
}

int dpipe_comm_destroy(void* comm) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    if (!comm) return DPIPE_OK;
    const int rc = r->CommDestroy(comm);
    return rc ? fail(r, "ncclCommDestroy", rc) : DPIPE_OK;
}

int dpipe_group_start(void) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    const int rc = r->GroupStart();
    return rc ? fail(r, "ncclGroupStart", rc) : DPIPE_OK;
}

int dpipe_group_end(void) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    const int rc = r->GroupEnd();
    return rc ? fail(r, "ncclGroupEnd", rc) : DPIPE_OK;
}

int dpipe_send(void* comm, const void* buf, long nbytes, int peer, void* stream) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    if (!comm || (!buf && nbytes > 0) || nbytes < 0 || peer < 0) { set_last_error("dpipe_send: bad argument"); return DPIPE_ERR_ARG; }
    const int rc = r->Send(buf, (size_t)nbytes, 0 /* ncclInt8 */, peer, comm, reinterpret_cast<hipStream_t>(stream));
    return rc ? fail(r, "ncclSend", rc) : DPIPE_OK;
}

int dpipe_recv(void* comm, void* buf, long nbytes, int peer, void* stream) {
    Rccl* r = rccl();
    if (!r) return DPIPE_ERR_UNSUPPORTED;
    if (!comm || (!buf && nbytes > 0) || nbytes < 0 || peer < 0) { set_last_error("dpipe_recv: bad argument"); return DPIPE_ERR_ARG; }
    const int rc = r->Recv(buf, (size_t)nbytes, 0 /* ncclInt8 */, peer, comm, reinterpret_cast<hipStream_t>(stream));
    return rc ? fail(r, "ncclRecv", rc) : DPIPE_OK;
}

}  // extern "C"
