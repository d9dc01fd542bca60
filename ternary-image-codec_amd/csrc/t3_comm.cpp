// t3_comm.cpp — the one exchange step of the multi-GPU path: an RCCL all-gather of the fixed-size per-frame
// index records (t3_frame_record, 96 B) from which every rank assembles the T3V-style frame index
// (io_t3p_t3v.cpp:252-289).  Frames themselves never cross GPUs.
//
// RCCL is bound at run time (dlopen of librccl.so.1): libt3hip.so has no link-time dependency on it, loads on a box
// without RCCL, and shares the copy a host process has already mapped (PyTorch-ROCm ships one under the same SONAME).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include <rccl/rccl.h>

#include "../../include/t3hip.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl R;
std::mutex mu;
thread_local std::string last_err;     // per thread: one context per thread may fail or read at the same time

int fail(const std::string& what) { last_err = what; return T3_E_COMM; }

int bind() {
    std::lock_guard<std::mutex> lk(mu);
    if (R.so) return T3_OK;
    const char* off = getenv("T3HIP_COMM_DISABLE");                  // tests: behave as on a box without RCCL
    if (off && off[0] == '1') return fail("RCCL disabled by T3HIP_COMM_DISABLE");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    for (const char* n : names) { so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
    if (!so) return fail(std::string("RCCL not found: ") + dlerror());
    auto sym = [&](const char* n) { return dlsym(so, n); };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.GetErrorString) { dlclose(so); return fail("RCCL symbols missing"); }
    R.so = so;
    return T3_OK;
}

int nccl_fail(ncclResult_t r, const char* what) { return fail(std::string(what) + ": " + R.GetErrorString(r)); }

}  // namespace

struct t3_comm { ncclComm_t comm; int world, rank, device; };

extern "C" {

const char* t3hip_comm_last_error(void) { return last_err.c_str(); }

int t3hip_comm_available(void) { return bind(); }

int t3hip_comm_unique_id(uint8_t id[T3_COMM_ID_BYTES]) {
    static_assert(T3_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return T3_E_ARG;
    if (int rc = bind()) return rc;
    ncclUniqueId u; const ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    memcpy(id, u.internal, T3_COMM_ID_BYTES);
    return T3_OK;
}

int t3hip_comm_create(const uint8_t id[T3_COMM_ID_BYTES], int world, int rank, t3_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return T3_E_ARG;
    if (!t3hip_is_ready()) return T3_E_NODEVICE;                    // the communicator is bound to the device of t3hip_init
    if (int rc = bind()) return rc;
    ncclUniqueId u; memcpy(u.internal, id, T3_COMM_ID_BYTES);
    t3_comm* c = new t3_comm{nullptr, world, rank, -1};
    (void)hipGetDevice(&c->device);
    const ncclResult_t r = R.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
    *out = c;
    return T3_OK;
}

int t3hip_comm_destroy(t3_comm* c) {
    if (!c) return T3_OK;
    const ncclResult_t r = R.CommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? T3_OK : nccl_fail(r, "ncclCommDestroy");
}

int t3hip_comm_world(const t3_comm* c) { return c ? c->world : 0; }
int t3hip_comm_rank(const t3_comm* c) { return c ? c->rank : -1; }

int t3hip_index_allgather(t3_comm* c, const t3_frame_record* d_local, uint64_t n_local, t3_frame_record* d_all, void* stream) {
    if (!c || (n_local && (!d_local || !d_all))) return T3_E_ARG;
    if (!n_local) return T3_OK;
    if (!R.AllGather) return fail("t3hip_index_allgather: RCCL not bound");       // (a t3_comm cannot exist without it)
    const ncclResult_t r = R.AllGather(d_local, d_all, (size_t)n_local * sizeof(t3_frame_record), ncclUint8, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? T3_OK : nccl_fail(r, "ncclAllGather");
}

}  // extern "C"
