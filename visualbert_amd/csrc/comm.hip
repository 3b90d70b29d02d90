// comm.hip -- gradient all-reduce over RCCL behind the C ABI (include/visualbert_hip.h: vb_comm_*, vb_allreduce_bucket).
//
// Replaces the communication half of nn.DataParallel in the reference (visualbert/models/model_wrapper.py:146 wraps the
// model; :75 takes loss.mean() over the replicas; visualbert/models/train.py:146 splits the batch): one process per GPU,
// every rank a full replica, the flat fp32 gradient arena averaged in place, bucket by bucket, on a side stream while
// backward is still producing the next bucket.  RCCL rings run over xGMI (point-to-point links): the caller keeps
// buckets large (whole encoder layers, 28 MB fp32) so a ring runs at link speed.
//
// librccl is bound with dlopen at the first use: the copy already mapped into the process wins (PyTorch ships one with
// the same soname), so the library never holds two RCCL instances, and a single-GPU user never loads RCCL at all.
// Host-only code: no kernels in this file.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

#ifndef VB_EMU
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

// the slice of the NCCL/RCCL API used here (rccl.h: same ABI as NCCL 2.x)
typedef struct { char internal[128]; } vbNcclUniqueId;
typedef void* vbNcclComm;
enum { vbNcclSuccess = 0 };
enum { vbNcclFloat32 = 7, vbNcclBfloat16 = 9 };          // ncclDataType_t
enum { vbNcclSum = 0, vbNcclAvg = 4 };                   // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(vbNcclUniqueId*) = nullptr;
    int (*CommInitRank)(vbNcclComm*, int, vbNcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, vbNcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(vbNcclComm) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {                        // a copy that is already mapped (same soname) first
            r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.handle) break;
        }
        for (int i = 0; !r.handle && i < 2; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    });
    return r;
}

struct VbComm {
    vbNcclComm comm;
    int rank, nranks;
};

}  // namespace

extern "C" int vb_comm_unique_id(void* host_id) {
    static_assert(sizeof(vbNcclUniqueId) == VB_COMM_ID_BYTES, "unique id size");
    if (!host_id) return VB_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return VB_ERR_UNSUPPORTED;
    vbNcclUniqueId id;
    if (r.GetUniqueId(&id) != vbNcclSuccess) return VB_ERR_LAUNCH;
    memcpy(host_id, &id, sizeof(id));
    return VB_OK;
}

extern "C" int vb_comm_init(const void* host_id, int rank, int nranks, void** comm) {
    if (!host_id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return VB_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return VB_ERR_UNSUPPORTED;
    vbNcclUniqueId id;
    memcpy(&id, host_id, sizeof(id));
    VbComm* c = new VbComm{nullptr, rank, nranks};
    if (r.CommInitRank(&c->comm, nranks, id, rank) != vbNcclSuccess) { delete c; return VB_ERR_LAUNCH; }
    *comm = c;
    return VB_OK;
}

extern "C" int vb_comm_nranks(void* comm) { return comm ? ((VbComm*)comm)->nranks : VB_ERR_ARG; }

extern "C" int vb_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* stream) {
    if (!comm || !buf || count < 0 || (dtype != VB_F32 && dtype != VB_BF16)) return VB_ERR_ARG;
    if (count == 0) return VB_OK;
    Rccl& r = rccl();
    if (!r.ok) return VB_ERR_UNSUPPORTED;
    VbComm* c = (VbComm*)comm;
    const int rc = r.AllReduce(buf, buf, (size_t)count, dtype == VB_F32 ? vbNcclFloat32 : vbNcclBfloat16,
                               average ? vbNcclAvg : vbNcclSum, c->comm, (hipStream_t)stream);
    return rc == vbNcclSuccess ? VB_OK : VB_ERR_LAUNCH;
}

extern "C" int vb_comm_destroy(void* comm) {
    if (!comm) return VB_ERR_ARG;
    VbComm* c = (VbComm*)comm;
    Rccl& r = rccl();
    int rc = VB_OK;
    if (r.ok && c->comm && r.CommDestroy(c->comm) != vbNcclSuccess) rc = VB_ERR_LAUNCH;
    delete c;
    return rc;
}

#else  // VB_EMU: the kernel-logic simulator has no devices to connect

extern "C" int vb_comm_unique_id(void*) { return VB_ERR_UNSUPPORTED; }
extern "C" int vb_comm_init(const void*, int, int, void**) { return VB_ERR_UNSUPPORTED; }
extern "C" int vb_comm_nranks(void*) { return VB_ERR_UNSUPPORTED; }
extern "C" int vb_allreduce_bucket(void*, void*, int64_t, int, int, void*) { return VB_ERR_UNSUPPORTED; }
extern "C" int vb_comm_destroy(void*) { return VB_ERR_UNSUPPORTED; }

#endif
