// Data-parallel gradient exchange over RCCL / xGMI.
//
// New functionality: the reference trains on one GPU (KITTI/trainer.py:45, NYUv2/train.py:234).  One process per
// GPU; every rank calls wmd_comm_allreduce on flat fp32 gradient buckets, on a side HIP stream chosen by the
// caller so that the decoder bucket (ready first in backward) is reduced while the encoder backward still runs
// (wavelet_monodepth_amd/ddp.py).  In-place sum, then scaling by 1/world inside the same stream.
#include <rccl/rccl.h>
#include <string.h>
#include <algorithm>
#include "wmd_internal.h"

struct wmd_comm {
    ncclComm_t comm;
    int world, rank;
};

namespace wmd {
__global__ void scale_kernel(float* __restrict__ x, size_t n, float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}
static int nccl_fail(const char* what, ncclResult_t r) { return fail(WMD_ERR_COMM, "%s: %s", what, ncclGetErrorString(r)); }
}  // namespace wmd

using namespace wmd;

extern "C" int wmd_comm_unique_id(void* unique_id_128) {
    if (!unique_id_128) return fail(WMD_ERR_BAD_ARG, "wmd_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(unique_id_128, &id, sizeof(id));
    return WMD_OK;
}

extern "C" int wmd_comm_init(wmd_comm** comm, const void* unique_id_128, int world, int rank) {
    if (!comm || !unique_id_128) return fail(WMD_ERR_BAD_ARG, "wmd_comm_init: null pointer");
    if (world < 1 || rank < 0 || rank >= world) return fail(WMD_ERR_BAD_ARG, "wmd_comm_init: world=%d rank=%d", world, rank);
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof(id));
    wmd_comm* c = new wmd_comm{nullptr, world, rank};
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail("ncclCommInitRank", r);
    }
    *comm = c;
    return WMD_OK;
}

extern "C" int wmd_comm_allreduce(wmd_comm* comm, float* buf, size_t n, float scale, void* stream) {
    if (!comm || !buf) return fail(WMD_ERR_BAD_ARG, "wmd_comm_allreduce: null pointer");
    if (n == 0) return WMD_OK;
    hipStream_t s = (hipStream_t)stream;
    ncclResult_t r = ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm->comm, s);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    if (scale != 1.f) {
        const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 8);
        hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, s, buf, n, scale);
        return check_launch("scale_kernel");
    }
    return WMD_OK;
}

extern "C" int wmd_comm_broadcast(wmd_comm* comm, float* buf, size_t n, int root, void* stream) {
    if (!comm || !buf) return fail(WMD_ERR_BAD_ARG, "wmd_comm_broadcast: null pointer");
    if (root < 0 || root >= comm->world) return fail(WMD_ERR_BAD_ARG, "wmd_comm_broadcast: root=%d world=%d", root, comm->world);
    if (n == 0) return WMD_OK;
    ncclResult_t r = ncclBroadcast(buf, buf, n, ncclFloat, root, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail("ncclBroadcast", r);
    return WMD_OK;
}

extern "C" int wmd_comm_info(wmd_comm* comm, int* rccl_version, int* world, int* rank) {
    if (!comm) return fail(WMD_ERR_BAD_ARG, "wmd_comm_info: null communicator");
    int v = 0, w = 0, r = 0;
    ncclResult_t e = ncclGetVersion(&v);
    if (e == ncclSuccess) e = ncclCommCount(comm->comm, &w);
    if (e == ncclSuccess) e = ncclCommUserRank(comm->comm, &r);
    if (e != ncclSuccess) return nccl_fail("wmd_comm_info", e);
    if (rccl_version) *rccl_version = v;
    if (world) *world = w;
    if (rank) *rank = r;
    return WMD_OK;
}

extern "C" int wmd_comm_destroy(wmd_comm* comm) {
    if (!comm) return WMD_OK;
    ncclResult_t r = ncclCommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return nccl_fail("ncclCommDestroy", r);
    return WMD_OK;
}
