// C1 -- the one exchange step of the sharded path: sum of the frame-summed accumulators of update_spatial
// (A1 = Y_i C^T and Cs = C C^T, Demix/dNMF.py:153-154) over the ranks that each hold a block of frames.
//
// RCCL is bound at run time.  The process already holds one (torch loads its bundled librccl for
// torch.distributed's "nccl" backend) and a second copy linked against another HIP runtime must not come in,
// so the handle is looked up among the loaded objects first and only then by name.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "common.hpp"

namespace dnmf {

struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
};

static const Rccl &rccl() {
    static const Rccl api = [] {
        Rccl a;
        void *h = nullptr;
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
        a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.get_unique_id && a.comm_init_rank && a.all_reduce && a.comm_destroy && a.error_string;
        return a;
    }();
    return api;
}

static int rccl_status(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return DNMF_OK;
    return fail((int)r, "%s: RCCL error %d (%s)", what, (int)r, rccl().error_string(r));
}

}  // namespace dnmf

extern "C" {

static_assert(DNMF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

int dnmf_comm_unique_id(void *id_host) {
    using namespace dnmf;
    DNMF_REQUIRE(id_host, DNMF_E_NULL, "dnmf_comm_unique_id: NULL id");
    DNMF_REQUIRE(rccl().ok, DNMF_E_UNSUPPORTED, "dnmf_comm_unique_id: no librccl in this process or on the loader path");
    ncclUniqueId id;
    const int rc = rccl_status(rccl().get_unique_id(&id), "dnmf_comm_unique_id");
    if (rc == DNMF_OK) std::memcpy(id_host, &id, sizeof(id));
    return rc;
}

int dnmf_comm_init(dnmf_comm_t *comm, const void *id_host, int nranks, int rank) {
    using namespace dnmf;
    DNMF_REQUIRE(comm && id_host, DNMF_E_NULL, "dnmf_comm_init: NULL argument");
    DNMF_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, DNMF_E_SHAPE, "dnmf_comm_init: rank %d of %d", rank, nranks);
    DNMF_REQUIRE(rccl().ok, DNMF_E_UNSUPPORTED, "dnmf_comm_init: no librccl in this process or on the loader path");
    ncclUniqueId id;
    std::memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = rccl_status(rccl().comm_init_rank(&c, nranks, id, rank), "dnmf_comm_init");
    *comm = rc == DNMF_OK ? static_cast<dnmf_comm_t>(c) : nullptr;
    return rc;
}

int dnmf_allreduce_sum_f32(dnmf_comm_t comm, float *buf, size_t count, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(comm && buf, DNMF_E_NULL, "dnmf_allreduce_sum_f32: NULL argument");
    DNMF_REQUIRE(rccl().ok, DNMF_E_UNSUPPORTED, "dnmf_allreduce_sum_f32: no librccl");
    if (count == 0) return DNMF_OK;
    return rccl_status(rccl().all_reduce(buf, buf, count, ncclFloat, ncclSum, static_cast<ncclComm_t>(comm),
                                         static_cast<hipStream_t>(stream)),
                       "dnmf_allreduce_sum_f32");
}

int dnmf_comm_destroy(dnmf_comm_t comm) {
    using namespace dnmf;
    if (!comm) return DNMF_OK;
    DNMF_REQUIRE(rccl().ok, DNMF_E_UNSUPPORTED, "dnmf_comm_destroy: no librccl");
    return rccl_status(rccl().comm_destroy(static_cast<ncclComm_t>(comm)), "dnmf_comm_destroy");
}

}  // extern "C"
