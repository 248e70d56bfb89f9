// Slab communicator of the 3-D case: the pencil transpose (all-to-all) and the scalar reductions INSIDE the library
// (the reference gets them from Dedalus' in-library MPI transposes, FWD_Solve_KDyn.py:118-134, README.md:83).
//
// Two transports behind one interface:
//   * RCCL: one communicator per context (ncclCommInitRank, one process per GPU), the transpose = ncclGroupStart + ncclSend/ncclRecv per
//     peer on a HIP stream of the solver — xGMI is point to point, so every GPU talks to its 7 peers at once;
//   * caller-provided functions (smo_comm_set_transport): for tests where several ranks share one GPU (RCCL refuses that) or run
//     over gloo — the same time loop, another wire.
// librccl is loaded with dlopen at the first use: libsmo.so itself has no RCCL dependency, and in a process that already holds a
// copy (PyTorch bundles one) that copy is the one used.
#pragma once
#include "smo_common.hpp"

namespace smo {

class SlabComm {
public:
    int rank = 0, world = 1;
    ~SlabComm();
    bool ready() const { return nccl_ != nullptr || a2a_ != nullptr; }
    bool is_rccl() const { return nccl_ != nullptr; }
    int init_rccl(int rank, int world, const void* unique_id);
    int set_transport(int rank, int world, smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user);
    // every rank sends `bytes_per_peer` bytes at src + p*bytes_per_peer to rank p and receives rank p's block at dst + p*bytes_per_peer
    int alltoall(const void* src, void* dst, size_t bytes_per_peer, hipStream_t s);
    // sum over the ranks of n doubles (host values in and out); synchronises `s`.  `dev_scratch`: >= n doubles of device memory
    int allreduce_sum(double* vals, int n, hipStream_t s, double* dev_scratch);
    static int unique_id(void* out128);
    // drop the transport (ncclCommDestroy / forget the caller's functions): ready() is false again and a new init may follow
    void reset();
    // path of the librccl this library bound (dladdr of ncclGetUniqueId; loads it if nobody has yet), "" if none could be loaded
    static const char* library_path();

private:
    void* nccl_ = nullptr;           // ncclComm_t
    smo_alltoall_fn a2a_ = nullptr;
    smo_allreduce_fn ared_ = nullptr;
    void* user_ = nullptr;
};

}  // namespace smo
