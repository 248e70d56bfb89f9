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

#include <atomic>
#include <condition_variable>
#include <mutex>

namespace smo {

// Third transport: the ranks are contexts of ONE process, each driven by its own host thread on its own GPU (smo_create_multi, csrc/multi.cpp).
// A transpose is then every rank pulling its W blocks over the node's point-to-point links straight out of the peers' send buffers (peer
// access), on the rank's own stream, ordered by HIP events — no RCCL, no second process:
//   rank r:  publish (src, dst), record ev_ready[r] on its stream           (everything enqueued so far: src produced, dst consumed)
//            -- host barrier: every rank has recorded --
//            wait ev_ready[p] of every peer; ONE gather kernel reads my block out of every peer's send buffer (src_p + r * bytes, over
//            xGMI, all links at once) into dst + p * bytes        [SMO_PEER_COPY=memcpy / no peer access: W hipMemcpyPeerAsync calls]
//            record ev_pulled[r]
//            -- host barrier --
//            wait ev_pulled[p] of every peer: from here on my src may be overwritten
// The host barriers only order the *calls* (an event must have been recorded before another thread may wait for it); the data dependencies
// stay on the GPUs.  They are generation-counter SPIN barriers (round 4): a step pair has 8 exchanges = 16 rendezvous of W threads, and a
// mutex / condition-variable rendezvous costs a futex wake per thread (tens of microseconds with 8 workers) where the kernels of a 256^3 / 8
// step pair take 0.56 ms; the spin form costs a cache-line bounce (it yields the core after a few thousand polls, so oversubscribed hosts
// still make progress).  A rank that fails raises `failed` and releases the others, so a collective never hangs the process.
// The events carry hipEventReleaseToSystem: the producer's writes must be visible to a KERNEL running on another device (the gather kernel
// reads the peers' buffers directly), not only to a copy engine.  Until a node with >= 2 GPUs has run tests/test_kdyn_multi_gpu.py the
// gather kernel is the default only between ranks that share a device; distinct devices default to the hipMemcpyPeerAsync calls
// (SMO_PEER_COPY=kernel / memcpy force one; use_kernel / smo_comm_get key 3 report the choice).
class PeerGroup {
public:
    explicit PeerGroup(const std::vector<int>& devices);
    ~PeerGroup();
    int world() const { return (int)dev.size(); }
    // chained = true: the caller guarantees that the NEXT writer of `src` on this rank is the pull of a later exchange on the same stream
    // (the time loop: transpose -> grid kernels -> transpose back -> per-mode update -> transpose ...).  That pull waits for every peer's
    // ev_ready of its own exchange, which the peer records after its pull of THIS one — so "everybody has pulled my block" is already implied
    // when it runs, and the second rendezvous, the ev_pulled record and the W - 1 waits on the peers' ev_pulled are left out: one rendezvous,
    // one record, W - 1 waits and one launch per exchange instead of two, two, 2 (W - 1) and one.
    int alltoall(int rank, const void* src, void* dst, size_t bytes_per_peer, hipStream_t s, bool chained = false);
    int allreduce_sum(int rank, double* vals, int n, hipStream_t s);
    void abort();                       // a rank failed outside a collective: release everybody waiting in one
    void reset();                       // before a new collective call sequence (all ranks idle)
    std::vector<int> dev;
    bool peer_access = false;           // hipDeviceEnablePeerAccess succeeded for every pair of distinct devices
    bool use_kernel = false;            // pulls as one gather kernel reading the peers' buffers directly (else hipMemcpyPeerAsync calls)
    bool distinct = false;              // at least two ranks sit on different devices (bytes really cross the links)
    unsigned long long barriers = 0;    // rendezvous passed by rank 0 (diagnostics: smo_comm_get key 4)
    std::vector<double> wait_ms;        // per rank: host time spent waiting in rendezvous so far

private:
    int barrier(int rank);              // SMO_OK, or SMO_ERR_STATE when a rank has failed
    alignas(64) std::atomic<int> waiting{0};
    alignas(64) std::atomic<unsigned long> generation{0};
    alignas(64) std::atomic<bool> failed{false};
    // published per exchange in one of TWO slots (exchange count of the rank, parity): after a chained exchange a rank may already publish
    // its next exchange while a slower peer still reads this one; it cannot get further ahead than that (the next rendezvous holds it)
    std::vector<const void*> pub_src;   // [2][W]
    std::vector<void*> pub_dst;         // [2][W]
    std::vector<unsigned long> seq;     // [W * 8]: exchanges started by each rank (one cache line apart)
    std::vector<hipEvent_t> ev_ready, ev_pulled;
    std::vector<double> red;            // [rank][64]
};

class SlabComm {
public:
    int rank = 0, world = 1;
    ~SlabComm();
    bool ready() const { return nccl_ != nullptr || a2a_ != nullptr || peers_ != nullptr || null_; }
    int set_peers(int rank, PeerGroup* g);
    bool is_rccl() const { return nccl_ != nullptr; }
    double wait_ms() const { return peers_ ? peers_->wait_ms[(size_t)rank * 8] : 0.0; }      // host time this rank has waited for its peers (multi-device contexts)
    int init_rccl(int rank, int world, const void* unique_id);
    int set_transport(int rank, int world, smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user);
    // every rank sends `bytes_per_peer` bytes at src + p*bytes_per_peer to rank p and receives rank p's block at dst + p*bytes_per_peer
    // (chained: see PeerGroup::alltoall; the other transports ignore it)
    int alltoall(const void* src, void* dst, size_t bytes_per_peer, hipStream_t s, bool chained = false);
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
    PeerGroup* peers_ = nullptr;     // not owned
    bool null_ = false;              // the null transport (set_transport with three null arguments): exchanges and reductions do nothing
};

}  // namespace smo
