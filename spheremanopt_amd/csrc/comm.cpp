// SlabComm: RCCL (dlopen'ed) or caller-provided transport.  See comm.hpp.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace smo {
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
std::string g_load_error;

std::string g_lib_path;       // file the bound ncclGetUniqueId lives in (dladdr): which of the process' RCCL copies this library talks to

void load_rccl() {
    // a copy already in the process (PyTorch's) wins: RTLD_NOLOAD first, then the system library.  SMO_RCCL_LIB (a path or soname)
    // replaces the candidate list: the way to pin one copy when a process holds several, and how the tests take librccl away.
    std::vector<std::string> names = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    if (const char* e = getenv("SMO_RCCL_LIB")) names.assign(1, std::string(e));
    void* h = nullptr;
    std::string why;
    for (const std::string& n : names) { h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (h) break; }
    (void)dlerror();                                       // RTLD_NOLOAD misses are not errors worth reporting: clear the state
    if (!h) for (const std::string& n : names) {
        h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char* e = dlerror();                         // ONE call: dlerror() clears the message it returns
        why += (why.empty() ? "" : "; ") + (e ? std::string(e) : n + ": not found");
    }
    if (!h) { g_load_error = "cannot load librccl: " + (why.empty() ? std::string("not found") : why); return; }
    g_rccl.handle = h;
    bool ok = true;
#define SMO_SYM(field, name)                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));          \
    if (!g_rccl.field) { ok = false; g_load_error = std::string("librccl lacks ") + name; }
    SMO_SYM(GetUniqueId, "ncclGetUniqueId")
    SMO_SYM(CommInitRank, "ncclCommInitRank")
    SMO_SYM(CommDestroy, "ncclCommDestroy")
    SMO_SYM(GroupStart, "ncclGroupStart")
    SMO_SYM(GroupEnd, "ncclGroupEnd")
    SMO_SYM(Send, "ncclSend")
    SMO_SYM(Recv, "ncclRecv")
    SMO_SYM(AllReduce, "ncclAllReduce")
    SMO_SYM(GetErrorString, "ncclGetErrorString")
#undef SMO_SYM
    if (!ok) { g_rccl.handle = nullptr; return; }
    Dl_info info{};
    if (dladdr(reinterpret_cast<void*>(g_rccl.GetUniqueId), &info) && info.dli_fname) g_lib_path = info.dli_fname;
}

int need_rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.handle) { set_error("RCCL unavailable: %s", g_load_error.c_str()); return SMO_ERR_UNSUPPORTED; }
    return SMO_OK;
}

#define SMO_NCCL(call)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);       \
            return SMO_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

}  // namespace

int SlabComm::unique_id(void* out128) {
    SMO_TRY(need_rccl());
    ncclUniqueId id;
    SMO_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof(id));
    return SMO_OK;
}

int SlabComm::init_rccl(int r, int w, const void* unique_id) {
    SMO_TRY(need_rccl());
    if (ready()) { set_error("smo_comm_init: the context already has a communicator"); return SMO_ERR_STATE; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    SMO_NCCL(g_rccl.CommInitRank(&c, w, id, r));
    nccl_ = c; rank = r; world = w;
    return SMO_OK;
}

int SlabComm::set_transport(int r, int w, smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user) {
    if (!a2a && !ared && !user) {                          // the null transport: nothing moves (profiling one rank's share on one GPU)
        if (ready()) { set_error("smo_comm_set_transport: the context already has a communicator"); return SMO_ERR_STATE; }
        null_ = true; rank = r; world = w;
        return SMO_OK;
    }
    if (!a2a || !ared) { set_error("smo_comm_set_transport: null function"); return SMO_ERR_ARG; }
    if (ready()) { set_error("smo_comm_set_transport: the context already has a communicator"); return SMO_ERR_STATE; }
    a2a_ = a2a; ared_ = ared; user_ = user; rank = r; world = w;
    return SMO_OK;
}

void SlabComm::reset() {
    if (nccl_ && g_rccl.handle) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(nccl_));
    nccl_ = nullptr; a2a_ = nullptr; ared_ = nullptr; user_ = nullptr; peers_ = nullptr; null_ = false;
}

int SlabComm::set_peers(int r, PeerGroup* g) {
    if (!g) { set_error("set_peers: null group"); return SMO_ERR_ARG; }
    if (ready()) { set_error("set_peers: the context already has a communicator"); return SMO_ERR_STATE; }
    peers_ = g; rank = r; world = g->world();
    return SMO_OK;
}

// ---- PeerGroup: ranks = threads of one process, one GPU each ---------------------------------------------------------------
// The pull of one exchange as ONE kernel: with peer access enabled a rank reads its blocks straight out of its W - 1 peers' send buffers (xGMI
// loads) and writes them to its own receive buffer; blockIdx.y walks the peers (starting with the rank itself, so that at any moment the ranks
// read from different sources), blockIdx.x strides over a block.  All links of the GPU carry traffic at once — W - 1 hipMemcpyPeerAsync calls on
// one stream would run one after the other, one link at a time — and an exchange costs the host one launch instead of W - 1 copies.
struct PeerSrcs { const double2* p[64]; };
__global__ __launch_bounds__(256) void peer_gather(double2* __restrict__ dst, PeerSrcs srcs, int W, int rank, size_t n2) {
    const int p = (rank + (int)blockIdx.y) % W;
    const double2* __restrict__ from = srcs.p[p] + (size_t)rank * n2;
    double2* __restrict__ to = dst + (size_t)p * n2;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) to[i] = from[i];
}

PeerGroup::PeerGroup(const std::vector<int>& devices) : dev(devices) {
    const int W = (int)dev.size();
    pub_src.assign(2 * (size_t)W, nullptr); pub_dst.assign(2 * (size_t)W, nullptr);
    seq.assign((size_t)W * 8, 0);
    ev_ready.assign(W, nullptr); ev_pulled.assign(W, nullptr);
    red.assign((size_t)W * 64, 0.0);
    wait_ms.assign((size_t)W * 8, 0.0);          // (stride 8 doubles = one cache line per rank below: see barrier())
    peer_access = true;
    for (int r = 0; r < W; ++r) {
        if (hipSetDevice(dev[r]) != hipSuccess) { peer_access = false; continue; }
        // system-scope release: what the event orders must be visible to a kernel on ANOTHER device
        (void)hipEventCreateWithFlags(&ev_ready[r], hipEventDisableTiming | hipEventReleaseToSystem);
        (void)hipEventCreateWithFlags(&ev_pulled[r], hipEventDisableTiming | hipEventReleaseToSystem);
        for (int p = 0; p < W; ++p) {
            if (dev[p] == dev[r]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dev[r], dev[p]) != hipSuccess || !can) { peer_access = false; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(dev[p], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_access = false;
            (void)hipGetLastError();
        }
    }
    for (int r = 1; r < W; ++r) distinct = distinct || dev[r] != dev[0];
    // the gather kernel needs peer access between every pair of distinct devices; SMO_PEER_COPY=memcpy forces the copy calls (which do not),
    // SMO_PEER_COPY=kernel the gather kernel.  Default: the kernel between ranks of ONE device (where it has run: every test of a one-GPU
    // box), the copy calls as soon as two ranks sit on different devices — a kernel reading a peer's HBM over xGMI has not yet run on
    // hardware; tests/test_kdyn_multi_gpu.py compares the two on such a node and the default follows once that has passed.
    const char* e = getenv("SMO_PEER_COPY");
    const std::string mode = e ? std::string(e) : std::string();
    use_kernel = peer_access && (mode == "kernel" || (mode != "memcpy" && !distinct));
}
PeerGroup::~PeerGroup() {
    for (size_t r = 0; r < dev.size(); ++r) {
        (void)hipSetDevice(dev[r]);
        if (ev_ready[r]) (void)hipEventDestroy(ev_ready[r]);
        if (ev_pulled[r]) (void)hipEventDestroy(ev_pulled[r]);
    }
}
int PeerGroup::barrier(int rank) {
    // generation-counter spin barrier: the last arrival resets the count and bumps the generation (release); the others poll it (acquire)
    struct Waited {
        double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Waited() { acc += 1e-6 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    } waited{wait_ms[(size_t)rank * 8]};
    if (failed.load(std::memory_order_acquire)) { set_error("another rank of the multi-device context failed"); return SMO_ERR_STATE; }
    const unsigned long gen = generation.load(std::memory_order_acquire);
    if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == world()) {
        waiting.store(0, std::memory_order_relaxed);
        generation.store(gen + 1, std::memory_order_release);
        return SMO_OK;
    }
    for (unsigned spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins) {
        if (failed.load(std::memory_order_acquire)) { set_error("another rank of the multi-device context failed"); return SMO_ERR_STATE; }
        if (spins < 4096) __builtin_ia32_pause();
        else std::this_thread::yield();                  // a peer is inside a long HIP call (or the host is oversubscribed): give the core away
    }
    return SMO_OK;
}
void PeerGroup::abort() { failed.store(true, std::memory_order_release); }
void PeerGroup::reset() {
    failed.store(false, std::memory_order_release);
    waiting.store(0, std::memory_order_release);
    for (size_t r = 0; r < dev.size(); ++r) seq[r * 8] = 0;      // (a failed call may have left the ranks at different counts)
}
int PeerGroup::alltoall(int rank, const void* src, void* dst, size_t bytes, hipStream_t s, bool chained) {
    const int W = world();
    const size_t slot = (size_t)(seq[(size_t)rank * 8]++ & 1ul) * W;      // every rank runs the same sequence of exchanges: same parity everywhere
    pub_src[slot + rank] = src; pub_dst[slot + rank] = dst;
    if (rank == 0) barriers += chained ? 1 : 2;
    SMO_HIP(hipEventRecord(ev_ready[rank], s));
    SMO_TRY(barrier(rank));
    for (int p = 0; p < W; ++p)
        if (p != rank) SMO_HIP(hipStreamWaitEvent(s, ev_ready[p], 0));
    if (use_kernel && bytes % 16 == 0 && W <= 64) {
        PeerSrcs srcs{};
        for (int p = 0; p < W; ++p) srcs.p[p] = static_cast<const double2*>(pub_src[slot + p]);
        const size_t n2 = bytes / 16;
        const unsigned gx = (unsigned)std::min<size_t>(128, (n2 + 255) / 256);
        hipLaunchKernelGGL(peer_gather, dim3(gx, W), dim3(256), 0, s, static_cast<double2*>(dst), srcs, W, rank, n2);
        SMO_HIP(hipGetLastError());
    } else {
        for (int q = 0; q < W; ++q) {
            const int p = (rank + q) % W;                    // start with my own block, then walk the ring: the peers pull from different sources
            const char* from = static_cast<const char*>(pub_src[slot + p]) + (size_t)rank * bytes;
            char* to = static_cast<char*>(dst) + (size_t)p * bytes;
            if (dev[p] == dev[rank]) SMO_HIP(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, s));
            else SMO_HIP(hipMemcpyPeerAsync(to, dev[rank], from, dev[p], bytes, s));
        }
    }
    if (chained) return SMO_OK;                              // the next exchange's pull orders the re-use of `src` (comm.hpp)
    SMO_HIP(hipEventRecord(ev_pulled[rank], s));
    SMO_TRY(barrier(rank));
    for (int p = 0; p < W; ++p)
        if (p != rank) SMO_HIP(hipStreamWaitEvent(s, ev_pulled[p], 0));
    return SMO_OK;
}
int PeerGroup::allreduce_sum(int rank, double* vals, int n, hipStream_t s) {
    if (n > 64) { set_error("PeerGroup::allreduce_sum: at most 64 values"); return SMO_ERR_ARG; }
    SMO_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) red[(size_t)rank * 64 + i] = vals[i];
    SMO_TRY(barrier(rank));
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int p = 0; p < world(); ++p) a += red[(size_t)p * 64 + i];      // the same order on every rank: identical sums
        vals[i] = a;
    }
    return barrier(rank);                                    // nobody overwrites its slot before everyone has read it
}
SlabComm::~SlabComm() { reset(); }

const char* SlabComm::library_path() {
    std::call_once(g_once, load_rccl);
    return g_rccl.handle ? g_lib_path.c_str() : "";
}

int SlabComm::alltoall(const void* src, void* dst, size_t bytes_per_peer, hipStream_t s, bool chained) {
    if (peers_) return peers_->alltoall(rank, src, dst, bytes_per_peer, s, chained);
    if (null_) return SMO_OK;
    if (nccl_) {
        ncclComm_t c = static_cast<ncclComm_t>(nccl_);
        const size_t cnt = bytes_per_peer / sizeof(double);
        SMO_NCCL(g_rccl.GroupStart());
        for (int p = 0; p < world; ++p) {
            SMO_NCCL(g_rccl.Send(static_cast<const char*>(src) + (size_t)p * bytes_per_peer, cnt, ncclDouble, p, c, s));
            SMO_NCCL(g_rccl.Recv(static_cast<char*>(dst) + (size_t)p * bytes_per_peer, cnt, ncclDouble, p, c, s));
        }
        SMO_NCCL(g_rccl.GroupEnd());
        return SMO_OK;
    }
    if (a2a_) {
        const int rc = a2a_(user_, src, dst, bytes_per_peer, s);
        if (rc != 0) { set_error("the caller's all-to-all transport failed (%d)", rc); return SMO_ERR_HIP; }
        return SMO_OK;
    }
    set_error("no communicator: call smo_comm_init (RCCL) or smo_comm_set_transport first");
    return SMO_ERR_STATE;
}

int SlabComm::allreduce_sum(double* vals, int n, hipStream_t s, double* dev_scratch) {
    if (peers_) return peers_->allreduce_sum(rank, vals, n, s);
    if (null_) {                                           // as if every rank had contributed what this one did
        SMO_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) vals[i] *= world;
        return SMO_OK;
    }
    if (nccl_) {
        ncclComm_t c = static_cast<ncclComm_t>(nccl_);
        SMO_HIP(hipMemcpyAsync(dev_scratch, vals, n * sizeof(double), hipMemcpyHostToDevice, s));
        SMO_NCCL(g_rccl.AllReduce(dev_scratch, dev_scratch, n, ncclDouble, ncclSum, c, s));
        SMO_HIP(hipMemcpyAsync(vals, dev_scratch, n * sizeof(double), hipMemcpyDeviceToHost, s));
        SMO_HIP(hipStreamSynchronize(s));
        return SMO_OK;
    }
    if (ared_) {
        SMO_HIP(hipStreamSynchronize(s));
        const int rc = ared_(user_, vals, n);
        if (rc != 0) { set_error("the caller's all-reduce transport failed (%d)", rc); return SMO_ERR_HIP; }
        return SMO_OK;
    }
    set_error("no communicator: call smo_comm_init (RCCL) or smo_comm_set_transport first");
    return SMO_ERR_STATE;
}

}  // namespace smo
