// SlabComm: RCCL (dlopen'ed) or caller-provided transport.  See comm.hpp.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
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
    if (!a2a || !ared) { set_error("smo_comm_set_transport: null function"); return SMO_ERR_ARG; }
    if (ready()) { set_error("smo_comm_set_transport: the context already has a communicator"); return SMO_ERR_STATE; }
    a2a_ = a2a; ared_ = ared; user_ = user; rank = r; world = w;
    return SMO_OK;
}

void SlabComm::reset() {
    if (nccl_ && g_rccl.handle) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(nccl_));
    nccl_ = nullptr; a2a_ = nullptr; ared_ = nullptr; user_ = nullptr;
}
SlabComm::~SlabComm() { reset(); }

const char* SlabComm::library_path() {
    std::call_once(g_once, load_rccl);
    return g_rccl.handle ? g_lib_path.c_str() : "";
}

int SlabComm::alltoall(const void* src, void* dst, size_t bytes_per_peer, hipStream_t s) {
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
