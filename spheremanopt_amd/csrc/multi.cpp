// One host process driving several GPUs of a node: the slab-decomposed 3-D case inside ONE context (smo_create_multi).
//
// the reference's Optimise_On_Multi_Sphere is a single Python process; under `mpiexec -np P` (README.md:83) every rank runs
// it redundantly and Dedalus moves the data (FWD_Solve_KDyn.py:118-134).  Here the caller stays ONE process with ONE context and hands over the
// reference's full flat vectors [3][G][G][G]; the library
//   * creates one KDYN slab context per device (rank i of ndev, csrc/kdyn.hip — the same kernels and the same in-library time loop as with one
//     process per GPU) and joins them in a PeerGroup (csrc/comm.hpp): a transpose = every rank pulling its blocks from the peers' send buffers
//     over the node's point-to-point links (one gather kernel, or hipMemcpyPeerAsync calls), ordered by HIP events; no RCCL, no launcher;
//   * runs every collective call (forward, adjoint, inner product) on one PERSISTENT worker thread per device (created with the context,
//     bound to its device once, woken per call) — a single host thread cannot issue the launches of 8 GPUs fast enough (about 70 launches
//     and as many event operations per step pair against 0.6 ms of kernels at 256^3 / 8);
//   * scatters X to / gathers grad J from the devices slab by slab (strided 2-D copies: z is the fastest axis and the one that is split).
// A failing rank releases the others (PeerGroup::abort): the call returns its error instead of hanging.
#include <algorithm>
#include <functional>
#include <memory>
#include <thread>

#include "comm.hpp"

namespace smo {
namespace {

class MultiKDyn : public Context {
public:
    MultiKDyn(const smo_config& c, int ndev, const int* ids) { cfg = c; dev.assign(ids, ids + ndev); }
    std::vector<int> dev;
    std::vector<std::unique_ptr<Context>> r;       // one slab context per device
    std::unique_ptr<PeerGroup> grp;
    int W = 0, G = 0, Gzr = 0;
    size_t n_local = 0;                             // doubles of one component's slab [3][G][G][Gzr]
    std::vector<double*> sx[2], sg[2];              // per rank: device staging of X / grad (host-buffer entry points)

    ~MultiKDyn() override {
        stop_workers();
        // members first: their destructors synchronise their streams on their own devices
        for (int i = 0; i < (int)r.size(); ++i) { (void)hipSetDevice(dev[i]); r[i].reset(); }
        grp.reset();
    }

    // ---- persistent workers: one host thread per device for the life of the context (round 3 spawned and joined W threads per call) ----
    struct Pool {
        std::vector<std::thread> th;
        std::mutex mu;
        std::condition_variable cv_job, cv_done;
        std::function<int(int)> job;
        unsigned long epoch = 0;
        int done = 0;
        bool quit = false;
        std::vector<int> rc;
        std::vector<std::string> msg;
    } pool_;
    void worker(int i) {
        const bool bound = hipSetDevice(dev[i]) == hipSuccess;
        unsigned long seen = 0;
        for (;;) {
            std::function<int(int)> f;
            {
                std::unique_lock<std::mutex> lk(pool_.mu);
                pool_.cv_job.wait(lk, [&] { return pool_.quit || pool_.epoch != seen; });
                if (pool_.quit) return;
                seen = pool_.epoch;
                f = pool_.job;
            }
            int rc = SMO_OK;
            std::string msg;
            if (!bound) { rc = SMO_ERR_HIP; msg = "hipSetDevice failed"; }
            else { rc = f(i); if (rc != SMO_OK) msg = last_error(); }
            if (rc != SMO_OK) grp->abort();              // release the ranks that wait for this one in a collective
            {
                std::lock_guard<std::mutex> lk(pool_.mu);
                pool_.rc[i] = rc; pool_.msg[i] = msg;
                if (++pool_.done == W) pool_.cv_done.notify_one();
            }
        }
    }
    void start_workers() {
        pool_.rc.assign(W, SMO_OK); pool_.msg.assign(W, std::string());
        for (int i = 0; i < W; ++i) pool_.th.emplace_back([this, i] { worker(i); });
    }
    void stop_workers() {
        { std::lock_guard<std::mutex> lk(pool_.mu); pool_.quit = true; }
        pool_.cv_job.notify_all();
        for (auto& t : pool_.th) if (t.joinable()) t.join();
        pool_.th.clear();
    }
    // run f(rank) on every device's worker; first error wins (its message is carried to the calling thread)
    int on_all(std::function<int(int)> f) {
        if (pool_.th.empty()) { set_error("multi-device context: no worker threads"); return SMO_ERR_STATE; }
        grp->reset();
        {
            std::unique_lock<std::mutex> lk(pool_.mu);
            pool_.job = std::move(f);
            pool_.done = 0;
            ++pool_.epoch;
            pool_.cv_job.notify_all();
            pool_.cv_done.wait(lk, [&] { return pool_.done == W; });
            pool_.job = nullptr;
        }
        const std::vector<int>& rc = pool_.rc;
        const std::vector<std::string>& msg = pool_.msg;
        // report the rank that failed on its own, not the ones that were released from a collective because of it
        int bad = -1;
        for (int i = 0; i < W; ++i)
            if (rc[i] != SMO_OK && (bad < 0 || msg[bad].find("another rank of the multi-device context failed") != std::string::npos)) bad = i;
        if (bad >= 0) { set_error("device %d (rank %d of %d): %s", dev[bad], bad, W, msg[bad].c_str()); return rc[bad]; }
        return SMO_OK;
    }
    int n_dev_ptrs() const override { return 2 * W; }      // smo_forward_dev / smo_adjoint_dev: one slab pointer per (component, device)
    int n_slabs() const override { return W; }
    int set_stream(hipStream_t) override {
        set_error("smo_set_stream: a multi-device context runs every rank on a private stream of its own device; a caller's stream cannot order them");
        return SMO_ERR_UNSUPPORTED;
    }

    int init() override {
        W = (int)dev.size();
        if (cfg.kind != SMO_KDYN) { set_error("smo_create_multi: KDYN only (the 1-D problems do not shard: SURVEY.md 8e)"); return SMO_ERR_UNSUPPORTED; }
        if (W < 2) { set_error("smo_create_multi: ndev = %d (use smo_create for one device)", W); return SMO_ERR_ARG; }
        if (cfg.world != 1 || cfg.rank != 0) { set_error("smo_create_multi: smo_config.rank / world describe one process per GPU; leave them 0 / 1"); return SMO_ERR_ARG; }
        int ndevs = 0;
        if (hipGetDeviceCount(&ndevs) != hipSuccess || ndevs <= 0) { set_error("no usable HIP device; libsmo has no CPU fallback"); return SMO_ERR_NO_DEVICE; }
        for (int d : dev)
            if (d < 0 || d >= ndevs) { set_error("smo_create_multi: device %d out of range (have %d)", d, ndevs); return SMO_ERR_ARG; }
        G = 3 * cfg.npts / 2;
        if (G % W != 0 || (cfg.npts / 2) % W != 0) { set_error("smo_create_multi: %d devices do not divide a = %d kx modes and G = %d grid planes", W, cfg.npts / 2, G); return SMO_ERR_UNSUPPORTED; }
        Gzr = G / W;
        n_local = (size_t)3 * G * G * Gzr;
        n_comp = 2;
        vec_len = (size_t)3 * G * G * G;                  // the caller's vectors are the reference's full ones
        cfg.device = dev[0];
        grp.reset(new PeerGroup(dev));
        start_workers();
        r.resize(W);
        // members one after the other (each sizes its stack from the free HBM of its own device; several ranks may share a device in tests)
        for (int i = 0; i < W; ++i) {
            smo_config c = cfg;
            c.rank = i; c.world = W; c.device = dev[i];
            SMO_HIP(hipSetDevice(dev[i]));
            r[i].reset(make_kdyn(c));
            if (!r[i]) return SMO_ERR_UNSUPPORTED;
            SMO_TRY(r[i]->init());
        }
        // collective: the ranks agree on the checkpoint interval and on the kept grid-side states, and set up their exchange pipeline
        SMO_TRY(on_all([&](int i) { return r[i]->comm_set_peers(grp.get(), i); }));
        stack_bytes = 0;
        for (auto& m : r) stack_bytes += m->stack_bytes;
        snapshot_doubles = 0;                              // snapshots stay distributed (smo_snapshot_read: not available)
        SMO_HIP(hipSetDevice(dev[0]));
        SMO_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int c = 0; c < 2; ++c) { sx[c].assign(W, nullptr); sg[c].assign(W, nullptr); }
        return SMO_OK;
    }

    int stage_buffers(int i) {
        for (int c = 0; c < 2; ++c) {
            if (!sx[c][i]) SMO_TRY(r[i]->pool.alloc(&sx[c][i], n_local));
            if (!sg[c][i]) SMO_TRY(r[i]->pool.alloc(&sg[c][i], n_local));
        }
        return SMO_OK;
    }
    // full host vector [3][G][G][G]  <->  rank i's slab [3][G][G][Gzr] (its z planes i*Gzr ...)
    int scatter(int i, const double* full, double* slab) {
        SMO_HIP(hipMemcpy2DAsync(slab, (size_t)Gzr * 8, full + (size_t)i * Gzr, (size_t)G * 8, (size_t)Gzr * 8, (size_t)3 * G * G, hipMemcpyHostToDevice, r[i]->stream));
        return SMO_OK;
    }
    int gather(int i, const double* slab, double* full) {
        SMO_HIP(hipMemcpy2DAsync(full + (size_t)i * Gzr, (size_t)G * 8, slab, (size_t)Gzr * 8, (size_t)Gzr * 8, (size_t)3 * G * G, hipMemcpyDeviceToHost, r[i]->stream));
        return SMO_OK;
    }

    // ---- device-resident form: X[c * W + i] = slab of component c on device i ----------------------------------------------
    int forward_dev(const double* const* X, double* J) override {
        have_forward = false;
        std::vector<double> Ji(W, 0.0);
        SMO_TRY(on_all([&](int i) { const double* x[2] = {X[i], X[W + i]}; return r[i]->forward_dev(x, &Ji[i]); }));
        *J = Ji[0];                                        // every rank holds the reduced value
        have_forward = true;
        return SMO_OK;
    }
    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        return on_all([&](int i) { double* g[2] = {grad[i], grad[W + i]}; return r[i]->adjoint_dev(nullptr, adjoint_type, g); });
    }
    int inner_dev(const double* x, const double* y, double* out) override {
        (void)x; (void)y; (void)out;
        set_error("smo_inner_dev on a multi-device context: pass host vectors (smo_inner) or one slab pointer per device (smo_inner_slabs)");
        return SMO_ERR_UNSUPPORTED;
    }
    int inner_slabs(const double* const* x, const double* const* y, double* out) override {
        std::vector<double> oi(W, 0.0);
        SMO_TRY(on_all([&](int i) { return r[i]->inner_dev(x[i], y[i], &oi[i]); }));      // reduced over the ranks inside (collective)
        *out = oi[0];
        return SMO_OK;
    }

    // ---- the reference's callbacks: full host vectors in, full host vectors out ------------------------------------------------
    int forward_host(const double* const* X, double* J) override {
        have_forward = false;
        std::vector<double> Ji(W, 0.0);
        SMO_TRY(on_all([&](int i) -> int {
            SMO_TRY(stage_buffers(i));
            for (int c = 0; c < 2; ++c) SMO_TRY(scatter(i, X[c], sx[c][i]));
            const double* x[2] = {sx[0][i], sx[1][i]};
            return r[i]->forward_dev(x, &Ji[i]);
        }));
        *J = Ji[0];
        have_forward = true;
        return SMO_OK;
    }
    int adjoint_host(const double* const*, int adjoint_type, double* const* grad) override {
        return on_all([&](int i) -> int {
            SMO_TRY(stage_buffers(i));
            double* g[2] = {sg[0][i], sg[1][i]};
            SMO_TRY(r[i]->adjoint_dev(nullptr, adjoint_type, g));
            for (int c = 0; c < 2; ++c) SMO_TRY(gather(i, sg[c][i], grad[c]));
            SMO_HIP(hipStreamSynchronize(r[i]->stream));
            return SMO_OK;
        });
    }
    int inner_host(const double* x, const double* y, double* out) override {
        std::vector<double> oi(W, 0.0);
        SMO_TRY(on_all([&](int i) -> int {
            SMO_TRY(stage_buffers(i));
            SMO_TRY(scatter(i, x, sg[0][i]));
            SMO_TRY(scatter(i, y, sg[1][i]));
            return r[i]->inner_dev(sg[0][i], sg[1][i], &oi[i]);      // reduced over the ranks inside (collective)
        }));
        *out = oi[0];
        return SMO_OK;
    }

    int snapshot_read(int, int, double*) override {
        set_error("smo_snapshot_read: the snapshots of a multi-device context stay distributed over its devices");
        return SMO_ERR_UNSUPPORTED;
    }
    Timing& tm() override { return r[0]->timing; }
    int sync_all() override {
        int cur = 0;
        SMO_HIP(hipGetDevice(&cur));
        int rc = SMO_OK;
        for (int i = 0; i < W && rc == SMO_OK; ++i)
            if (hipSetDevice(dev[i]) != hipSuccess || hipStreamSynchronize(r[i]->stream) != hipSuccess) { set_error("multi-device context: sync of device %d failed", dev[i]); rc = SMO_ERR_HIP; }
        (void)hipSetDevice(cur);                           // the caller's current device is not ours to change
        return rc;
    }
    double info(int key) const override {
        if (r.empty()) return 0.0;
        if (key == 4) { double m = 0.0; for (auto& x : r) m = std::max(m, x->info(4)); return m; }      // host issue time: the slowest worker
        return r[0]->info(key);
    }
    double comm_info(int key) const override {
        if (key == 2) return 0.0;                          // no RCCL communicator here
        if (key == 3) return grp ? (grp->use_kernel ? 2.0 : 1.0) : 0.0;      // transposes: 2 = gather kernel, 1 = hipMemcpyPeerAsync calls
        if (key == 4) return grp ? (double)grp->barriers : 0.0;              // host rendezvous passed so far
        return r.empty() ? 0.0 : r[0]->comm_info(key);
    }
};

}  // namespace

Context* make_multi(const smo_config& cfg, int ndev, const int* dev_ids) { return new MultiKDyn(cfg, ndev, dev_ids); }

}  // namespace smo
