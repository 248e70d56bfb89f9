// extern "C" surface of libsmo (declarations and reference citations: include/smo.h).
#include "smo_common.hpp"
#include "comm.hpp"

using smo::Context;

struct smo_ctx {
    Context* impl;
};

#define CHECK_CTX(c)                                  \
    do {                                              \
        if (!(c) || !(c)->impl) {                     \
            smo::set_error("null context");           \
            return SMO_ERR_ARG;                       \
        }                                             \
    } while (0)

extern "C" {

const char* smo_last_error(void) { return smo::last_error(); }
const char* smo_version(void) { return "libsmo 0.1.0 (gfx950)"; }

int smo_device_count(int* count) {
    if (!count) return SMO_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return SMO_OK;
}

int smo_create(const smo_config* cfg, smo_ctx** out) {
    if (!cfg || !out) { smo::set_error("smo_create: null argument"); return SMO_ERR_ARG; }
    *out = nullptr;
    if (cfg->batch < 1 || cfg->n_iters < 1 || cfg->npts < 4 || !(cfg->dt > 0) || !(cfg->x1 > cfg->x0) || cfg->world < 1 ||
        cfg->rank < 0 || cfg->rank >= cfg->world || cfg->ckpt < 0) {
        smo::set_error("smo_create: bad config (npts=%d n_iters=%d dt=%g batch=%d interval=[%g,%g] rank=%d/%d)", cfg->npts,
                       cfg->n_iters, cfg->dt, cfg->batch, cfg->x0, cfg->x1, cfg->rank, cfg->world);
        return SMO_ERR_ARG;
    }
    Context* c = nullptr;
    switch (cfg->kind) {
        case SMO_SH23: c = smo::make_sh23(*cfg); break;
        case SMO_SHB23: c = smo::make_shb23(*cfg); break;
        case SMO_KDYN: c = smo::make_kdyn(*cfg); break;
        case SMO_POIS: c = smo::make_pois(*cfg); break;
        default: smo::set_error("smo_create: unknown kind %d", cfg->kind); return SMO_ERR_ARG;
    }
    if (!c) return SMO_ERR_UNSUPPORTED;      // message set by the factory
    int rc = c->init();
    if (rc != SMO_OK) { delete c; return rc; }
    *out = new smo_ctx{c};
    return SMO_OK;
}

int smo_create_multi(const smo_config* cfg, int ndev, const int* dev_ids, smo_ctx** out) {
    if (!cfg || !out || !dev_ids) { smo::set_error("smo_create_multi: null argument"); return SMO_ERR_ARG; }
    *out = nullptr;
    if (cfg->batch != 1 || cfg->n_iters < 1 || cfg->npts < 4 || !(cfg->dt > 0) || !(cfg->x1 > cfg->x0) || cfg->ckpt < 0 || ndev < 1 || ndev > 64) {
        smo::set_error("smo_create_multi: bad config (npts=%d n_iters=%d dt=%g batch=%d ndev=%d)", cfg->npts, cfg->n_iters, cfg->dt, cfg->batch, ndev);
        return SMO_ERR_ARG;
    }
    Context* c = smo::make_multi(*cfg, ndev, dev_ids);
    int rc = c->init();
    if (rc != SMO_OK) { delete c; return rc; }
    *out = new smo_ctx{c};
    return SMO_OK;
}

void smo_destroy(smo_ctx* ctx) {
    if (!ctx) return;
    if (ctx->impl) {
        (void)hipSetDevice(ctx->impl->cfg.device);
        delete ctx->impl;
    }
    delete ctx;
}

int smo_ncomp(const smo_ctx* ctx) { return (ctx && ctx->impl) ? ctx->impl->n_comp : 0; }

int smo_vec_len(const smo_ctx* ctx, size_t* len) {
    CHECK_CTX(ctx);
    if (!len) return SMO_ERR_ARG;
    *len = ctx->impl->vec_len;
    return SMO_OK;
}
int smo_stack_bytes(const smo_ctx* ctx, size_t* bytes) {
    CHECK_CTX(ctx);
    if (!bytes) return SMO_ERR_ARG;
    *bytes = ctx->impl->stack_bytes;
    return SMO_OK;
}

int smo_get(const smo_ctx* ctx, int key, double* value) {
    CHECK_CTX(ctx);
    if (!value || key < 0 || key > 5) { smo::set_error("smo_get: bad argument"); return SMO_ERR_ARG; }
    *value = ctx->impl->info(key);
    return SMO_OK;
}

// `dev`: the list holds device pointers (a multi-device context then dereferences one slab per component AND device: every one is checked)
static int check_vecs(const smo_ctx* ctx, const double* const* X, const char* who, bool dev = false) {
    if (!X) { smo::set_error("%s: null vector list", who); return SMO_ERR_ARG; }
    const int n = dev ? ctx->impl->n_dev_ptrs() : ctx->impl->n_comp;
    for (int c = 0; c < n; ++c)
        if (!X[c]) { smo::set_error("%s: pointer %d of %d is null", who, c, n); return SMO_ERR_ARG; }
    return SMO_OK;
}

int smo_forward(smo_ctx* ctx, const double* const* X, double* J) {
    CHECK_CTX(ctx);
    SMO_TRY(check_vecs(ctx, X, "smo_forward"));
    if (!J) return SMO_ERR_ARG;
    return ctx->impl->forward_host(X, J);
}
int smo_forward_dev(smo_ctx* ctx, const double* const* X, double* J) {
    CHECK_CTX(ctx);
    SMO_TRY(check_vecs(ctx, X, "smo_forward_dev", true));
    if (!J) return SMO_ERR_ARG;
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->forward_dev(X, J);
}

static int check_adjoint(smo_ctx* ctx, int adjoint_type, double* const* grad, bool dev = false) {
    if (adjoint_type != SMO_ADJ_DISCRETE && adjoint_type != SMO_ADJ_CONTINUOUS) {
        smo::set_error("smo_adjoint: adjoint_type %d", adjoint_type);
        return SMO_ERR_ARG;
    }
    if (!ctx->impl->have_forward) {
        smo::set_error("smo_adjoint: no forward solve on this context yet (the adjoint replays its snapshot stack)");
        return SMO_ERR_STATE;
    }
    return check_vecs(ctx, grad, "smo_adjoint(grad)", dev);
}
int smo_adjoint(smo_ctx* ctx, const double* const* X, int adjoint_type, double* const* grad) {
    CHECK_CTX(ctx);
    SMO_TRY(check_adjoint(ctx, adjoint_type, grad));
    return ctx->impl->adjoint_host(X, adjoint_type, grad);
}
int smo_adjoint_dev(smo_ctx* ctx, const double* const* X, int adjoint_type, double* const* grad) {
    CHECK_CTX(ctx);
    SMO_TRY(check_adjoint(ctx, adjoint_type, grad, true));
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->adjoint_dev(X, adjoint_type, grad);
}

int smo_inner(smo_ctx* ctx, const double* x, const double* y, double* out) {
    CHECK_CTX(ctx);
    if (!x || !y || !out) { smo::set_error("smo_inner: null argument"); return SMO_ERR_ARG; }
    return ctx->impl->inner_host(x, y, out);
}
int smo_inner_dev(smo_ctx* ctx, const double* x, const double* y, double* out) {
    CHECK_CTX(ctx);
    if (!x || !y || !out) { smo::set_error("smo_inner_dev: null argument"); return SMO_ERR_ARG; }
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->inner_dev(x, y, out);
}

int smo_inner_slabs(smo_ctx* ctx, const double* const* x, const double* const* y, double* out) {
    CHECK_CTX(ctx);
    if (!x || !y || !out) { smo::set_error("smo_inner_slabs: null argument"); return SMO_ERR_ARG; }
    for (int i = 0; i < ctx->impl->n_slabs(); ++i)
        if (!x[i] || !y[i]) { smo::set_error("smo_inner_slabs: slab %d of %d is null", i, ctx->impl->n_slabs()); return SMO_ERR_ARG; }
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->inner_slabs(x, y, out);
}

int smo_snapshot_len(const smo_ctx* ctx, size_t* n) {
    CHECK_CTX(ctx);
    if (!n) return SMO_ERR_ARG;
    *n = ctx->impl->snapshot_doubles;
    return SMO_OK;
}
int smo_snapshot_read(smo_ctx* ctx, int b, int index, double* out) {
    CHECK_CTX(ctx);
    if (!out || b < 0 || b >= ctx->impl->cfg.batch || index < 0 || index > ctx->impl->cfg.n_iters) {
        smo::set_error("smo_snapshot_read: bad argument (b=%d index=%d)", b, index);
        return SMO_ERR_ARG;
    }
    if (!ctx->impl->have_forward) { smo::set_error("smo_snapshot_read: no forward solve yet"); return SMO_ERR_STATE; }
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->snapshot_read(b, index, out);
}

int smo_transform(smo_ctx* ctx, int which, const double* in, double* out) {
    CHECK_CTX(ctx);
    if (!in || !out || which < 0 || which > 3) { smo::set_error("smo_transform: bad argument"); return SMO_ERR_ARG; }
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->transform_host(which, in, out);
}

int smo_kdyn_op(smo_ctx* ctx, int op, int i0, int i1, void* p0, void* p1, double* out) {
    CHECK_CTX(ctx);
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->kdyn_op(op, i0, i1, p0, p1, out);
}

int smo_comm_unique_id(void* id128) {
    if (!id128) { smo::set_error("smo_comm_unique_id: null argument"); return SMO_ERR_ARG; }
    return smo::SlabComm::unique_id(id128);
}
int smo_comm_init(smo_ctx* ctx, const void* id128) {
    CHECK_CTX(ctx);
    if (!id128) { smo::set_error("smo_comm_init: null id"); return SMO_ERR_ARG; }
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->comm_init(id128);
}
int smo_comm_set_transport(smo_ctx* ctx, smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user) {
    CHECK_CTX(ctx);
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->comm_set_transport(a2a, ared, user);
}
const char* smo_comm_library(void) { return smo::SlabComm::library_path(); }
int smo_comm_get(const smo_ctx* ctx, int key, double* value) {
    CHECK_CTX(ctx);
    if (!value || key < 0 || key > 4) { smo::set_error("smo_comm_get: bad argument"); return SMO_ERR_ARG; }
    *value = ctx->impl->comm_info(key);
    return SMO_OK;
}

int smo_set_stream(smo_ctx* ctx, void* hip_stream) {
    CHECK_CTX(ctx);
    SMO_HIP(hipSetDevice(ctx->impl->cfg.device));
    return ctx->impl->set_stream(static_cast<hipStream_t>(hip_stream));
}

int smo_timing_enable(smo_ctx* ctx, int on) {
    CHECK_CTX(ctx);
    if (on < 0 || on - 2 >= 64) { smo::set_error("smo_timing_enable: on = %d (0, 1, or 2 + class index < 64)", on); return SMO_ERR_ARG; }
    ctx->impl->tm().reset();
    ctx->impl->tm().on = (on != 0);
    ctx->impl->tm().mask = (on >= 2) ? (1ull << (on - 2)) : ~0ull;
    return SMO_OK;
}
int smo_timing_select(smo_ctx* ctx, unsigned long long class_mask) {
    CHECK_CTX(ctx);
    ctx->impl->tm().reset();
    ctx->impl->tm().on = (class_mask != 0);
    ctx->impl->tm().mask = class_mask;
    return SMO_OK;
}
int smo_timing_stride(smo_ctx* ctx, int every) {
    CHECK_CTX(ctx);
    if (every < 1) { smo::set_error("smo_timing_stride: every = %d (>= 1 expected)", every); return SMO_ERR_ARG; }
    ctx->impl->tm().stride = every;
    return SMO_OK;
}
int smo_timing_classes(const smo_ctx* ctx) { return (ctx && ctx->impl) ? (int)ctx->impl->tm().cls.size() : 0; }
int smo_timing_get(smo_ctx* ctx, int k, const char** name, long long* launches, double* total_ms, double* bytes_per_launch) {
    CHECK_CTX(ctx);
    auto& t = ctx->impl->tm();
    if (k < 0 || k >= (int)t.cls.size()) { smo::set_error("smo_timing_get: class %d", k); return SMO_ERR_ARG; }
    (void)ctx->impl->sync_all();
    t.flush();
    if (name) *name = t.cls[k].name.c_str();
    if (launches) *launches = t.cls[k].launches;
    if (total_ms) *total_ms = t.cls[k].total_ms;
    if (bytes_per_launch) *bytes_per_launch = t.cls[k].bytes_per_launch;
    return SMO_OK;
}
int smo_timing_hbm_bytes(smo_ctx* ctx, int k, double* bytes_per_launch) {
    CHECK_CTX(ctx);
    auto& t = ctx->impl->tm();
    if (k < 0 || k >= (int)t.cls.size() || !bytes_per_launch) { smo::set_error("smo_timing_hbm_bytes: class %d", k); return SMO_ERR_ARG; }
    *bytes_per_launch = t.cls[k].hbm_bytes;
    return SMO_OK;
}

}  // extern "C"
