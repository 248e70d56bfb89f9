/*
 * smo.h — C-ABI of the MI355X-native forward/adjoint spectral-solve hot path of SphereManOpt.
 *
 * The reference is pure Python; its optimiser (Sphere_Grad_Descent.py:692) reaches the hot path only through
 * three callbacks  f / Grad_f / Inner_Product  (SURVEY.md section 8b).  This header is what those callbacks
 * bind to (ctypes stub: INTEGRATION.md; shipped binding: spheremanopt_amd/_capi.py):
 *
 *   smo_create    <- per-call solver construction the reference repeats inside every f / Grad_f:
 *                    FWD_Solve_Build_Lin   FWD_Solve_SH23.py:279-332, FWD_Solve_KDyn.py:362-450,
 *                    LBVP build            FWD_Solve_SHB23.py:563-587;  GEN_BUFFER (snapshot stack)
 *                    FWD_Solve_SH23.py:238-272, FWD_Solve_KDyn.py:319-355, FWD_Solve_SHB23.py:270-314
 *   smo_forward   <- FWD_Solve_IVP_Lin      FWD_Solve_SH23.py:409-545, FWD_Solve_KDyn.py:529-689,
 *                    FWD_Solve_IVP_Discrete FWD_Solve_SHB23.py:525-678          (returns the minimised value -J)
 *   smo_adjoint   <- ADJ_Solve_IVP_Lin      FWD_Solve_SH23.py:598-729 (+Compatib_Cond :552-596),
 *                                           FWD_Solve_KDyn.py:766-1004 (+Compatib_Cond :696-764),
 *                    ADJ_Solve_IVP_Discrete FWD_Solve_SHB23.py:796-920
 *   smo_inner     <- Inner_Prod             FWD_Solve_SH23.py:158-172, Inner_Prod_3 FWD_Solve_KDyn.py:173-181,
 *                    Inner_Prod_Discrete    FWD_Solve_SHB23.py:189-193
 *   SMO_POIS (stratified plane-Poiseuille optimal mixing, "Discrete" formulation): smo_forward <- FWD_Solve_Discrete
 *                    FWD_Solve_Poiseuille.py:777-1155, smo_adjoint <- ADJ_Solve_Discrete :1320-1659, smo_inner <- Inner_Prod_Discrete
 *                    :282-299, smo_transform <- transform / transformInverse / transformAdjoint / transformInverseAdjoint :44-89
 *
 * Conventions
 *   - every function returns SMO_OK (0) or an error code; smo_last_error() gives the message (thread local).
 *   - vectors are the reference's flat float64 grid vectors (Vec_to_Field / Field_to_Vec layouts):
 *       SH23 : 1 component, G = 2*npts values on the scale-2 Fourier grid
 *       SHB23: 1 component, npts values on the ascending Gauss-Chebyshev grid
 *       KDYN : 2 components (B0, U), each 3*G^3 (x,y,z parts concatenated, each C-ordered [x][y][z]), G = 3*npts/2
 *       POIS : 1 component [u.flatten(), w.flatten()] of the (Nx, Nz) grids, z fastest (Field_to_Vec, FWD_Solve_Poiseuille.py:160-207)
 *   - "_dev" entry points take pointers into the HBM of the context's device (no copies); the plain ones take
 *     caller-owned host buffers and stage them through context-owned device buffers.
 *   - all entry points are synchronous (return after the context's stream has drained).
 *   - smo_adjoint replays the snapshot stack filled by the last smo_forward of the same context and returns
 *     SMO_ERR_STATE if there was none (the reference's hidden contract).
 *   - `batch` > 1 runs that many independent 1-D problems per call (SH23/SHB23 only): vectors are then
 *     [batch][len] and J / inner results are arrays of `batch` doubles.
 *   - one host thread per context; contexts are independent.
 *   - There is NO CPU fallback: without a usable HIP device smo_create fails with SMO_ERR_NO_DEVICE.
 */
#ifndef SMO_H_
#define SMO_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smo_ctx smo_ctx;

enum { SMO_OK = 0, SMO_ERR_ARG = 1, SMO_ERR_NO_DEVICE = 2, SMO_ERR_HIP = 3, SMO_ERR_STATE = 4, SMO_ERR_NOMEM = 5,
       SMO_ERR_UNSUPPORTED = 6 };

enum { SMO_SH23 = 1, SMO_SHB23 = 2, SMO_KDYN = 3, SMO_POIS = 4 }; /* smo_config.kind */
enum { SMO_POIS_KINETIC = 0, SMO_POIS_MIXNORM = 1,              /* smo_config.cost (POIS): the reference's switch s (FWD_Solve_Poiseuille.py:1761-1762), Discrete */
       SMO_POIS_CNTS_KINETIC = 2, SMO_POIS_CNTS_MIXNORM = 3 };   /* ... and the same two in the "Continuous" formulation (:614-775, :1161-1318) */
enum { SMO_COST_FINAL = 0, SMO_COST_INTEGRATED = 1 };           /* smo_config.cost (KDYN) */
enum { SMO_SHB_DISCRETE = 0, SMO_SHB_CONTINUOUS = 1 };          /* smo_config.cost (SHB23): formulation, FWD_Solve_SHB23.py:213-217 */
enum { SMO_ADJ_DISCRETE = 0, SMO_ADJ_CONTINUOUS = 1 };          /* `adjoint_type` argument */

typedef struct smo_config {
    int    kind;        /* SMO_SH23 | SMO_SHB23 | SMO_KDYN | SMO_POIS */
    int    npts;        /* Npts as the reference's Generate_IC receives it (SH23 256, SHB23 512, KDYN 128).  Like the reference, every kind takes the size it
                           is handed: SH23 ANY Npts >= 4; SHB23 ANY grid length in [4, 1024] (Continuous: Npts modes on the 2*Npts grid, Npts <= 512);
                           KDYN ANY even Npts >= 6 (FWD_Solve_SH23.py:279-332, FWD_Solve_SHB23.py:196-217, FWD_Solve_KDyn.py:362-450).  Tuned kernels
                           (compile-time transform lengths with factors 2, 3, 5, 7) exist for SH23 2^k, 3*2^k, 5*2^k, 7*2^k, 15*2^k in [16, 1024]; SHB23 2^k,
                           3*2^k in [64, 1024]; KDYN 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 60, 64, 72, 80, 96, 100, 112, 120, 128, 144, 160, 192, 200,
                           224, 240, 256, 320.  Every other size (other prime factors, odd lengths) runs the same pipeline with the transform length a
                           run-time value (csrc/kdyn_any.hpp, sh23_*_any, the SHB23 kernels' NH = 0 form): 1.4-2.6x slower per grid point.  Outside these
                           ranges: SMO_ERR_UNSUPPORTED */
    double x0, x1;      /* interval of every axis: SH23 (0,12pi), SHB23 (-20,20), KDYN (0,2pi) */
    double dt;          /* time step */
    int    n_iters;     /* N_ITERS (the forward solve executes N_ITERS+1 steps for SH23/KDYN, like the reference) */
    double param;       /* SH23/SHB23: a (-0.3 / -0.1);  KDYN: Rm */
    int    cost;        /* KDYN: SMO_COST_FINAL | SMO_COST_INTEGRATED.  SHB23: SMO_SHB_DISCRETE (npts grid values = npts modes; snapshots
                           are grid states) | SMO_SHB_CONTINUOUS (npts modes, vectors hold the 2*npts values of the scale-2 grid,
                           snapshots are the npts T-coefficients; smo_adjoint must then be called with adjoint_type Continuous) */
    int    batch;       /* independent problems per call (>=1; KDYN: 1) */
    int    device;      /* HIP device ordinal */
    /* slab decomposition of the 3-D case (one process per GPU; the exchange itself is done by the host layer):  */
    int    rank;        /* this process' slab index   (0 when world == 1) */
    int    world;       /* number of slabs            (1 = single GPU) */
    /* windowed checkpointing of the adjoint stack (KDYN, single GPU): keep every `ckpt`-th snapshot and recompute the
     * states in between, one window at a time, during the adjoint sweep (+ (ckpt-1)/ckpt forward steps per adjoint step).
     * 1 = keep everything (the reference's N_SUB_ITERS = N_ITERS); 0 = smallest interval whose stack fits the free HBM. */
    int    ckpt;
    /* second resolution / further parameters (POIS only; zero elsewhere): npts = Nx and npts2 = Nz as the solver receives them (Discrete:
     * already scaled by 3/2, FWD_Solve_Poiseuille.py:1752-1755; Continuous: the mode counts, vectors then live on the 3/2 grid and
     * smo_adjoint must be called with adjoint_type Continuous); x0,x1 = the x interval (0, 4 pi), z is (-1, 1); param = Reynolds,
     * param2 = Richardson, param3 = Prandtl (0 -> 1), param4 = delta of the base density profile (0 -> 0.25); cost = s. */
    int    npts2;
    double param2, param3, param4;
} smo_config;

/* ---- life cycle ------------------------------------------------------------------------------------------- */
int         smo_create(const smo_config* cfg, smo_ctx** out);
/* ONE host process driving several GPUs of a node ("one host process driving the devices").  The reference's optimiser is a
 * single Python process; under `mpiexec -np P` every rank runs it redundantly and Dedalus moves the data (README.md:83, FWD_Solve_KDyn.py:118-134).
 * A multi-device context slab-decomposes a KDYN problem over `ndev` devices (cfg->rank / world stay 0 / 1, cfg->device is ignored; ndev must
 * divide npts/2 and 3*npts/2) behind the single-GPU calling sequence:
 *   smo_forward / smo_adjoint / smo_inner  take the reference's FULL flat host vectors [3][G][G][G] and return the full gradient: the library
 *       scatters / gathers the z slabs itself (the allgather of Field_to_Vec, FWD_Solve_KDyn.py:118-123, disappears);
 *   smo_forward_dev / smo_adjoint_dev      take ndev slab pointers per component instead: X[c * ndev + i] = component c's slab
 *       [3][G][G][G/ndev] in the HBM of dev_ids[i];
 * inside, one persistent worker thread per device runs the same in-library time loop as with one process per GPU, and a transpose is every
 * device pulling its blocks out of its peers' send buffers (peer access over xGMI), ordered by HIP events — ONE gather kernel per exchange that
 * reads all peers at once, or one hipMemcpyPeerAsync per peer (SMO_PEER_COPY=kernel|memcpy; smo_comm_get key 3 reports which is in use:
 * the kernel between ranks that share a device, the copy calls between distinct devices until the kernel has been verified on a multi-GPU
 * node) — no RCCL, no launcher, nothing for the caller to set up.  The same device may be listed more than once (tests on a one-GPU box).
 * Not available on such a context: smo_snapshot_read, smo_set_stream (SMO_ERR_UNSUPPORTED: every rank runs on a private stream of its own
 * device); smo_inner_dev is replaced by smo_inner_slabs; smo_timing_* report device 0's kernels.  smo_inner with full HOST vectors scatters
 * both vectors again on every call — an optimiser should keep its vectors distributed (one slab per device, smo_vec_*) and call
 * smo_inner_slabs (spheremanopt_amd/devvec.py: MultiDeviceVector). */
int         smo_create_multi(const smo_config* cfg, int ndev, const int* dev_ids, smo_ctx** out);
void        smo_destroy(smo_ctx* ctx);
const char* smo_last_error(void);
const char* smo_version(void);
int         smo_device_count(int* count);                      /* never initialises a device */

/* ---- geometry --------------------------------------------------------------------------------------------- */
int smo_ncomp(const smo_ctx* ctx);                              /* number of norm-constrained vectors (1 or 2) */
int smo_vec_len(const smo_ctx* ctx, size_t* len);               /* doubles per component (per batch member) */
int smo_stack_bytes(const smo_ctx* ctx, size_t* bytes);         /* HBM held by the snapshot stack(s) */
/* key 0: checkpoint interval actually in use (1 = every snapshot kept);  key 1: KDYN: bytes of the y-side stack kept by the forward
 * solve so that the adjoint skips the z/y passes of every snapshot (0 if not in use); SHB23: how many calls fell back from the
 * multi-workgroup cluster to one workgroup per problem because a cluster all-gather timed out (a busy GPU);  key 2 (KDYN): number of
 * solves replayed from a captured HIP graph (small grids on one GPU: the whole forward solve / adjoint sweep is one graph launch); SHB23:
 * workgroups that co-operate on one problem (1 = no cluster: batch > 1, fewer than 256 modes, SMO_SHB_CLUSTER=0 or after a time-out);
 * key 3 (KDYN): layout of the y-transformed work fields, 0 = planes [c][kx][y][z], 1 = z-block major [c][z/8][kx][y][z%8];
 * key 4 (KDYN): milliseconds of HOST time the last smo_forward + smo_adjoint spent issuing work (entry of the call until everything is enqueued,
 * minus the waits for other ranks in host rendezvous; a multi-device context reports its slowest worker) — to be set against the kernels' time;
 * key 5 (KDYN): first snapshot index of the dense tail of the checkpoint schedule (every state from there on is kept), -1 = uniform interval. */
int smo_get(const smo_ctx* ctx, int key, double* value);

/* ---- the three callbacks, host buffers ---------------------------------------------------------------------- */
int smo_forward(smo_ctx* ctx, const double* const* X, double* J);
int smo_adjoint(smo_ctx* ctx, const double* const* X, int adjoint_type, double* const* grad);
int smo_inner(smo_ctx* ctx, const double* x, const double* y, double* out);

/* ---- the three callbacks, device-resident buffers ------------------------------------------------------------ */
int smo_forward_dev(smo_ctx* ctx, const double* const* X_dev, double* J_host);
int smo_adjoint_dev(smo_ctx* ctx, const double* const* X_dev, int adjoint_type, double* const* grad_dev);
int smo_inner_dev(smo_ctx* ctx, const double* x_dev, const double* y_dev, double* out_host);
/* <x,y> of device-resident vectors given slab by slab: x_slabs[i] / y_slabs[i] = the slab in the HBM of the i-th device of a multi-device
 * context (smo_create_multi; the layout smo_forward_dev takes per component); a plain context has one slab, the vector itself. */
int smo_inner_slabs(smo_ctx* ctx, const double* const* x_slabs, const double* const* y_slabs, double* out_host);

/* ---- device-resident vectors for the caller's own vector algebra ------------------------------------------------
 * The reference's optimiser does X + alpha*d, coeff*X, -1.*g + beta*t and deepcopy on full-size NumPy vectors
 * (Sphere_Grad_Descent.py:284, 296-298, 625-690, 755-756, 813) and the callbacks then move them over PCIe on every call.  With
 * these entry points the vectors stay in HBM (spheremanopt_amd/devvec.py wraps them in a class with +, -, scalar *, deepcopy) and
 * go to smo_forward_dev / smo_adjoint_dev / smo_inner_dev as they are.
 *   - buffers come from a per-device pool (smo_vec_free returns them to it; smo_vec_pool_release gives the memory back);
 *   - smo_vec_axpby: out[i] = fl( fl(a*x[i]) + fl(b*y[i]) ), y == NULL: out[i] = fl(a*x[i]); products and sum are rounded
 *     separately (no fused multiply-add), i.e. bit for bit what NumPy computes for a*x + b*y; out may alias x or y;
 *   - all of them are synchronous — and they run on a stream of the pool's own: the operands must be COMPLETE when the call is made
 *     (results of smo_*_dev calls are: those return after their stream has drained; work a caller has enqueued itself on another
 *     stream, e.g. the one handed to smo_set_stream, must be synchronised first);
 *   - smo_vec_axpby takes any 8-byte-aligned device pointers (pool buffers are 256-byte aligned and use 16-byte accesses; a view at an
 *     odd element offset of a caller's own buffer runs the same arithmetic one element per lane); anything else: SMO_ERR_ARG. */
int smo_vec_alloc(int device, size_t n, double** out_dev);
int smo_vec_free(int device, double* dev);
int smo_vec_pool_release(int device);
int smo_vec_pool_bytes(int device, size_t* live, size_t* pooled);
int smo_vec_upload(int device, double* dev, const double* host, size_t n);
int smo_vec_download(int device, const double* dev, double* host, size_t n);
int smo_vec_axpby(int device, size_t n, double a, const double* x_dev, double b, const double* y_dev, double* out_dev);
/* page-locked host memory for the host-buffer entry points (smo_forward / smo_adjoint / smo_inner copy at PCIe rate from / to it;
 * from pageable memory the runtime stages through its own bounce buffers) */
int smo_host_alloc(size_t bytes, void** out);
int smo_host_free(void* p);

/* ---- introspection used by the parity tests and the benchmark ------------------------------------------------ */
/* Copy snapshot `index` (0..n_iters) of batch member `b` to the host in the reference's GEN_BUFFER element order:
 * SH23 complex128[Nc]; SHB23 float64[N]; KDYN complex128[3][a][m][m]; POIS complex128[3][Nx/2][Nz] (u_fwd, w_fwd, b_fwd of the
 * non-negative wavenumbers).  `out` receives smo_snapshot_len doubles. */
int smo_snapshot_len(const smo_ctx* ctx, size_t* ndoubles);
int smo_snapshot_read(smo_ctx* ctx, int b, int index, double* out);

/* The reference's standalone transforms, host buffers: which = 0 transform (grid -> coefficients), 1 transformInverse,
 * 2 transformAdjoint (coefficients -> grid), 3 transformInverseAdjoint (grid -> coefficients).
 *   SHB23 (FWD_Solve_SHB23.py:36-67): npts doubles either way.
 *   POIS  (FWD_Solve_Poiseuille.py:44-89): grid = float64[Nx][Nz] of a REAL field, coefficients = complex128[a][Nz] for the
 *          non-negative x wavenumbers n = 0..a-1, a = Nx/2 (the other half of the reference's complex spectrum is the Hermitian mirror). */
/*   KDYN  (the transforms Dedalus performs for field['c'] / field['g'], FWD_Solve_KDyn.py:139-171, 635-637): which = 0: three grid fields
 *          float64[3][G][G][G] -> truncated coefficients complex128[3][a][m][m]; which = 1: the inverse (zero padding, c2r ignores Im of kx = 0). */
int smo_transform(smo_ctx* ctx, int which, const double* in, double* out);

/* ---- slab-decomposed 3-D case: one process per GPU ---------------------------------
 * With smo_config.world > 1 a KDYN context owns the kx-slab [rank*a/world, (rank+1)*a/world) of every coefficient
 * field and the z-slab of every grid field; vectors are then the LOCAL slabs [3][G][G][G/world].  The pencil transpose sits
 * between the z and the y pass, where a field is smallest (16*a*m*G bytes per component), as one all-to-all per direction per
 * step carrying every field of that direction.
 *
 * (1) In-library time loop — what the reference gets from Dedalus' in-library MPI transposes (FWD_Solve_KDyn.py:118-134,
 *     README.md:83 `mpiexec -np 4`): give the context a communicator ONCE, then smo_forward[_dev] / smo_adjoint[_dev] /
 *     smo_inner[_dev] are called collectively by all ranks exactly like their single-GPU forms (J and <x,y> come back reduced).
 *       smo_comm_unique_id   rank 0: 128 opaque bytes (an RCCL unique id) to hand to every rank by any means (MPI, a file,
 *                            torch.distributed, ...);
 *       smo_comm_init        collective: ncclCommInitRank over smo_config.world ranks; the transposes are then grouped
 *                            ncclSend/ncclRecv on HIP streams of the solver (chunk-pipelined with the grid-side kernels), the scalar
 *                            reductions ncclAllReduce;
 *       smo_comm_set_transport  instead of RCCL: caller-provided all-to-all / all-reduce (tests in which ranks share a GPU or run over
 *                            gloo); collective as well.
 *     Both agree, across the ranks, on the checkpoint interval and on whether the grid-side states are kept (every rank takes those
 *     decisions from its own free HBM) before the first exchange could mismatch.
 * (2) Phase-level entry (smo_kdyn_op): the same loop cut into the phases between two exchanges, for a host layer that performs
 *     the transposes itself (spheremanopt_amd/kdyn_slab.py: torch.distributed; kept as the CPU/gloo test harness). */
typedef int (*smo_alltoall_fn)(void* user, const void* src_dev, void* dst_dev, size_t bytes_per_peer, void* hip_stream);
typedef int (*smo_allreduce_fn)(void* user, double* host_values, int n);      /* in place: sum over the ranks */
int smo_comm_unique_id(void* id128);
int smo_comm_init(smo_ctx* ctx, const void* id128);
int smo_comm_set_transport(smo_ctx* ctx, smo_alltoall_fn all_to_all, smo_allreduce_fn all_reduce_sum, void* user);
/* (all three arguments NULL: the null transport — exchanges and reductions do nothing.  For profiling one rank's share of a W-way decomposition
 * on a single GPU through the real in-library loop, tools/prof_slab_geometry.py; the results of such a solve are meaningless.) */
/* key 0: pipelined z chunks per exchange; 1: field-group exchanges per forward+adjoint step pair; 2: 1 if the transport is RCCL;
 * multi-device contexts: 3: how a transpose pulls (2 = gather kernel, 1 = hipMemcpyPeerAsync calls), 4: host rendezvous passed so far */
int smo_comm_get(const smo_ctx* ctx, int key, double* value);
/* File of the librccl this library bound for smo_comm_unique_id / smo_comm_init (dladdr of its ncclGetUniqueId), "" if none can be
 * loaded.  A copy already in the process (PyTorch bundles one) is reused, else the system library is loaded; the environment variable
 * SMO_RCCL_LIB (a path or soname) replaces that search.  What the reference gets from `mpiexec` picking ONE MPI library for every rank
 * (README.md:83): a process holding two RCCL copies would otherwise not know which one carries the transposes. */
const char* smo_comm_library(void);
/* If smo_comm_init / smo_comm_set_transport fail after the transport exists (the ranks chose different checkpoint intervals from their
 * free HBM, ...), the transport is dropped again: the context is back in its "no communicator" state (smo_forward: SMO_ERR_STATE)
 * and the call may be repeated. */

/* The exchange buffers hold [chunk][peer][field group][3][a/world][m][G/world/chunks] complex128; peer blocks are contiguous.
 *   z-side buffer: my kx, peer = z block; written by the inverse z pass, read by the forward z pass
 *   y-side buffer: peer = kx block, my z; read by the inverse y pass, written by the forward y pass
 * Phases only enqueue work on the context's stream (see smo_set_stream); SMO_KD_ENERGY / SMO_KD_SYNC synchronise. */
enum {
    SMO_KD_SET_BUFFERS = 0,   /* p0 = z-side, p1 = y-side exchange buffer, each 2*elems complex128 (elems: SMO_KD_EXCHANGE_ELEMS);
                                 layout [peer][field group][3][a/W][m][G/W]: z side = my kx, peer's z block; y side = peer's kx, my z  */
    SMO_KD_EXCHANGE_ELEMS = 1,/* out <- complex128 elements of ONE field group summed over all peers (= 3*(a/W)*m*G)                */
    SMO_KD_G2C_A = 2,         /* p0 = local grid vector [3][G][G][G/W]: x, y passes grid -> spectrum           [then exchange y->z] */
    SMO_KD_G2C_C = 3,         /* z pass + truncation; i0 = 0: store as snapshot 0 (B0), 1: store as scratch (U^)                     */
    SMO_KD_C2G_A = 4,         /* i0 = 0: dt*alpha*G^ (discrete grad B), 1: G^/scratch, 2: nu^ ; z pass         [then exchange z->y] */
    SMO_KD_C2G_B = 5,         /* y, x passes spectrum -> grid; p0 = output local grid vector, NULL = the context's U field           */
    SMO_KD_FWD_A = 6,         /* i0 = step n: z inverse pass of snapshot n                                  [exchange z->y, 1 group] */
    SMO_KD_FWD_B = 7,         /* i0 = n: y pass, fused x pass (c2r, U x B, r2c), y pass                     [exchange y->z, 1 group] */
    SMO_KD_FWD_C = 8,         /* i0 = n: z forward pass + curl, projection, CNAB1 update -> snapshot n+1                             */
    SMO_KD_ENERGY = 9,        /* i0 = n: out <- this slab's share of <B_n,B_n> (synchronises)                                        */
    SMO_KD_ADJ_INIT = 10,     /* i0 = adjoint_type: terminal condition from snapshot N                                               */
    SMO_KD_ADJ_A = 11,        /* i0 = snapshot index: z inverse pass of curl(G^) [and of B^_i0]       [exchange z->y, 1 or 2 groups:
                                 1 when smo_get(ctx, 1) > 0 (grid-side states kept by the forward solve) and i0 < n_iters, else 2]   */
    SMO_KD_ADJ_B = 12,        /* i0 = snapshot index: y pass(es), fused x pass (two cross products; the second one, (curl G) x B_i0, is
                                 added to a running sum that stays on the grid side), y pass of the first [exchange y->z, 1 group]    */
    SMO_KD_ADJ_C = 13,        /* i0 = snapshot index: z forward pass + G^ update                                                     */
    SMO_KD_SYNC = 14,         /* wait for the context's stream                                                                     */
    SMO_KD_SET_CHUNKS = 15,   /* i0 = K: cut the local z slab into K equal chunks.  The grid-side phases (G2C_A, C2G_B, FWD_B, ADJ_B, NU_B) then
                                 take the chunk index in i1 and the exchange buffers become [chunk][peer][field group][3][a/W][m][G/W/K]:
                                 every chunk is one contiguous all-to-all, so the host layer can overlap the exchange of one chunk with
                                 the grid work on another.  K = 1 (default) is the layout described above.                            */
    SMO_KD_NU_B = 16,         /* after the last adjoint step; i1 = chunk: y pass of the running sum                 [exchange y->z, 1 group] */
    SMO_KD_NU_C = 17          /* z forward pass of it, projection, factor -dt  ->  nu^ (= the reference's nu recursion: it acts as the
                                 identity on the solenoidal sums it carries, so the transforms are applied once instead of per step)   */
};
int smo_kdyn_op(smo_ctx* ctx, int op, int i0, int i1, void* p0, void* p1, double* out);

/* Enqueue all further work of the context on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream) so
 * that collectives issued by the host layer on that stream are ordered with the kernels without host synchronisation.
 * SMO_ERR_UNSUPPORTED on a multi-device context (smo_create_multi). */
int smo_set_stream(smo_ctx* ctx, void* hip_stream);

/* HIP-event timing of the kernels launched by the context (measured on the context's stream).
 * smo_timing_enable(ctx, on) resets the accumulators; on = 0 off, 1 every class, 2 + k only class k, k < 64 (two event records per
 * launch cost ~2 us on the GPU, so the benchmark times only the dominant class inside its timed region); smo_timing_get returns, for kernel class `k`
 * (0 <= k < smo_timing_classes), its name, number of launches, total milliseconds and the ALGORITHMIC bytes
 * one launch moves (DESIGN.md section "kernels"), so  achieved GB/s = bytes * launches / ms / 1e6. */
int         smo_timing_enable(smo_ctx* ctx, int on);
/* time exactly the classes whose bit is set in `class_mask` (bit k = class k; 0 = off); resets the accumulators.  The multi-GPU benchmark
 * times its dominant kernel class AND the "slab_exchange(all-to-all)" class (events around every grouped send/recv, on the stream it runs on). */
int         smo_timing_select(smo_ctx* ctx, unsigned long long class_mask);
/* of the selected classes, time every `every`-th launch only (default 1 = all): the averages become those of a uniform sample of the launches,
 * the event records (two per timed launch, each a marker in the queue that keeps the next kernel from overlapping the previous one's tail)
 * cost 1/every as much.  Stays in force until changed; smo_timing_get's `launches` counts the TIMED launches. */
int         smo_timing_stride(smo_ctx* ctx, int every);
int         smo_timing_classes(const smo_ctx* ctx);
int         smo_timing_get(smo_ctx* ctx, int k, const char** name, long long* launches, double* total_ms,
                           double* bytes_per_launch);
/* Compulsory HBM bytes of one launch of class `k` AS FUSED (every input read once, every output written once) —
 * the denominator of the roofline fraction.  The algorithmic figure of smo_timing_get also prices the axis passes the fusion removed
 * and is therefore >= this one. */
int         smo_timing_hbm_bytes(smo_ctx* ctx, int k, double* bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SMO_H_ */
