/*
 * bohip.h -- C ABI of libbohip.so: the MI355X (gfx950) implementation of the GP-posterior +
 * acquisition-scoring hot path of jbrea/BayesianOptimization.jl.
 *
 * The reference has no FFI of its own; its seam is Julia dispatch on the model type
 * (reference src/models/gp.jl:2-18).  Each entry point below names the reference generic
 * function / call site it replaces.  A Julia model type `BOHipGPE` (julia/BOHip.jl) forwards
 * those generic functions here with `ccall`; bayesianoptimization.jl_amd/ does the same with ctypes.
 *
 * Conventions
 *   - Float64 / Int64 only.  Matrices are Julia-shaped: d x n COLUMN-major, i.e. every
 *     observation / candidate is d contiguous doubles.
 *   - The library owns all device memory and a host mirror of x, y (Julia reads model.x/.y).
 *     Caller owns every pointer it passes; none is retained after the call returns.
 *   - Every call is blocking (internal stream synchronised before return) unless it is a
 *     *_dev entry point, which enqueues on the handle's stream and returns.
 *   - Return value: 0 = OK, negative = error (see BOHIP_E_*); bohip_last_error() gives text.
 *     Nothing throws or aborts across the ABI.  A handle is not re-entrant; distinct
 *     handles are independent.  One handle = one device; bohip_mgp (below) spans a device list.
 */
#ifndef BOHIP_H
#define BOHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bohip_gp bohip_gp;

/* error codes */
#define BOHIP_OK 0
#define BOHIP_E_ARG (-1)      /* bad argument (null pointer, d mismatch, negative size)          */
#define BOHIP_E_NOTPD (-2)    /* Cholesky met a non-positive pivot; bohip_gp_info(...,PIVOT)      */
#define BOHIP_E_HIP (-3)      /* HIP runtime error (text in bohip_last_error)                    */
#define BOHIP_E_NODEVICE (-4) /* no gfx950 device visible -- there is NO CPU fallback             */
#define BOHIP_E_STATE (-5)    /* call not valid in this state (e.g. predict with 0 observations) */
#define BOHIP_E_UNSUPPORTED (-6)
#define BOHIP_E_COMM (-7)     /* RCCL error, or the devices disagree after the exchange (multi-GPU entry points) */

/* kernel_id: GaussianProcesses.jl kernels the reference's tests/defaults construct
 * (README.md:24, test/acquisition.jl:2, src/BayesianOptimization.jl:259-262) */
#define BOHIP_KERN_SEARD 0
#define BOHIP_KERN_SEISO 1
#define BOHIP_KERN_MAT52ARD 2

/* acq_id + acq_params: the functors of reference src/acquisitionfunctions.jl
 *   EI  :47-50  params {tau}          PI :24-27 params {tau}
 *   UCB :96     params {beta_t}       MI :141   params {sqrt_alpha, gamma_hat}
 *   MAXMEAN :110-111 params {}  (nullable)                                             */
#define BOHIP_ACQ_EI 0
#define BOHIP_ACQ_PI 1
#define BOHIP_ACQ_UCB 2
#define BOHIP_ACQ_MI 3
#define BOHIP_ACQ_MAXMEAN 4

/* 16-byte arg-max record, the unit exchanged between GPUs (one per rank) */
typedef struct bohip_best {
    double val;  /* best score; -Inf if nothing beat -Inf (reference src/acquisition.jl:55) */
    int64_t idx; /* 0-based column of the winner, -1 if none; ties -> smallest index (:62)  */
} bohip_best;

/* ---- lifetime ---------------------------------------------------------------------------
 * Replaces ElasticGPE(d; mean, kernel, logNoise, capacity) (README.md:22-27).  `capacity` is
 * the initial observation capacity; storage grows geometrically beyond it.  device = HIP
 * ordinal.  Fails with BOHIP_E_NODEVICE when no GPU is visible.                           */
int bohip_gp_create(int64_t d, int64_t capacity, int kernel_id, int device, bohip_gp **out);
void bohip_gp_destroy(bohip_gp *gp);

/* ---- hyper-parameters (GP.set_params! at reference src/models/gp.jl:60) ------------------
 * loglen: d log length-scales (SEIso: loglen[0] only), logsig: log signal std, lognoise:
 * logNoise, mean_const: MeanConst beta (0 for MeanZero).  Marks the factor stale; the next
 * append/refit/predict rebuilds K, its Cholesky factor and alpha from the stored x, y.     */
int bohip_gp_set_hyper(bohip_gp *gp, const double *loglen, double logsig, double lognoise, double mean_const);

/* ---- update!(model, x, y) (reference src/models/gp.jl:11; called at
 * src/BayesianOptimization.jl:169,194-196).  X: d x p column-major, y: p.  Incremental:
 * new covariance rows, Cholesky extension (ElasticPDMats append!), alpha.  p == 0 is a no-op
 * that only brings a stale factor up to date.                                             */
int bohip_gp_append(bohip_gp *gp, const double *X, const double *y, int64_t p);

/* ---- GP.fit! / update_target! role (reference src/models/gp.jl:14-16,61): full rebuild of
 * K (SEArd assembly), its Cholesky factor and alpha from the stored observations.          */
int bohip_gp_refit(bohip_gp *gp);

/* ---- dims(model) :9, maxy(model) :10, model.x / model.y field reads ---------------------- */
int bohip_gp_dims(const bohip_gp *gp, int64_t *d, int64_t *n);
int bohip_gp_maxy(const bohip_gp *gp, double *maxy); /* -Inf when n == 0 */
int bohip_gp_get_xy(const bohip_gp *gp, double *X /* d x n, nullable */, double *y /* n, nullable */);
/* log marginal likelihood of the current factor: -0.5 (y-b)'alpha - sum log L_ii - n/2 log 2pi
 * (gp.mll, read by MAP fitting at reference src/models/gp.jl:61-63)                         */
int bohip_gp_mll(bohip_gp *gp, double *mll);
/* mll and its analytic gradient w.r.t. the log hyper-parameters, in the reference's get_params order
 * [logNoise; mean; kernel (loglen..., logsig)] -- the role of GP.update_target_and_dtarget! + gp.dtarget in
 * optimizemodel! (reference src/models/gp.jl:59-64).  d_kern has d + 1 entries (SEArd, Mat52Ard) or 2 (SEIso).
 * d mll/d theta = 1/2 tr((alpha alpha' - cK^-1) dcK/dtheta); cK^-1 = W'W is formed on the device.          */
int bohip_gp_mll_grad(bohip_gp *gp, double *mll, double *d_lognoise, double *d_mean, double *d_kern);

/* ---- mean_var(model, X::Matrix) (reference src/models/gp.jl:8 -> GP.predict_f):
 * Xs d x R column-major; mu, var length R (latent f variance, clamped at 0).                */
int bohip_gp_predict(bohip_gp *gp, const double *Xs, int64_t R, double *mu, double *var);
/* Full posterior covariance (predict_f(gp, X; full_cov = true)): the input of the reference's JOINT draw
 * myrand(model, X::Matrix) = rand(gp, X), src/models/gp.jl:7.  cov is R x R (symmetric, both halves written, diagonal
 * NOT clamped); R is limited to one candidate chunk (>= 1024).  cov = K** - V'V with V'V on the MFMA engine.   */
int bohip_gp_predict_cov(bohip_gp *gp, const double *Xs, int64_t R, double *mu, double *cov);

/* ---- acquisitionfunction(a, model)(X) + the arg-max of acquire_max (reference
 * src/acquisitionfunctions.jl:4-9, src/acquisition.jl:54-68) fused: score all R columns, return
 * per-column scores (nullable) and the best (value, index) under strict '>' from -Inf.       */
int bohip_gp_score(bohip_gp *gp, int acq_id, const double *acq_params, const double *Xs, int64_t R,
                   double *score /* R, nullable */, bohip_best *best);

/* ---- wrap_gradient role (reference src/acquisition.jl:11-17): score and d(score)/dx, the
 * latter d x R column-major.  SEArd / SEIso kernels.                                        */
int bohip_gp_score_grad(bohip_gp *gp, int acq_id, const double *acq_params, const double *Xs, int64_t R,
                        double *score, double *grad);

/* ---- acquire_max(acquisition, model, lowerbounds, upperbounds, restarts) for the gradient-based methods (reference
 * src/acquisition.jl:48-68; nlopt_setup :23-35: method :LD_LBFGS, bounds, maxeval, ftol_rel, xtol_abs).
 * starts: d x R start columns (the reference draws them with latin_hypercube_sampling, src/utils.jl:101-120, from
 * Julia's global RNG, so they are an input).  Every start is refined by a projected L-BFGS ascent on the device, each on its
 * own schedule, one value+gradient pass of the model per evaluation, the state stays on the device.
 * x_out (d x R), f_out (R): best point seen per start (nullable).  best / best_x: the maximiser over the starts under
 * strict '>' (first maximum wins, :58-66); idx = -1 and best_x = lowerbounds if nothing beat -Inf (:55-56).
 * maxeval bounds the number of model passes (NLopt counts per start; here every start consumes one per pass).      */
int bohip_gp_acquire_max(bohip_gp *gp, int acq_id, const double *acq_params, const double *lowerbounds,
                         const double *upperbounds, const double *starts, int64_t R, int64_t maxeval, double ftol_rel,
                         double xtol_abs, double *x_out, double *f_out, bohip_best *best, double *best_x,
                         int64_t *evals_out);
/* NLopt's maxtime option (forwarded by the reference at src/acquisition.jl:24-27): wall-clock budget in seconds of ONE
 * bohip_gp_acquire_max call, checked once per ascent iteration; 0 (default) = unlimited.                          */
int bohip_gp_set_maxtime(bohip_gp *gp, double seconds);
/* The other NLopt stop criteria the reference forwards with setproperty! (src/acquisition.jl:24-27; its own test sets
 * ftol_abs = eps(), test/acquisition.jl:6,9).  With NLopt's meaning, per start point, for a MAXIMISATION:
 *   ftol_abs  stop when an iteration improves the value by <= ftol_abs          (0 = NLopt's default: criterion off)
 *   xtol_rel  stop when |dx_k| <= xtol_rel |x_k| in EVERY coordinate             (0 = off)
 *   stopval   stop as soon as a value >= stopval is reached                      (+Inf = off)
 * They apply to every later bohip_gp_acquire_max call of the handle, beside its ftol_rel / xtol_abs arguments.      */
int bohip_gp_set_ascent_stop(bohip_gp *gp, double ftol_abs, double xtol_rel, double stopval);
/* Jitter escalation when the factorisation fails (the role of GaussianProcesses.jl's make_posdef! behind update!/fit!,
 * src/models/gp.jl:11,16 -- UPSTREAM-UNVERIFIED, so OFF by default and BOHIP_E_NOTPD reports the failing pivot): with
 * max_tries > 0 a failed refit is repeated with rel x mean(diag cK) added to the diagonal, x10 per further try.
 * BOHIP_INFO_JITTER_STEPS tells how many tries the last refit needed.  Env BOHIP_JITTER="rel[,tries]" sets it for every
 * handle of the process.                                                                                          */
int bohip_gp_set_jitter(bohip_gp *gp, double rel, int max_tries);

/* ---- ThompsonSamplingSimple (reference src/acquisitionfunctions.jl:107-108, myrand
 * src/models/gp.jl:6-7) in its batched form: S independent draws mu_j + sigma_j z_sj over the R
 * candidates, arg-max per draw.  z comes from a counter-based generator keyed (seed, s, j + j0)
 * so any shard of the candidate set reproduces the same stream (j0 = global column offset). */
int bohip_gp_thompson(bohip_gp *gp, const double *Xs, int64_t R, int64_t S, uint64_t seed, int64_t j0,
                      bohip_best *best /* S records, idx local to this shard */);
/* the generator itself, exposed so tests and other shards can reproduce z (host side) */
double bohip_thompson_normal(uint64_t seed, int64_t s, int64_t j);

/* ---- :GN_DIRECT_L, the reference's default search for ThompsonSamplingSimple (reference src/acquisition.jl:7-9: restarts 1,
 * maxeval 2000; nlopt_setup :20-38 hands the acquisition to NLopt) and for any acquisition the caller selects it for.
 * NLopt is not vendored in the reference; csrc/direct_l.h restates DIRECT-L with the rules of NLopt's cdirect.c for this variant
 * (MAXIMISATION; a rectangle's size = its longest side, one potentially optimal rectangle per size class, cubes trisected along
 * every side best value first, other rectangles along their first longest side).  Batched: every iteration's new centres are ONE
 * scoring call.
 *   bohip_direct_*  the host bookkeeping as an ask / tell object, for objectives evaluated by the caller (works without a device):
 *     ask   X = d x cap column-major; *n = points of this iteration (0: the search is over).  cap = 0 (X may be NULL) only
 *           reports *n; asking again hands out the same batch until it is told; cap < *n is BOHIP_E_ARG
 *     tell  their values in ask's order (NaN counts as -Inf)
 *     best  first maximum so far, its point, evaluations, iterations
 *   bohip_gp_direct_max  the whole search against a resident model in one call: acq_id = BOHIP_ACQ_* scores through bohip_gp_score;
 *     BOHIP_ACQ_THOMPSON_DRAW evaluates x -> myrand(model, x) (src/acquisitionfunctions.jl:107-108, src/models/gp.jl:6-7): one
 *     posterior draw per evaluated point, mu + sigma z, z = bohip_thompson_normal(seed, 0, e) for the e-th evaluation.
 *     maxtime: NLopt's wall-clock budget in seconds (0 = none), checked once per iteration; stopval: +Inf = off.
 *     best_x = the box centre's image when nothing was finite. */
typedef struct bohip_direct bohip_direct;
int bohip_direct_create(int64_t d, const double *lb, const double *ub, int64_t maxeval, double stopval, double maxtime,
                        bohip_direct **out);
void bohip_direct_destroy(bohip_direct *s);
int bohip_direct_ask(bohip_direct *s, double *X, int64_t cap, int64_t *n);
int bohip_direct_tell(bohip_direct *s, const double *f, int64_t n);
int bohip_direct_best(const bohip_direct *s, double *best_f, double *best_x, int64_t *evaluations, int64_t *iterations);
#define BOHIP_ACQ_THOMPSON_DRAW 5
int bohip_gp_direct_max(bohip_gp *gp, int acq_id, const double *acq_params, const double *lb, const double *ub, int64_t maxeval,
                        double stopval, double maxtime, uint64_t seed, double *best_f, double *best_x, int64_t *evaluations,
                        int64_t *device_calls);

/* ---- device-resident variants (inputs already in HBM; results stay in HBM) ----------------
 * dXs: device pointer, d x R column-major.  d_score (nullable), d_best: device pointers.
 * Enqueued on the handle's stream; no host synchronisation.                                */
int bohip_gp_score_dev(bohip_gp *gp, int acq_id, const double *acq_params, const double *dXs, int64_t R,
                       double *d_score, bohip_best *d_best);
int bohip_gp_predict_dev(bohip_gp *gp, const double *dXs, int64_t R, double *d_mu, double *d_var);
/* stream = hipStream_t (NULL -> the handle's own stream).  Lets a host framework order our
 * kernels with its own work.                                                               */
int bohip_gp_set_stream(bohip_gp *gp, void *stream);
int bohip_gp_synchronize(bohip_gp *gp);

/* ---- introspection for tests / benchmarks ------------------------------------------------ */
/* copies the n x n lower Cholesky factor L (row-major; == Julia's column-major upper U) */
int bohip_gp_get_factor(bohip_gp *gp, double *L);
int bohip_gp_get_alpha(bohip_gp *gp, double *alpha);
#define BOHIP_INFO_PIVOT 0        /* 1-based failing pivot of the last BOHIP_E_NOTPD, else 0     */
#define BOHIP_INFO_CAPACITY 1     /* current observation capacity                                */
#define BOHIP_INFO_REFITS 2       /* number of full refits so far                                */
#define BOHIP_INFO_APPENDS 3      /* number of incremental factor extensions so far              */
#define BOHIP_INFO_CHOL_FORM 4    /* factorisation form of the last refit: 0 launch chain, 1-3 dataflow forms, 4 executor */
#define BOHIP_INFO_CHOL_FALLBACKS 5   /* refits of this handle that timed out on a dependency and were redone launch-chained */
#define BOHIP_INFO_CHOL_ABORT_TILES 6 /* row tiles of the last factorisation that timed out (0: never)  */
#define BOHIP_INFO_JITTER_STEPS 7 /* jitter tries the last refit needed (0: none; see bohip_gp_set_jitter) */
#define BOHIP_INFO_SCORE_LAUNCHES 8 /* K*' chunks (= k_trigemm_sq launches) of the last whole-K scoring pass */
#define BOHIP_INFO_SCORE_CHUNK 9  /* candidates per K*' chunk of the last scoring call (equal-sized multiples of 512) */
#define BOHIP_INFO_KERNEL_CLOCK_MHZ 10 /* core clock the chip sustained under k_trigemm_sq since the previous read (timing enabled:
                                        * a sample of its workgroups counts core-clock cycles against the 100 MHz wall clock), else 0 */
#define BOHIP_INFO_COMM_NRANKS 11 /* ranks of the communicator attached by bohip_gp_comm_init, read back from it (ncclCommCount); 0: none */
#define BOHIP_INFO_COMM_RCCL_VERSION 13 /* ncclGetVersion of the RCCL the library bound (loads it if no multi-GPU call has yet) */
#define BOHIP_INFO_CHOL_LOCK_SKIPS 14    /* refits that took the launch-chained form because another process kept the refit lock for the whole wait */
#define BOHIP_INFO_COMM_EXCHANGES 12 /* RCCL all-gathers this handle has issued so far (bohip_gp_score_sharded_dev / _thompson_sharded) */
int bohip_gp_info(const bohip_gp *gp, int what, int64_t *value);
/* Benchmarks only (bench.py, tools/): the executor form of the factorisation grows W = L^-1 behind the pivot chain in
 * pieces of `blocks` 128-blocks (default 8).  0 switches those queues off -- the factorisation then runs alone and can
 * be timed against its own N^3/3 flops, the inverse follows as a stage of its own.  PROCESS-wide (every handle),
 * returns the previous value.  Same effect as env BOHIP_CHOL_INV_G at load time.  NOT a synchronisation point:
 * call it while no model update is running on another thread (one in flight may use either setting).            */
int bohip_debug_set_chol_inv_g(int blocks);
/* Sharded scoring: candidates are scored by one of three summation schedules chosen by batch size
 * (row-wise, split-K, whole-K MFMA jobs); they agree to ~1e-11 relative but not bit for bit.  A rank that scores a
 * shard of a larger candidate set announces the size of the WHOLE set here, so that every shard takes the schedule
 * the unsharded call would take and the G-GPU scores equal the 1-GPU scores bit for bit.  0 (default) = no hint.  */
int bohip_gp_set_batch_hint(bohip_gp *gp, int64_t total_candidates);
/* per-stage device times (ms, HIP events on the handle's stream) of the LAST call when timing
 * is enabled: names/values for up to `cap` stages; returns the number of stages.
 * on = 1: every stage; on = 2: only the dominant kernel (k_trigemm_sq) is bracketed by events -- two
 * event records per call instead of six (the records themselves cost ~4 us each on the stream);
 * on = 3: as 2, but the records of successive calls accumulate unread until bohip_gp_get_timing, which then
 * returns one entry per recorded launch (a timed loop pays for the two records only, not for reading them). */
int bohip_gp_enable_timing(bohip_gp *gp, int on);
int bohip_gp_get_timing(bohip_gp *gp, const char **names, double *ms, int cap);

/* ======================================================================================================
 * Multi-GPU.  The reference runs its restarts one after the other
 * (src/acquisition.jl:58-66); here the candidate columns are cut into contiguous shards, every GPU scores its shard
 * against its own replica of the model, and the arg-max over the restarts (strict '>', first maximum wins, :62) is
 * ONE RCCL all-gather of the 16-byte (value, GLOBAL column) records followed by the same (value desc, index asc)
 * reduction on every GPU.  RCCL has no MAXLOC, hence all-gather + local reduce; 128 B at 8 GPUs.
 * The winner is bit-identical to the one-GPU result of bohip_gp_score on the whole set.
 * ---- one process, a device list ------------------------------------------------------------------------ */
typedef struct bohip_mgp bohip_mgp;
/* devices: n_devices distinct HIP ordinals (ncclCommInitAll over them).  shards_per_device >= 1 logical shards per
 * device (1 in production; > 1 exercises the G-shard partition and exchange on fewer GPUs).  Shard s of
 * G = n_devices * shards_per_device holds columns [s*floor(R/G) + min(s, R%G), ...) and lives on device s / spd.   */
int bohip_mgp_create(int64_t d, int64_t capacity, int kernel_id, const int *devices, int n_devices,
                     int shards_per_device, bohip_mgp **out);
void bohip_mgp_destroy(bohip_mgp *mgp);
/* replicated model: same meaning as the bohip_gp_* calls, applied to every device (each factors redundantly) */
int bohip_mgp_set_hyper(bohip_mgp *mgp, const double *loglen, double logsig, double lognoise, double mean_const);
int bohip_mgp_append(bohip_mgp *mgp, const double *X, const double *y, int64_t p);
int bohip_mgp_refit(bohip_mgp *mgp);
/* acquisitionfunction(a, model)(X) + the arg-max of acquire_max over ALL devices (src/acquisitionfunctions.jl:4-9,
 * src/acquisition.jl:54-68).  Xs: d x R host, score: R host (nullable), best: global record.                      */
int bohip_mgp_score(bohip_mgp *mgp, int acq_id, const double *acq_params, const double *Xs, int64_t R, double *score,
                    bohip_best *best);
/* the same with the candidate shards already resident in each device's HBM */
int bohip_mgp_set_candidates(bohip_mgp *mgp, const double *Xs, int64_t R);
int bohip_mgp_score_resident(bohip_mgp *mgp, int acq_id, const double *acq_params, bohip_best *best);
/* ThompsonSamplingSimple batched (src/acquisitionfunctions.jl:107-108): S draws over R candidates, one all-gather of
 * S records per shard; best: S global records.                                                                    */
int bohip_mgp_thompson(bohip_mgp *mgp, const double *Xs, int64_t R, int64_t S, uint64_t seed, bohip_best *best);
/* acquire_max with the start columns sharded over the devices (src/acquisition.jl:48-68); arguments as
 * bohip_gp_acquire_max, indices global.                                                                           */
int bohip_mgp_acquire_max(bohip_mgp *mgp, int acq_id, const double *acq_params, const double *lowerbounds,
                          const double *upperbounds, const double *starts, int64_t R, int64_t maxeval, double ftol_rel,
                          double xtol_abs, double *x_out, double *f_out, bohip_best *best, double *best_x,
                          int64_t *evals_out);
/* the per-handle options of bohip_gp_set_maxtime / bohip_gp_set_jitter, applied to every replica */
int bohip_mgp_set_maxtime(bohip_mgp *mgp, double seconds);
int bohip_mgp_set_ascent_stop(bohip_mgp *mgp, double ftol_abs, double xtol_rel, double stopval);
int bohip_mgp_set_jitter(bohip_mgp *mgp, double rel, int max_tries);
bohip_gp *bohip_mgp_handle(bohip_mgp *mgp, int i); /* replica on the i-th listed device (borrowed, for dims/maxy/get_xy/mll...) */
#define BOHIP_MGP_INFO_DEVICES 0
#define BOHIP_MGP_INFO_SHARDS 1
#define BOHIP_MGP_INFO_EXCHANGES 2    /* RCCL all-gathers performed so far */
#define BOHIP_MGP_INFO_RCCL_VERSION 3
#define BOHIP_MGP_INFO_COMM_NRANKS 4   /* ranks of the in-library communicator, read back from it (ncclCommCount); 0: one device */
int bohip_mgp_info(const bohip_mgp *mgp, int what, int64_t *value);

/* ---- one process per device (torch.distributed.run, Distributed.jl, MPI) ----------------------------------
 * Rank 0 calls bohip_comm_unique_id and ships the bytes to the other ranks with the transport it has; every rank
 * attaches a communicator to its handle (ncclCommInitRank on the handle's device: collective, blocks until all
 * ranks arrive).                                                                                              */
#define BOHIP_UNIQUE_ID_BYTES 128
int bohip_comm_unique_id(void *id, int64_t nbytes);
int bohip_gp_comm_init(bohip_gp *gp, const void *id, int64_t nbytes, int rank, int nranks);
int bohip_gp_comm_destroy(bohip_gp *gp);
/* Score this rank's shard (dXs: d x R_local in HBM, columns [col_offset, col_offset + R_local) of R_total), exchange
 * and reduce on the handle's stream: `best` (device or pinned-host pointer) receives the GLOBAL winner, identical on
 * every rank.  Enqueues only (a *_dev entry point); d_score nullable.                                           */
int bohip_gp_score_sharded_dev(bohip_gp *gp, int acq_id, const double *acq_params, const double *dXs, int64_t R_local,
                               int64_t col_offset, int64_t R_total, double *d_score, bohip_best *best);
/* Thompson form: Xs host (d x R_local), best host (S global records, identical on every rank).  Blocking.       */
int bohip_gp_thompson_sharded(bohip_gp *gp, const double *Xs, int64_t R_local, int64_t S, uint64_t seed,
                              int64_t col_offset, int64_t R_total, bohip_best *best);

const char *bohip_last_error(void); /* thread-local */
const char *bohip_version(void);
int bohip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* BOHIP_H */
