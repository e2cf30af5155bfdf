// multigpu.hip -- the multi-GPU entry points of libbohip (included by bohip.hip; SURVEY.md section 8-B2 "multi-GPU variant
// takes a device list" and 8-E1).  Two forms of the same design:
//
//   bohip_mgp_*            ONE process, G devices: one replica of the model per device (x, y and the hyper-parameters are
//                          broadcast, every device factors redundantly -- 192 KB in beats 36 MB of L across xGMI), the
//                          candidate columns are cut into contiguous shards, every device scores its shards on its own
//                          stream (one host worker thread per device issues the launches), ONE ncclAllGather of the
//                          16-byte (value, GLOBAL index) records over a communicator made by ncclCommInitAll, then the
//                          same (value desc, index asc) reduction kernel on every device.
//   bohip_gp_comm_* /      one process PER device (torch.distributed.run, Distributed.jl, MPI): the caller moves a
//   bohip_gp_*_sharded     ncclUniqueId between its ranks by whatever transport it has, every rank attaches a communicator to
//                          its handle (ncclCommInitRank) and scores its shard; exchange and reduction as above, on the
//                          handle's stream.
//
// The reference has no counterpart (it is single-threaded, src/acquisition.jl:54-68 runs the restarts one after the
// other); what is replaced is that loop's arg-max over the restarts: first maximum wins (:62), hence (value desc, index
// asc) on GLOBAL column numbers, which makes the G-device winner identical to the one-device winner bit for bit.
// RCCL has no MAXLOC reduction, so the "all-reduce of the per-GPU arg-max" is an all-gather of G records (128 B at G = 8:
// latency-bound, nowhere near the 153 GB/s of an xGMI link) followed by a local reduce.
#include <rccl/rccl.h>   // types and prototypes only: the library itself is bound at first use, see rccl_api()
#include <dlfcn.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

// RCCL is bound with dlopen at the first multi-GPU call instead of a DT_NEEDED entry: a process holds ONE librccl.so.1
// (PyTorch-ROCm ships its own copy under the same SONAME; whichever the process loaded first is the one dlopen returns,
// so a torch process shares torch's copy and a Julia process gets /opt/rocm's), single-GPU users never load it, and
// the library's load order cannot disturb the host framework (preloading torch's copy ahead of torch made the process
// abort in a static destructor at exit).
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("BOHIP_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            api.error = dlerror();
        }
        if (!api.lib) return;
        bool ok = true;
#define BIND(field, sym) ok = ((api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, #sym))) != nullptr) && ok
        BIND(GetVersion, ncclGetVersion); BIND(GetUniqueId, ncclGetUniqueId); BIND(CommInitRank, ncclCommInitRank);
        BIND(CommInitAll, ncclCommInitAll); BIND(CommDestroy, ncclCommDestroy); BIND(AllGather, ncclAllGather); BIND(CommCount, ncclCommCount);
        BIND(GroupStart, ncclGroupStart); BIND(GroupEnd, ncclGroupEnd); BIND(GetErrorString, ncclGetErrorString);
#undef BIND
        if (!ok) { api.error = "librccl lacks a required symbol"; api.lib = nullptr; }
    });
    return api.lib ? &api : nullptr;
}
#define RCCL_OR_FAIL(R)                                                                                              \
    RcclApi* R = rccl_api();                                                                                         \
    if (!R) return fail(BOHIP_E_COMM, "RCCL (librccl.so.1) could not be loaded: " + rccl_api_error())
static std::string rccl_api_error() {
    static RcclApi* dummy = rccl_api();
    (void)dummy;
    return "set BOHIP_RCCL_LIB or add /opt/rocm/lib to the loader path";
}

#define NCCLCHK(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess)                                                                            \
            return fail(BOHIP_E_COMM, std::string(#expr) + ": " + rccl_api()->GetErrorString(r_) + " (" __FILE__  \
                                          ":" + std::to_string(__LINE__) + ")");                          \
    } while (0)

// contiguous column range of shard s out of G (the first R % G shards hold one column more); same rule as
// bayesianoptimization.jl_amd/dist.py:shard_bounds
static inline int64_t shard_lo(int64_t R, int64_t G, int64_t s) {
    const int64_t base = R / G, rem = R % G;
    return s * base + std::min(s, rem);
}

// ---- one persistent host thread per device: HIP launches are asynchronous, but the per-call host work (argument
// marshalling, the staged H2D of the candidate shard, the blocking pivot check at the end of a refit) is not, and issued
// from one thread it would serialise over the devices.
struct DevWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<int()> job;
    bool has_job = false, quit = false, done = true;
    int rc = 0;
    std::string err;
    int device = 0;
    void loop() {
        hipSetDevice(device);
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = std::move(job);
                has_job = false;
            }
            const int r = j();
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                err = r != 0 ? g_err : std::string();
                done = true;
            }
            cv_done.notify_one();
        }
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_one();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return done; });
        return rc;
    }
};

struct bohip_mgp {
    int nd = 0, spd = 1, d = 0;
    bool threads = true;
    std::vector<int> devs;
    std::vector<bohip_gp*> h;
    std::vector<ncclComm_t> comm;
    std::vector<Best*> dsend, drecv;   // per device: [spd][S] own records, [nd * spd][S] gathered records
    Best* hfinal = nullptr;            // pinned, device-visible: [nd][S] reduced records (every device writes its own copy)
    int64_t rec_cap = 0;               // S capacity of the record buffers
    std::vector<double*> dcand;        // resident candidate shards (bohip_mgp_set_candidates)
    std::vector<int64_t> cand_cap;
    int64_t R_res = 0;
    std::vector<DevWorker*> workers;
    int64_t exchanges = 0;
    // round 5: pinned staging of the candidates, one buffer per device (hipMemcpyAsync from pageable memory is staged by the runtime on the
    // calling thread; from pinned memory the copy engines of nd devices really run side by side)
    std::vector<double*> hstage;
    std::vector<size_t> hstage_cap;
};
// host candidates -> device i: through that device's pinned staging buffer (filled on the device's worker thread), then one async copy
static int mgp_stage_h2d(bohip_mgp* m, int i, double* dst, const double* src, size_t n_doubles, hipStream_t st) {
    if (n_doubles == 0) return 0;
    if (m->hstage_cap[i] < n_doubles) {
        if (m->hstage[i]) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipHostFree(m->hstage[i])); m->hstage[i] = nullptr; m->hstage_cap[i] = 0; }
        const size_t cap = n_doubles + n_doubles / 2 + 64;
        HIPCHK(hipHostMalloc((void**)&m->hstage[i], cap * 8, hipHostMallocPortable));
        m->hstage_cap[i] = cap;
    } else {
        HIPCHK(hipStreamSynchronize(st));   // (an earlier copy out of this buffer must have left it; the calls of this library are synchronous anyway)
    }
    std::memcpy(m->hstage[i], src, n_doubles * 8);
    HIPCHK(hipMemcpyAsync(dst, m->hstage[i], n_doubles * 8, hipMemcpyHostToDevice, st));
    return 0;
}

// run fn(i) for every device: on the device's worker thread, or inline (BOHIP_MGP_THREADS=0 / one device)
static int mgp_for_each(bohip_mgp* m, const std::function<int(int)>& fn) {
    int rc = 0;
    if (!m->threads) {
        for (int i = 0; i < m->nd; ++i) {
            if (hipSetDevice(m->devs[i]) != hipSuccess) return fail(BOHIP_E_HIP, "hipSetDevice failed");
            const int r = fn(i);
            if (r != 0 && rc == 0) rc = r;
        }
        return rc;
    }
    for (int i = 0; i < m->nd; ++i) m->workers[i]->post([&fn, i] { return fn(i); });
    std::string err;
    for (int i = 0; i < m->nd; ++i) {
        const int r = m->workers[i]->wait();
        if (r != 0 && rc == 0) { rc = r; err = m->workers[i]->err; }
    }
    if (rc != 0) g_err = err;
    return rc;
}

static int mgp_ensure_records(bohip_mgp* m, int64_t S) {
    if (m->rec_cap >= S) return 0;
    for (int i = 0; i < m->nd; ++i) {
        HIPCHK(hipSetDevice(m->devs[i]));
        if (m->dsend[i]) HIPCHK(hipFree(m->dsend[i]));
        if (m->drecv[i]) HIPCHK(hipFree(m->drecv[i]));
        m->dsend[i] = m->drecv[i] = nullptr;
        HIPCHK(hipMalloc(&m->dsend[i], (size_t)m->spd * S * sizeof(Best)));
        HIPCHK(hipMalloc(&m->drecv[i], (size_t)m->nd * m->spd * S * sizeof(Best)));
    }
    if (m->hfinal) HIPCHK(hipHostFree(m->hfinal));
    m->hfinal = nullptr;
    // every device's reduce kernel writes its copy into this block: portable (pinned for every device context, whichever was current
    // here) and mapped (device-visible through the host address under unified addressing) -- requested explicitly, not left to the default
    HIPCHK(hipSetDevice(m->devs[0]));
    HIPCHK(hipHostMalloc((void**)&m->hfinal, (size_t)m->nd * S * sizeof(Best), hipHostMallocPortable | hipHostMallocMapped));
    m->rec_cap = S;
    return 0;
}

// The exchange step: every device contributes its spd * S records, receives all nd * spd * S, reduces slot by slot and
// writes its copy of the S winners into pinned host memory.  Returns after every device's stream has drained, with the
// copies compared: a disagreement would mean the collective delivered different data to different ranks.
static int mgp_exchange(bohip_mgp* m, int64_t S, Best* out) {
    RCCL_OR_FAIL(R);
    NCCLCHK(R->GroupStart());
    for (int i = 0; i < m->nd; ++i) {
        const ncclResult_t r = R->AllGather(m->dsend[i], m->drecv[i], (size_t)(2 * m->spd * S), ncclInt64, m->comm[i], m->h[i]->stream);
        if (r != ncclSuccess) { R->GroupEnd(); return fail(BOHIP_E_COMM, std::string("ncclAllGather: ") + R->GetErrorString(r)); }
    }
    NCCLCHK(R->GroupEnd());
    // reduce + drain per device on the device's own worker thread: nd streams are waited for side by side, not one after another
    CHK(mgp_for_each(m, [&](int i) -> int {
        HIPCHK(hipSetDevice(m->devs[i]));
        hipLaunchKernelGGL(k_reduce_records, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, m->h[i]->stream, m->drecv[i],
                           m->nd * m->spd, (int)S, m->hfinal + (int64_t)i * S);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(m->h[i]->stream));
        return 0;
    }));
    // every device reduced the same gathered records with the same kernel: its copy can only differ if the collective delivered
    // different data to different ranks.  Checked on request (BOHIP_MGP_VERIFY=1, and in the tests), not on every step.
    static const bool verify = [] { const char* e = getenv("BOHIP_MGP_VERIFY"); return e != nullptr && atoi(e) != 0; }();
    if (verify)
        for (int i = 1; i < m->nd; ++i)
            if (std::memcmp(m->hfinal, m->hfinal + (int64_t)i * S, (size_t)S * sizeof(Best)) != 0)
                return fail(BOHIP_E_COMM, "devices disagree on the reduced arg-max records");
    std::memcpy(out, m->hfinal, (size_t)S * sizeof(Best));
    m->exchanges++;
    return 0;
}

static int mgp_acq_params(int acq_id, const double* acq_params, AcqParams* ap) {
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (acq_id != BOHIP_ACQ_MAXMEAN && !acq_params) return fail(BOHIP_E_ARG, "acq_params required for this acquisition");
    (void)ap;
    return 0;
}

// batch_hint for the duration of a sharded call, restored on every exit path (an early error return used to leave the replica --
// which callers can borrow through bohip_mgp_handle -- choosing its scoring path for a stale batch size)
struct HintGuard {
    bohip_gp* g;
    HintGuard(bohip_gp* g_, int64_t R) : g(g_) { g->batch_hint = R; }
    ~HintGuard() { g->batch_hint = 0; }
};
// score the shards of device i: candidates [lo_dev, hi_dev) are at dXs_dev (device pointer, d contiguous doubles each)
static int mgp_score_device(bohip_mgp* m, int i, int acq_id, const double* acq_params, const double* dXs_dev, int64_t R,
                            bool want_scores) {
    bohip_gp* g = m->h[i];
    const int64_t G = (int64_t)m->nd * m->spd;
    const int64_t lo_dev = shard_lo(R, G, (int64_t)i * m->spd);
    HintGuard hint(g, R);   // every shard takes the summation schedule of the whole set: G-device scores == 1-device scores bit for bit
    for (int ls = 0; ls < m->spd; ++ls) {
        const int64_t s = (int64_t)i * m->spd + ls, lo = shard_lo(R, G, s), hi = shard_lo(R, G, s + 1);
        Best* rec = m->dsend[i] + ls;
        if (hi > lo) {
            CHK(score_core(g, acq_id, acq_params, dXs_dev + (lo - lo_dev) * g->d, hi - lo, nullptr, nullptr,
                           want_scores ? g->dscore + (lo - lo_dev) : nullptr, rec, lo));
        } else {
            static const Best none{-INFINITY, -1};
            HIPCHK(hipMemcpyAsync(rec, &none, sizeof(Best), hipMemcpyHostToDevice, g->stream));
        }
    }
    return 0;
}

extern "C" {

int bohip_mgp_create(int64_t d, int64_t capacity, int kernel_id, const int* devices, int n_devices, int shards_per_device,
                     bohip_mgp** out) {
    if (!out) return fail(BOHIP_E_ARG, "out is null");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(BOHIP_E_ARG, "device list must hold 1..64 ordinals");
    if (shards_per_device < 1 || shards_per_device > 64) return fail(BOHIP_E_ARG, "shards_per_device must be in [1, 64]");
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(BOHIP_E_ARG, "device list holds an ordinal twice (use shards_per_device for logical shards)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(BOHIP_E_NODEVICE, "no HIP device visible; libbohip has no CPU fallback");
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return fail(BOHIP_E_ARG, "device ordinal out of range");
    bohip_mgp* m = new bohip_mgp();
    m->nd = n_devices; m->spd = shards_per_device; m->d = (int)d;
    m->devs.assign(devices, devices + n_devices);
    m->h.assign(n_devices, nullptr);
    m->comm.assign(n_devices, nullptr);
    m->dsend.assign(n_devices, nullptr);
    m->drecv.assign(n_devices, nullptr);
    m->dcand.assign(n_devices, nullptr);
    m->cand_cap.assign(n_devices, 0);
    m->hstage.assign(n_devices, nullptr);
    m->hstage_cap.assign(n_devices, 0);
    m->threads = n_devices > 1;
    if (const char* e = getenv("BOHIP_MGP_THREADS")) m->threads = atoi(e) != 0;
    int rc = 0;
    for (int i = 0; i < n_devices && rc == 0; ++i) rc = bohip_gp_create(d, capacity, kernel_id, devices[i], &m->h[i]);
    if (rc == 0) {
        RcclApi* R = rccl_api();
        if (!R) rc = fail(BOHIP_E_COMM, "RCCL (librccl.so.1) could not be loaded: " + rccl_api_error());
        else {
            const ncclResult_t r = R->CommInitAll(m->comm.data(), n_devices, devices);
            if (r != ncclSuccess) rc = fail(BOHIP_E_COMM, std::string("ncclCommInitAll: ") + R->GetErrorString(r));
        }
    }
    if (rc == 0) rc = mgp_ensure_records(m, 1);
    if (rc == 0 && m->threads) {
        for (int i = 0; i < n_devices; ++i) {
            DevWorker* w = new DevWorker();
            w->device = devices[i];
            w->th = std::thread([w] { w->loop(); });
            m->workers.push_back(w);
        }
    }
    if (rc != 0) {
        const std::string keep = g_err;
        bohip_mgp_destroy(m);
        g_err = keep;
        return rc;
    }
    *out = m;
    return 0;
}

void bohip_mgp_destroy(bohip_mgp* m) {
    if (!m) return;
    for (DevWorker* w : m->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
        }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (int i = 0; i < m->nd; ++i) {
        if (!m->h[i]) continue;   // creation failed before this device was touched
        hipSetDevice(m->devs[i]);
        if (m->h[i]->stream) hipStreamSynchronize(m->h[i]->stream);
        if (m->comm[i] && rccl_api()) rccl_api()->CommDestroy(m->comm[i]);
        if (m->dsend[i]) hipFree(m->dsend[i]);
        if (m->drecv[i]) hipFree(m->drecv[i]);
        if (m->dcand[i]) hipFree(m->dcand[i]);
        if (m->h[i]) bohip_gp_destroy(m->h[i]);
    }
    if (m->hfinal) hipHostFree(m->hfinal);
    for (double* p : m->hstage) if (p) hipHostFree(p);
    (void)hipGetLastError();   // a failed call above must not surface as the "last error" of an unrelated later launch
    delete m;
}

int bohip_mgp_set_hyper(bohip_mgp* m, const double* loglen, double logsig, double lognoise, double mean_const) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    for (int i = 0; i < m->nd; ++i) CHK(bohip_gp_set_hyper(m->h[i], loglen, logsig, lognoise, mean_const));
    return 0;
}

int bohip_mgp_append(bohip_mgp* m, const double* X, const double* y, int64_t p) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    std::vector<int64_t> n_old((size_t)m->nd);
    for (int i = 0; i < m->nd; ++i) n_old[i] = m->h[i]->n;
    const int rc = mgp_for_each(m, [&](int i) { return bohip_gp_append(m->h[i], X, y, p); });
    if (rc != 0 && rc != BOHIP_E_NOTPD) {
        // a replica failed (allocation, HIP error): the others may have taken the observations.  Roll every replica back to the
        // common state before the call -- observations dropped, factor marked stale (rebuilt at the next use) -- so that the
        // handle stays consistent; BOHIP_E_NOTPD is the same on every replica (same data, same arithmetic) and keeps the
        // single-handle semantics: the observations stay, the factor is stale.
        const std::string keep = g_err;
        for (int i = 0; i < m->nd; ++i) {
            bohip_gp* g = m->h[i];
            if (g->n != n_old[i]) {
                g->n = n_old[i];
                g->hX.resize((size_t)g->n * g->d);
                g->hy.resize((size_t)g->n);
            }
            g->stale = true;
            g->n_factored = 0;
        }
        g_err = keep;
    }
    return rc;
}

int bohip_mgp_refit(bohip_mgp* m) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    return mgp_for_each(m, [&](int i) { return bohip_gp_refit(m->h[i]); });
}

int bohip_mgp_score(bohip_mgp* m, int acq_id, const double* acq_params, const double* Xs, int64_t R, double* score,
                    bohip_best* best) {
    if (!m || R < 0 || (R > 0 && !Xs) || !best) return fail(BOHIP_E_ARG, "bad arguments");
    CHK(mgp_acq_params(acq_id, acq_params, nullptr));
    if (R == 0) { best->val = -INFINITY; best->idx = -1; return 0; }
    CHK(mgp_ensure_records(m, 1));
    const int64_t G = (int64_t)m->nd * m->spd;
    CHK(mgp_for_each(m, [&](int i) -> int {
        bohip_gp* g = m->h[i];
        HIPCHK(hipSetDevice(g->device));
        t_reset(g);
        const int64_t lo = shard_lo(R, G, (int64_t)i * m->spd), hi = shard_lo(R, G, (int64_t)(i + 1) * m->spd), Rd = hi - lo;
        if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
        CHK(ensure_xs(g, std::max<int64_t>(Rd, 1)));
        CHK(ensure_score_scratch(g, std::max<int64_t>(Rd, 1)));
        CHK(mgp_stage_h2d(m, i, g->dXs, Xs + lo * g->d, (size_t)Rd * g->d, g->stream));
        CHK(mgp_score_device(m, i, acq_id, acq_params, g->dXs, R, score != nullptr));
        if (score && Rd > 0) HIPCHK(hipMemcpyAsync(score + lo, g->dscore, (size_t)Rd * 8, hipMemcpyDeviceToHost, g->stream));
        return 0;
    }));
    return mgp_exchange(m, 1, reinterpret_cast<Best*>(best));
}

int bohip_mgp_set_candidates(bohip_mgp* m, const double* Xs, int64_t R) {
    if (!m || R < 0 || (R > 0 && !Xs)) return fail(BOHIP_E_ARG, "bad arguments");
    const int64_t G = (int64_t)m->nd * m->spd;
    CHK(mgp_for_each(m, [&](int i) -> int {
        bohip_gp* g = m->h[i];
        HIPCHK(hipSetDevice(g->device));
        const int64_t lo = shard_lo(R, G, (int64_t)i * m->spd), hi = shard_lo(R, G, (int64_t)(i + 1) * m->spd), Rd = hi - lo;
        if (m->cand_cap[i] < Rd * g->d) {
            if (m->dcand[i]) HIPCHK(hipFree(m->dcand[i]));
            m->dcand[i] = nullptr; m->cand_cap[i] = 0;
            HIPCHK(hipMalloc(&m->dcand[i], std::max<size_t>(8, (size_t)Rd * g->d * 8)));
            m->cand_cap[i] = Rd * g->d;
        }
        CHK(mgp_stage_h2d(m, i, m->dcand[i], Xs + lo * g->d, (size_t)Rd * g->d, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        return 0;
    }));
    m->R_res = R;
    return 0;
}

int bohip_mgp_score_resident(bohip_mgp* m, int acq_id, const double* acq_params, bohip_best* best) {
    if (!m || !best) return fail(BOHIP_E_ARG, "bad arguments");
    CHK(mgp_acq_params(acq_id, acq_params, nullptr));
    const int64_t R = m->R_res;
    if (R == 0) { best->val = -INFINITY; best->idx = -1; return 0; }
    CHK(mgp_ensure_records(m, 1));
    const int64_t G = (int64_t)m->nd * m->spd;
    CHK(mgp_for_each(m, [&](int i) -> int {
        bohip_gp* g = m->h[i];
        HIPCHK(hipSetDevice(g->device));
        t_reset(g);
        if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
        const int64_t Rd = shard_lo(R, G, (int64_t)(i + 1) * m->spd) - shard_lo(R, G, (int64_t)i * m->spd);
        CHK(ensure_score_scratch(g, std::max<int64_t>(Rd, 1)));
        return mgp_score_device(m, i, acq_id, acq_params, m->dcand[i], R, false);
    }));
    return mgp_exchange(m, 1, reinterpret_cast<Best*>(best));
}

int bohip_mgp_thompson(bohip_mgp* m, const double* Xs, int64_t R, int64_t S, uint64_t seed, bohip_best* best) {
    if (!m || R <= 0 || S <= 0 || !Xs || !best) return fail(BOHIP_E_ARG, "bad arguments");
    CHK(mgp_ensure_records(m, S));
    const int64_t G = (int64_t)m->nd * m->spd;
    CHK(mgp_for_each(m, [&](int i) -> int {
        bohip_gp* g = m->h[i];
        HIPCHK(hipSetDevice(g->device));
        t_reset(g);
        if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
        const int64_t lo_dev = shard_lo(R, G, (int64_t)i * m->spd), hi_dev = shard_lo(R, G, (int64_t)(i + 1) * m->spd);
        const int64_t Rd = hi_dev - lo_dev;
        CHK(ensure_xs(g, std::max<int64_t>(Rd, 1)));
        CHK(ensure_score_scratch(g, std::max<int64_t>(Rd, 1)));
        CHK(mgp_stage_h2d(m, i, g->dXs, Xs + lo_dev * g->d, (size_t)Rd * g->d, g->stream));
        HintGuard hint(g, R);
        for (int ls = 0; ls < m->spd; ++ls) {
            const int64_t s = (int64_t)i * m->spd + ls, lo = shard_lo(R, G, s), hi = shard_lo(R, G, s + 1);
            Best* rec = m->dsend[i] + (int64_t)ls * S;
            if (hi > lo)
                CHK(score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, g->dXs + (lo - lo_dev) * g->d, hi - lo, g->dmu, g->dvar, nullptr, nullptr));
            // hi == lo: R = 0 candidates -> k_thompson writes (-Inf, -1) for every draw
            hipLaunchKernelGGL(k_thompson, dim3((unsigned)S), dim3(256), 0, g->stream, g->dmu, g->dvar, hi - lo, seed, lo, rec,
                               (long long)lo);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }));
    return mgp_exchange(m, S, reinterpret_cast<Best*>(best));
}

int bohip_mgp_acquire_max(bohip_mgp* m, int acq_id, const double* acq_params, const double* lowerbounds,
                          const double* upperbounds, const double* starts, int64_t R, int64_t maxeval, double ftol_rel,
                          double xtol_abs, double* x_out, double* f_out, bohip_best* best, double* best_x,
                          int64_t* evals_out) {
    if (!m || !lowerbounds || !upperbounds || R < 0 || (R > 0 && !starts) || !best) return fail(BOHIP_E_ARG, "bad arguments");
    CHK(mgp_acq_params(acq_id, acq_params, nullptr));
    if (evals_out) *evals_out = 0;
    if (R == 0) { best->val = -INFINITY; best->idx = -1; return 0; }
    CHK(mgp_ensure_records(m, 1));
    const int d = m->d;
    const int64_t G = (int64_t)m->nd * m->spd;
    std::vector<Best> rec((size_t)m->nd);
    std::vector<double> bx((size_t)m->nd * d);
    std::vector<int64_t> ev((size_t)m->nd, 0);
    CHK(mgp_for_each(m, [&](int i) -> int {
        // the spd logical shards of a device advance as ONE lock-step batch: the ascent treats its start columns independently
        bohip_gp* g = m->h[i];
        const int64_t lo = shard_lo(R, G, (int64_t)i * m->spd), hi = shard_lo(R, G, (int64_t)(i + 1) * m->spd), Rd = hi - lo;
        bohip_best b{-INFINITY, -1};
        HintGuard hint(g, R);
        const int rc = Rd > 0 ? bohip_gp_acquire_max(g, acq_id, acq_params, lowerbounds, upperbounds, starts + lo * d, Rd, maxeval,
                                                     ftol_rel, xtol_abs, x_out ? x_out + lo * d : nullptr,
                                                     f_out ? f_out + lo : nullptr, &b, bx.data() + (size_t)i * d, &ev[i])
                              : 0;
        if (rc != 0) return rc;
        rec[i].val = b.val;
        rec[i].idx = b.idx >= 0 ? b.idx + lo : -1;
        HIPCHK(hipSetDevice(g->device));
        for (int ls = 0; ls < m->spd; ++ls) {   // the device's record sits in its first slot; the others are empty
            const Best none{-INFINITY, -1};
            HIPCHK(hipMemcpyAsync(m->dsend[i] + ls, ls == 0 ? &rec[i] : &none, sizeof(Best), hipMemcpyHostToDevice, g->stream));
        }
        HIPCHK(hipStreamSynchronize(g->stream));
        return 0;
    }));
    CHK(mgp_exchange(m, 1, reinterpret_cast<Best*>(best)));
    int64_t evals = 0;
    for (int i = 0; i < m->nd; ++i) evals = std::max(evals, ev[i]);
    if (evals_out) *evals_out = evals;
    if (best_x) {
        for (int k = 0; k < d; ++k) best_x[k] = lowerbounds[k];   // reference src/acquisition.jl:56
        for (int i = 0; i < m->nd; ++i)
            if (best->idx >= 0 && rec[i].idx == best->idx) std::memcpy(best_x, bx.data() + (size_t)i * d, (size_t)d * 8);
    }
    return 0;
}

int bohip_mgp_set_maxtime(bohip_mgp* m, double seconds) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    for (int i = 0; i < m->nd; ++i) CHK(bohip_gp_set_maxtime(m->h[i], seconds));
    return 0;
}
int bohip_mgp_set_ascent_stop(bohip_mgp* m, double ftol_abs, double xtol_rel, double stopval) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    for (int i = 0; i < m->nd; ++i) CHK(bohip_gp_set_ascent_stop(m->h[i], ftol_abs, xtol_rel, stopval));
    return 0;
}
int bohip_mgp_set_jitter(bohip_mgp* m, double rel, int max_tries) {
    if (!m) return fail(BOHIP_E_ARG, "null handle");
    for (int i = 0; i < m->nd; ++i) CHK(bohip_gp_set_jitter(m->h[i], rel, max_tries));
    return 0;
}
bohip_gp* bohip_mgp_handle(bohip_mgp* m, int i) { return (m && i >= 0 && i < m->nd) ? m->h[i] : nullptr; }

// BOHIP_INFO_COMM_NRANKS of a handle: what the communicator itself says (ncclCommCount), not what the caller passed to comm_init
static int comm_nranks(const bohip_gp* g, int64_t* value) {
    *value = 0;
    if (!g->comm) return 0;
    int n = 0;
    RCCL_OR_FAIL(R);
    NCCLCHK(R->CommCount((ncclComm_t)g->comm, &n));
    *value = n;
    return 0;
}
static int comm_rccl_version(int64_t* value) {
    int v = 0;
    RCCL_OR_FAIL(R);
    NCCLCHK(R->GetVersion(&v));
    *value = v;
    return 0;
}
int bohip_mgp_info(const bohip_mgp* m, int what, int64_t* value) {
    if (!m || !value) return fail(BOHIP_E_ARG, "null argument");
    switch (what) {
        case BOHIP_MGP_INFO_DEVICES: *value = m->nd; return 0;
        case BOHIP_MGP_INFO_SHARDS: *value = (int64_t)m->nd * m->spd; return 0;
        case BOHIP_MGP_INFO_EXCHANGES: *value = m->exchanges; return 0;
        case BOHIP_MGP_INFO_RCCL_VERSION: {
            int v = 0;
            RCCL_OR_FAIL(R);
            NCCLCHK(R->GetVersion(&v));
            *value = v;
            return 0;
        }
        case BOHIP_MGP_INFO_COMM_NRANKS: {   // ncclCommCount of the communicator the records travel over (0: one device, no communicator)
            *value = 0;
            if (m->comm.empty() || !m->comm[0]) return 0;
            int n = 0;
            RCCL_OR_FAIL(R);
            NCCLCHK(R->CommCount(m->comm[0], &n));
            *value = n;
            return 0;
        }
        default: return fail(BOHIP_E_ARG, "unknown info id");
    }
}

// ---- one process per device ---------------------------------------------------------------------------------------
int bohip_comm_unique_id(void* id, int64_t nbytes) {
    if (!id || nbytes < (int64_t)sizeof(ncclUniqueId)) return fail(BOHIP_E_ARG, "id buffer must hold BOHIP_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    RCCL_OR_FAIL(R);
    NCCLCHK(R->GetUniqueId(&u));
    std::memset(id, 0, (size_t)nbytes);
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

static int comm_ensure_records(bohip_gp* g, int64_t S) {
    if (g->crec_cap >= S) return 0;
    if (g->csend) HIPCHK(hipFree(g->csend));
    if (g->crecv) HIPCHK(hipFree(g->crecv));
    if (g->cfinal) HIPCHK(hipFree(g->cfinal));
    g->csend = g->crecv = g->cfinal = nullptr; g->crec_cap = 0;
    HIPCHK(hipMalloc(&g->csend, (size_t)S * sizeof(Best)));
    HIPCHK(hipMalloc(&g->crecv, (size_t)g->comm_n * S * sizeof(Best)));
    HIPCHK(hipMalloc(&g->cfinal, (size_t)S * sizeof(Best)));
    g->crec_cap = S;
    return 0;
}

int bohip_gp_comm_init(bohip_gp* g, const void* id, int64_t nbytes, int rank, int nranks) {
    if (!g || !id || nbytes < (int64_t)sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(BOHIP_E_ARG, "bad arguments");
    if (g->comm) return fail(BOHIP_E_STATE, "handle already has a communicator");
    HIPCHK(hipSetDevice(g->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    RCCL_OR_FAIL(R);
    NCCLCHK(R->CommInitRank(&c, nranks, u, rank));
    g->comm = c; g->comm_rank = rank; g->comm_n = nranks;
    return comm_ensure_records(g, 1);
}

int bohip_gp_comm_destroy(bohip_gp* g) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    if (!g->comm) return 0;
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->stream));
    if (rccl_api()) rccl_api()->CommDestroy((ncclComm_t)g->comm);
    g->comm = nullptr; g->comm_n = 0; g->comm_rank = 0;
    for (Best** p : {&g->csend, &g->crecv, &g->cfinal})
        if (*p) { hipFree(*p); *p = nullptr; }
    g->crec_cap = 0;
    return 0;
}

int bohip_gp_score_sharded_dev(bohip_gp* g, int acq_id, const double* acq_params, const double* dXs, int64_t R_local,
                               int64_t col_offset, int64_t R_total, double* d_score, bohip_best* best) {
    if (!g || R_local < 0 || (R_local > 0 && !dXs) || !best || col_offset < 0 || R_total < R_local)
        return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (!g->comm) return fail(BOHIP_E_STATE, "no communicator: call bohip_gp_comm_init first");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(comm_ensure_records(g, 1));
    if (R_local > 0) {
        const int64_t hint = g->batch_hint;
        g->batch_hint = R_total;
        const int rc = score_core(g, acq_id, acq_params, dXs, R_local, nullptr, nullptr, d_score, g->csend, col_offset);
        g->batch_hint = hint;
        CHK(rc);
    } else {
        static const Best none{-INFINITY, -1};
        HIPCHK(hipMemcpyAsync(g->csend, &none, sizeof(Best), hipMemcpyHostToDevice, g->stream));
    }
    t_begin(g, "exchange");
    RCCL_OR_FAIL(R);
    NCCLCHK(R->AllGather(g->csend, g->crecv, 2, ncclInt64, (ncclComm_t)g->comm, g->stream));
    g->comm_exchanges++;
    hipLaunchKernelGGL(k_reduce_records, dim3(1), dim3(256), 0, g->stream, g->crecv, g->comm_n, 1, reinterpret_cast<Best*>(best));
    HIPCHK(hipGetLastError());
    t_end(g);
    return 0;
}

int bohip_gp_thompson_sharded(bohip_gp* g, const double* Xs, int64_t R_local, int64_t S, uint64_t seed, int64_t col_offset,
                              int64_t R_total, bohip_best* best) {
    if (!g || R_local < 0 || S <= 0 || (R_local > 0 && !Xs) || !best || col_offset < 0 || R_total < R_local)
        return fail(BOHIP_E_ARG, "bad arguments");
    if (!g->comm) return fail(BOHIP_E_STATE, "no communicator: call bohip_gp_comm_init first");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(comm_ensure_records(g, S));
    if (R_local > 0) {
        CHK(ensure_xs(g, R_local));
        CHK(ensure_score_scratch(g, R_local));
        HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R_local * g->d * 8, hipMemcpyHostToDevice, g->stream));
        const int64_t hint = g->batch_hint;
        g->batch_hint = R_total;
        const int rc = score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, g->dXs, R_local, g->dmu, g->dvar, nullptr, nullptr);
        g->batch_hint = hint;
        CHK(rc);
    }
    hipLaunchKernelGGL(k_thompson, dim3((unsigned)S), dim3(256), 0, g->stream, g->dmu, g->dvar, R_local, seed, col_offset, g->csend,
                       (long long)col_offset);
    HIPCHK(hipGetLastError());
    RCCL_OR_FAIL(R);
    NCCLCHK(R->AllGather(g->csend, g->crecv, (size_t)(2 * S), ncclInt64, (ncclComm_t)g->comm, g->stream));
    g->comm_exchanges++;
    hipLaunchKernelGGL(k_reduce_records, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, g->stream, g->crecv, g->comm_n, (int)S,
                       g->cfinal);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best, g->cfinal, (size_t)S * sizeof(Best), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return 0;
}

}  // extern "C"
