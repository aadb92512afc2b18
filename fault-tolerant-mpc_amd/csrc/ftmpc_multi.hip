// ftmpc_multi.hip -- the batch axis across the GPUs of one node, inside ONE process.
//
// Instances are independent (the reference never couples them: one controller object per vehicle,
// ft_mpc/controllers/spiraling_mpc.py:288-317), so the batch is split contiguously: device g owns instances
// [B g / G, B (g+1) / G).  One host thread + one ftmpc_handle + one stream set per device; NO collective and
// nothing on xGMI: every worker reads its slice of the caller's arrays and writes its slice of the caller's
// outputs (that IS the "gather on the host").  Two ways to use it:
//   ftmpc_multi_solve_batch                   host buffers in / out, pinned staging per device (ftmpc_solve_batch)
//   ftmpc_multi_upload / _step / _download    shards stay RESIDENT in each device's HBM between steps
//                                             (what bench.py --gpus N times; Monte-Carlo campaigns)
// included by ftmpc_capi.hip (single translation unit).
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include <sched.h>

struct ftmpc_multi {
    struct Dev {
        int device = 0;
        ftmpc_handle* h = nullptr;
        std::thread th;
        int rc = 0;
        std::string err;
        // resident shard
        int64_t lo = 0, n = 0, cap = 0;
        bool has_warm = false, has_uref = false;
        int64_t xs = 0, us = 0, cap_xr = 0, cap_ur = 0;
        double *x0 = nullptr, *ub = nullptr, *stuck = nullptr, *xref = nullptr, *uref = nullptr, *warm = nullptr, *u0 = nullptr,
               *U = nullptr;
        int32_t *status = nullptr, *iters = nullptr;
        hipStream_t s = nullptr;
        // pinned staging of the resident upload: two halves, so that the host copy of one piece runs under the DMA of the other
        void* pin = nullptr;
        hipEvent_t pin_ev[2] = {nullptr, nullptr};
        int pin_turn = 0;
        int cpus_pinned = 0;     // host cores this device's worker thread is bound to (0: affinity left alone)
    };
    ftmpc_config cfg;
    std::vector<Dev> dev;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(ftmpc_multi::Dev&)> job;
    std::string err;
    int64_t B = 0;
};

namespace {

thread_local std::string g_multi_create_error;

// Binds the calling thread to the host cores nearest `device` (the PCI function's local_cpulist in sysfs, intersected with
// the cores this process may use); when the two do not meet -- containers often grant cores of one socket only -- the
// affinity is left alone.  Returns the number of cores bound to.
int pin_thread_near(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return 0;
    for (char* c = bus; *c; ++c)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return 0;
    char line[4096] = {0};
    const bool got = std::fgets(line, (int)sizeof(line), f) != nullptr;
    std::fclose(f);
    if (!got) return 0;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
    int n = 0;
    for (const char* p = line; *p;) {     // "0-15,128-143"
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, &want);
                ++n;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    if (n == 0 || sched_setaffinity(0, sizeof(want), &want) != 0) return 0;
    return n;
}

void multi_worker(ftmpc_multi* m, int r) {
    ftmpc_multi::Dev& d = m->dev[r];
    (void)hipSetDevice(d.device);
    d.cpus_pinned = pin_thread_near(d.device);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(ftmpc_multi::Dev&)> fn;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_job.wait(lk, [&] { return m->quit || m->gen != seen; });
            if (m->quit) return;
            seen = m->gen;
            fn = m->job;
        }
        const int rc = fn(d);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            d.rc = rc;
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// runs fn on every device's thread at once and waits for all of them; first failing code wins
int multi_run(ftmpc_multi* m, std::function<int(ftmpc_multi::Dev&)> fn) {
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job = std::move(fn);
        m->pending = (int)m->dev.size();
        ++m->gen;
    }
    m->cv_job.notify_all();
    std::unique_lock<std::mutex> lk(m->mu);
    m->cv_done.wait(lk, [&] { return m->pending == 0; });
    for (auto& d : m->dev)
        if (d.rc != FTMPC_OK) {
            m->err = "device " + std::to_string(d.device) + ": " + (d.err.empty() && d.h ? d.h->err : d.err);
            return d.rc;
        }
    return FTMPC_OK;
}

int dev_fail(ftmpc_multi::Dev& d, int code, const std::string& msg) {
    d.err = msg;
    return code;
}

template <typename T>
int dev_grow(ftmpc_multi::Dev& d, T** p, int64_t count) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    if (count <= 0) return FTMPC_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();   // clear the runtime's sticky "last error"
        return dev_fail(d, FTMPC_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    return FTMPC_OK;
}

#define DEV_TRY(d, expr)                                                                          \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return dev_fail((d), FTMPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// host -> device copy of a pageable array through the device's pinned staging halves (PIN_HALF bytes each)
constexpr size_t PIN_HALF = (size_t)8 << 20;
int dev_upload(ftmpc_multi::Dev& d, void* dst, const void* src, size_t bytes) {
    if (!d.pin) {      // all or nothing: a staging buffer without its two recorded events must not survive a failed set-up
        void* pin = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t e = hipHostMalloc(&pin, 2 * PIN_HALF, hipHostMallocDefault);
        for (int t = 0; t < 2 && e == hipSuccess; ++t) e = hipEventCreateWithFlags(&ev[t], hipEventDisableTiming);
        for (int t = 0; t < 2 && e == hipSuccess; ++t) e = hipEventRecord(ev[t], d.s);
        if (e != hipSuccess) {
            for (int t = 0; t < 2; ++t)
                if (ev[t]) (void)hipEventDestroy(ev[t]);
            if (pin) (void)hipHostFree(pin);
            (void)hipGetLastError();
            return dev_fail(d, FTMPC_ERR_HIP, std::string("pinned staging set-up: ") + hipGetErrorString(e));
        }
        d.pin = pin;
        d.pin_ev[0] = ev[0];
        d.pin_ev[1] = ev[1];
    }
    for (size_t off = 0; off < bytes; off += PIN_HALF) {
        const size_t cnt = std::min(PIN_HALF, bytes - off);
        const int t = d.pin_turn;
        d.pin_turn ^= 1;
        char* half = static_cast<char*>(d.pin) + (size_t)t * PIN_HALF;
        DEV_TRY(d, hipEventSynchronize(d.pin_ev[t]));       // the DMA that last read this half is done
        std::memcpy(half, static_cast<const char*>(src) + off, cnt);
        DEV_TRY(d, hipMemcpyAsync(static_cast<char*>(dst) + off, half, cnt, hipMemcpyHostToDevice, d.s));
        DEV_TRY(d, hipEventRecord(d.pin_ev[t], d.s));
    }
    return FTMPC_OK;
}

void shard(int64_t B, int G, int g, int64_t& lo, int64_t& hi) {
    lo = B * g / G;
    hi = B * (g + 1) / G;
}

}  // namespace

extern "C" {

int ftmpc_multi_create(const ftmpc_config* cfg, const int32_t* device_ids, int32_t n_devices, ftmpc_multi** out) {
    if (!cfg || !out) return FTMPC_ERR_ARG;
    *out = nullptr;
    // the ABI guard of ftmpc_create, BEFORE the struct is copied: a caller built against an older, shorter ftmpc_config must be
    // told, not read past its end (nothing beyond struct_size's own offset is touched until it has been checked)
    if (cfg->struct_size != (int32_t)sizeof(ftmpc_config)) {
        g_multi_create_error = "ftmpc_config.struct_size = " + std::to_string(cfg->struct_size) + " but this library was built with sizeof(ftmpc_config) = " +
                               std::to_string(sizeof(ftmpc_config)) + " (fill the struct with ftmpc_default_config of the SAME header)";
        return FTMPC_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_multi_create_error = "no HIP device visible (this library has no CPU fallback)";
        return FTMPC_ERR_NODEVICE;
    }
    if (n_devices <= 0) n_devices = ndev;
    if (n_devices > 64) {
        g_multi_create_error = "more than 64 device slots";
        return FTMPC_ERR_ARG;
    }
    ftmpc_multi* m = new (std::nothrow) ftmpc_multi();
    if (!m) return FTMPC_ERR_ALLOC;
    m->cfg = *cfg;
    m->dev.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        m->dev[i].device = device_ids ? device_ids[i] : i;     // the same ordinal may appear twice (two handles on one GPU)
        if (m->dev[i].device < 0 || m->dev[i].device >= ndev) {
            g_multi_create_error = "device id out of range";
            delete m;
            return FTMPC_ERR_ARG;
        }
    }
    for (int i = 0; i < n_devices; ++i) m->dev[i].th = std::thread(multi_worker, m, i);
    // every handle is created by the thread that will drive it
    const int rc = multi_run(m, [m](ftmpc_multi::Dev& d) -> int {
        ftmpc_config c = m->cfg;
        c.device_id = d.device;
        int rc2 = ftmpc_create(&c, &d.h);
        if (rc2 != FTMPC_OK) return dev_fail(d, rc2, ftmpc_last_error(nullptr));
        DEV_TRY(d, hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking));
        return FTMPC_OK;
    });
    if (rc != FTMPC_OK) {
        g_multi_create_error = m->err;
        ftmpc_multi_destroy(m);
        return rc;
    }
    *out = m;
    return FTMPC_OK;
}

int ftmpc_multi_destroy(ftmpc_multi* m) {
    if (!m) return FTMPC_OK;
    (void)multi_run(m, [](ftmpc_multi::Dev& d) -> int {
        void* ptrs[] = {d.x0, d.ub, d.stuck, d.xref, d.uref, d.warm, d.u0, d.U, d.status, d.iters};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (d.pin) (void)hipHostFree(d.pin);
        for (hipEvent_t e : d.pin_ev)
            if (e) (void)hipEventDestroy(e);
        if (d.s) (void)hipStreamDestroy(d.s);
        if (d.h) (void)ftmpc_destroy(d.h);
        d.h = nullptr;
        return FTMPC_OK;
    });
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->quit = true;
    }
    m->cv_job.notify_all();
    for (auto& d : m->dev)
        if (d.th.joinable()) d.th.join();
    delete m;
    return FTMPC_OK;
}

const char* ftmpc_multi_last_error(const ftmpc_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int32_t ftmpc_multi_device_count(const ftmpc_multi* m) { return m ? (int32_t)m->dev.size() : 0; }

int32_t ftmpc_multi_worker_cpus(const ftmpc_multi* m, int32_t slot) {
    return (m && slot >= 0 && slot < (int32_t)m->dev.size()) ? m->dev[slot].cpus_pinned : 0;
}

int ftmpc_multi_shard_bounds(const ftmpc_multi* m, int64_t B, int32_t slot, int64_t* lo, int64_t* hi) {
    if (!m || !lo || !hi || slot < 0 || slot >= (int32_t)m->dev.size() || B < 0) return FTMPC_ERR_ARG;
    shard(B, (int)m->dev.size(), slot, *lo, *hi);
    return FTMPC_OK;
}

int ftmpc_multi_solve_batch(ftmpc_multi* m, int64_t B, const double* x0, const double* ub, const double* stuck,
                            const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride, double* warmU,
                            double* out_u0, double* out_U, int32_t* status, int32_t* iters) {
    if (!m) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !out_u0) {
        m->err = "null buffer or negative batch";
        return FTMPC_ERR_ARG;
    }
    if (B == 0) return FTMPC_OK;
    const int G = (int)m->dev.size();
    const int N = m->cfg.N, NT = m->cfg.NT;
    const int64_t nw = (int64_t)N * NT;
    return multi_run(m, [=](ftmpc_multi::Dev& d) -> int {
        const int g = (int)(&d - m->dev.data());
        int64_t lo, hi;
        shard(B, G, g, lo, hi);
        if (hi <= lo) return FTMPC_OK;
        return ftmpc_solve_batch(d.h, hi - lo, x0 + lo * 13, ub + lo * NT, stuck + lo * NT, xref + lo * xref_stride, xref_stride,
                                 uref ? uref + lo * uref_stride : nullptr, uref_stride, warmU ? warmU + lo * nw : nullptr,
                                 out_u0 + lo * NT, out_U ? out_U + lo * nw : nullptr, status ? status + lo : nullptr,
                                 iters ? iters + lo : nullptr);
    });
}

int ftmpc_multi_upload(ftmpc_multi* m, int64_t B, const double* x0, const double* ub, const double* stuck, const double* xref,
                       int64_t xref_stride, const double* uref, int64_t uref_stride, const double* warmU) {
    if (!m) return FTMPC_ERR_ARG;
    if (B <= 0 || !x0 || !ub || !stuck || !xref) {
        m->err = "null buffer or empty batch";
        return FTMPC_ERR_ARG;
    }
    const int N = m->cfg.N, NT = m->cfg.NT;
    if ((xref_stride != 0 && xref_stride < 9 * (N + 1)) || (uref && uref_stride != 0 && uref_stride < 6 * (N + 1))) {
        m->err = "reference strides must be 0 or at least one window";
        return FTMPC_ERR_ARG;
    }
    const int G = (int)m->dev.size();
    const int64_t nw = (int64_t)N * NT;
    m->B = B;
    return multi_run(m, [=](ftmpc_multi::Dev& d) -> int {
        const int g = (int)(&d - m->dev.data());
        int64_t lo, hi;
        shard(B, G, g, lo, hi);
        const int64_t n = hi - lo;
        d.lo = lo;
        d.n = n;
        d.has_warm = warmU != nullptr;
        d.has_uref = uref != nullptr;
        d.xs = xref_stride;
        d.us = uref_stride;
        if (n <= 0) return FTMPC_OK;
        int rc;
        if (n > d.cap) {
            d.cap = 0;   // nothing is usable until every buffer below exists again (a failure part-way must not leave a stale capacity)
            if ((rc = dev_grow(d, &d.x0, n * 13)) || (rc = dev_grow(d, &d.ub, n * NT)) || (rc = dev_grow(d, &d.stuck, n * NT)) ||
                (rc = dev_grow(d, &d.warm, n * nw)) || (rc = dev_grow(d, &d.u0, n * NT)) || (rc = dev_grow(d, &d.U, n * nw)) ||
                (rc = dev_grow(d, &d.status, n)) || (rc = dev_grow(d, &d.iters, n)))
                return rc;
            d.cap = n;
        }
        const int64_t nxr = xref_stride == 0 ? 9 * (N + 1) : n * xref_stride;
        const int64_t nur = !uref ? 0 : (uref_stride == 0 ? 6 * (N + 1) : n * uref_stride);
        if (nxr > d.cap_xr) {
            d.cap_xr = 0;
            if ((rc = dev_grow(d, &d.xref, nxr))) return rc;
            d.cap_xr = nxr;
        }
        if (nur > d.cap_ur) {
            d.cap_ur = 0;
            if ((rc = dev_grow(d, &d.uref, nur))) return rc;
            d.cap_ur = nur;
        }
        if ((rc = ftmpc_reserve(d.h, n)) != FTMPC_OK) return dev_fail(d, rc, ftmpc_last_error(d.h));
        // the caller's arrays are pageable: through the pinned halves, so that the DMA engine streams instead of the
        // runtime's own bounce copies
        if ((rc = dev_upload(d, d.x0, x0 + lo * 13, (size_t)n * 13 * 8)) || (rc = dev_upload(d, d.ub, ub + lo * NT, (size_t)n * NT * 8)) ||
            (rc = dev_upload(d, d.stuck, stuck + lo * NT, (size_t)n * NT * 8)) ||
            (rc = dev_upload(d, d.xref, xref + lo * xref_stride, (size_t)nxr * 8)))
            return rc;
        if (uref && (rc = dev_upload(d, d.uref, uref + lo * uref_stride, (size_t)nur * 8))) return rc;
        if (warmU && (rc = dev_upload(d, d.warm, warmU + lo * nw, (size_t)n * nw * 8))) return rc;
        DEV_TRY(d, hipStreamSynchronize(d.s));
        return FTMPC_OK;
    });
}

int ftmpc_multi_step(ftmpc_multi* m, int32_t steps, int32_t keep_U) {
    if (!m || steps < 0) return FTMPC_ERR_ARG;
    if (m->B <= 0) {
        m->err = "nothing uploaded";
        return FTMPC_ERR_ARG;
    }
    return multi_run(m, [=](ftmpc_multi::Dev& d) -> int {
        if (d.n <= 0) return FTMPC_OK;
        for (int s = 0; s < steps; ++s) {
            const int rc = ftmpc_solve_batch_device(d.h, d.n, d.x0, d.ub, d.stuck, d.xref, d.xs, d.has_uref ? d.uref : nullptr, d.us,
                                                    d.has_warm ? d.warm : nullptr, d.u0, keep_U ? d.U : nullptr, d.status, d.iters, d.s);
            if (rc != FTMPC_OK) return dev_fail(d, rc, ftmpc_last_error(d.h));
        }
        DEV_TRY(d, hipStreamSynchronize(d.s));
        return FTMPC_OK;
    });
}

int ftmpc_multi_download(ftmpc_multi* m, double* out_u0, double* out_U, int32_t* status, int32_t* iters) {
    if (!m) return FTMPC_ERR_ARG;
    if (m->B <= 0) {
        m->err = "nothing uploaded";
        return FTMPC_ERR_ARG;
    }
    const int N = m->cfg.N, NT = m->cfg.NT;
    const int64_t nw = (int64_t)N * NT;
    return multi_run(m, [=](ftmpc_multi::Dev& d) -> int {
        if (d.n <= 0) return FTMPC_OK;
        if (out_u0) DEV_TRY(d, hipMemcpyAsync(out_u0 + d.lo * NT, d.u0, (size_t)d.n * NT * 8, hipMemcpyDeviceToHost, d.s));
        if (out_U) DEV_TRY(d, hipMemcpyAsync(out_U + d.lo * nw, d.U, (size_t)d.n * nw * 8, hipMemcpyDeviceToHost, d.s));
        if (status) DEV_TRY(d, hipMemcpyAsync(status + d.lo, d.status, (size_t)d.n * 4, hipMemcpyDeviceToHost, d.s));
        if (iters) DEV_TRY(d, hipMemcpyAsync(iters + d.lo, d.iters, (size_t)d.n * 4, hipMemcpyDeviceToHost, d.s));
        DEV_TRY(d, hipStreamSynchronize(d.s));
        return FTMPC_OK;
    });
}

const char* ftmpc_multi_routed_kernel_name(const ftmpc_multi* m, int32_t slot) {
    return (m && !m->dev.empty()) ? ftmpc_routed_kernel_name(m->dev[0].h, slot) : ftmpc_kernel_name(slot);
}

int ftmpc_multi_set_profiling(ftmpc_multi* m, int32_t enabled) {
    if (!m) return FTMPC_ERR_ARG;
    for (auto& d : m->dev) (void)ftmpc_set_profiling(d.h, enabled);
    return FTMPC_OK;
}

int ftmpc_multi_last_kernel_ms(ftmpc_multi* m, int32_t slot, float* ms, int32_t n_slots) {
    if (!m || !ms || slot < 0 || slot >= (int32_t)m->dev.size()) return FTMPC_ERR_ARG;
    return multi_run(m, [=](ftmpc_multi::Dev& d) -> int {
        if ((int)(&d - m->dev.data()) != slot) return FTMPC_OK;
        const int rc = ftmpc_last_kernel_ms(d.h, ms, n_slots);
        return rc == FTMPC_OK ? rc : dev_fail(d, rc, ftmpc_last_error(d.h));
    });
}

}  // extern "C"
