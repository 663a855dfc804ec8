// Measurement probes of the C-ABI (fp64 MFMA issue rate, GEMM kernel ablations, HBM store patterns):
// what scripts/probe*.py and the numbers in DESIGN.md section 4 come from.  Not on the product path.
#include <chrono>

#include "gpmi_ctx.h"

using namespace gpmi;

extern "C" {

// out[2] = shader cycles per MFMA per SIMD
int gpmi_probe_mfma_f64_ex(gpmi_ctx* c, int blocks_per_cu, int nacc, int iters, double* out) {
    if (!c || !out) return fail_arg("gpmi_probe_mfma_f64_ex: null argument");
    if (blocks_per_cu < 1 || blocks_per_cu > 4 || iters < 1) return fail_arg("gpmi_probe_mfma_f64_ex: waves per SIMD must be 1 .. 4");
    if (nacc != 4 && nacc != 8 && nacc != 16) return fail_arg("gpmi_probe_mfma_f64_ex: nacc must be 4, 8 or 16");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->red.ensure(16 * 8));
    hipStream_t s = c->stream;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * blocks_per_cu;        // 4-wave groups in flight (for the flop count below)
    double* sink = c->red.as<double>();
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(sink + 8);
    HIP_TRY(launch_probe_mfma(s, sink, 64, cus, blocks_per_cu, nacc, clk));   // warm-up
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, s));
    HIP_TRY(launch_probe_mfma(s, sink, iters, cus, blocks_per_cu, nacc, clk));
    HIP_TRY(hipEventRecord(b, s));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    const double n_mfma_wave = (double)iters * nacc;
    out[0] = (double)blocks * 4 * n_mfma_wave * 2048.0 / (ms * 1e-3) / 1e12;
    out[1] = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.;      // s_memrealtime ticks at 100 MHz
    out[2] = (double)h[0] / (n_mfma_wave * blocks_per_cu);           // waves per SIMD = blocks per CU
    return GPMI_OK;
}

int gpmi_probe_mfma_f64(gpmi_ctx* c, double* tflops) {
    if (!tflops) return fail_arg("gpmi_probe_mfma_f64: null argument");
    double out[3];
    int rc = gpmi_probe_mfma_f64_ex(c, 2, 16, 2048, out);
    if (rc == GPMI_OK) *tflops = out[0];
    return rc;
}

// Timing of one GEMM launch shape on scratch buffers (results discarded).
// variant: ablation bits (1: no global loads in the K loop, 2: no LDS writes / barriers,
// 4: epilogue without the C read, 8: no epilogue).  out[0] = TFLOP/s over computed tiles,
// out[1] = ms per launch.
// pseudo-random operands in (-1, 1) (variant bit 32 of gpmi_probe_gemm): constant operands keep the matrix pipe's data
// paths from toggling, and a chip that is not at its power limit hides what a power-limited one shows
__global__ void probe_fill_random_kernel(double* A, int64_t ld, int64_t ncols, unsigned long long seed) {
    double* row = A + (int64_t)blockIdx.y * ld;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long x = seed + (unsigned long long)blockIdx.y * 0x9E3779B97F4A7C15ull + (unsigned long long)j * 0xBF58476D1CE4E5B9ull;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        row[j] = (double)(long long)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
}

int gpmi_probe_gemm(gpmi_ctx* c, int64_t M, int64_t N, int64_t K, int lower, int variant, int reps,
                    double* out) {
    if (!c || !out) return fail_arg("gpmi_probe_gemm: null argument");
    if (M <= 0 || N <= 0 || K <= 0 || M % TILE || N % IB || K % 16 || reps < 1)
        return fail_arg("gpmi_probe_gemm: M%128, N%64, K%16 must be 0");
    HIP_TRY(hipSetDevice(c->device));
    Tuning tn = c->tune;                 // this call's private copy carries the ablation bits
    TuneScope tune_scope(&tn);
    hipStream_t s = c->stream;
    const int64_t ldc = N + c->ld_pad, ldk = K + c->ld_pad;
    DevBuf C, A, B;
    int rc = GPMI_OK;
    hipError_t e;
    hipEvent_t ea = nullptr, eb = nullptr;
    do {
        if ((e = C.ensure((size_t)M * ldc * 8)) != hipSuccess || (e = A.ensure((size_t)M * ldk * 8)) != hipSuccess ||
            (e = B.ensure((size_t)N * ldk * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc"); break; }
        (void)hipMemsetAsync(C.p, 0, (size_t)M * ldc * 8, s);
        if (variant & 32) {
            hipLaunchKernelGGL(probe_fill_random_kernel, dim3(8, (unsigned)M), dim3(256), 0, s, A.as<double>(), ldk, K, 0x1234ull);
            hipLaunchKernelGGL(probe_fill_random_kernel, dim3(8, (unsigned)N), dim3(256), 0, s, B.as<double>(), ldk, K, 0x9876ull);
            hipLaunchKernelGGL(probe_fill_random_kernel, dim3(64, (unsigned)M), dim3(256), 0, s, C.as<double>(), ldc, N, 0x5555ull);
            variant &= ~32;
        } else {
            (void)launch_fill_rows(s, A.as<double>(), ldk, M, K, 0.001);
            (void)launch_fill_rows(s, B.as<double>(), ldk, N, K, -0.002);
        }
        GemmArgs g;
        g.C = C.as<double>(); g.A = A.as<double>(); g.B = B.as<double>();
        g.ldc = ldc; g.lda = g.ldb = ldk; g.M = M; g.N = N; g.K = K; g.mode = 0; g.lower = lower; g.diag_off = 0;
        DevBuf stamps;
        if (variant & 16) {
            if ((e = stamps.ensure(4096 * 16 * 8)) != hipSuccess) { rc = fail_runtime(e, "hipMalloc"); break; }
            (void)hipMemsetAsync(stamps.p, 0, 4096 * 16 * 8, s);
            tn.gemm_stamps = stamps.as<unsigned long long>();
        }
        tn.gemm_dbg = variant;
        e = launch_gemm_nt(s, g);
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        (void)hipEventRecord(ea, s);
        for (int r = 0; r < reps && e == hipSuccess; ++r) e = launch_gemm_nt(s, g);
        (void)hipEventRecord(eb, s);
        hipError_t e2 = hipEventSynchronize(eb);
        tn.gemm_dbg = 0;
        if (e != hipSuccess) { rc = fail_runtime(e, "gemm launch"); break; }
        if (e2 != hipSuccess) { rc = fail_runtime(e2, "gemm sync"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ea, eb);
        out[1] = ms / reps;
        out[0] = gemm_nt_flops(g) / (out[1] * 1e-3) / 1e12;
        if (variant & 16) {
            std::vector<unsigned long long> h(4096 * 16);
            (void)hipMemcpy(h.data(), stamps.p, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (size_t i = 0; i < h.size(); i += 4)
                if (h[i + 1]) { for (int q = 0; q < 4; ++q) sum[q] += (double)h[i + q]; ++cnt; }
            if (cnt) fprintf(stderr, "[gemm stamps] waves %d: prologue %.0f  loop %.0f  epilogue-loads %.0f  epilogue-stores %.0f cycles\n",
                             cnt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt);
            tn.gemm_stamps = nullptr;
            stamps.release();
        }
    } while (0);
    tn.gemm_dbg = 0;
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    (void)hipStreamSynchronize(s);
    C.release(); A.release(); B.release();
    return rc;
}

int gpmi_probe_hbm_ex(gpmi_ctx* c, int64_t bytes, int mode, int blocks, double* gbps) {
    if (!c || !gbps || bytes < 4096 || blocks < 1 || mode < 0 || mode > 6) return fail_arg("gpmi_probe_hbm_ex: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->red.ensure(16 * 8));
    hipStream_t s = c->stream;
    DevBuf buf;
    HIP_TRY(buf.ensure((size_t)bytes));
    hipError_t e = launch_probe_write(s, buf.as<double>(), bytes / 8, mode == 4 ? 0 : mode, blocks, c->red.as<double>());
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    const int reps = 3;
    for (int r = 0; r < reps && e == hipSuccess; ++r)
        e = launch_probe_write(s, buf.as<double>(), bytes / 8, mode, blocks, c->red.as<double>());
    (void)hipEventRecord(b, s);
    hipError_t e2 = hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    buf.release();
    if (e != hipSuccess) return fail_runtime(e, "probe kernel");
    if (e2 != hipSuccess) return fail_runtime(e2, "probe sync");
    *gbps = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return GPMI_OK;
}

// What the device reports about itself (hipDeviceProp_t): out[0] compute units, out[1] shader clock (kHz),
// out[2] memory clock (kHz), out[3] memory bus width (bits), out[4] total global memory (bytes), out[5] L2 (bytes),
// out[6] LDS per workgroup (bytes), out[7] wavefront size.  bench.py derives the peaks it prices against from
// these (fp64 matrix: CUs x 4 SIMDs x 32 flop/clk x clock; HBM: 2 x memory clock x bus width / 8).
int gpmi_device_info(gpmi_ctx* c, double* out, int count) {
    if (!c || !out || count < 8) return fail_arg("gpmi_device_info: need 8 output slots");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    out[0] = prop.multiProcessorCount;
    out[1] = prop.clockRate;
    out[2] = prop.memoryClockRate;
    out[3] = prop.memoryBusWidth;
    out[4] = (double)prop.totalGlobalMem;
    out[5] = prop.l2CacheSize;
    out[6] = (double)prop.sharedMemPerBlock;
    out[7] = prop.warpSize;
    return GPMI_OK;
}

// Panel kernels alone on scratch data: kind 0 potrf128 (one 128 x 128 block), kind 1 trsm128 on m rows.
// out_us = median-free mean microseconds per launch over reps; stamps_out (64 entries, may be null) = the
// s_memtime stamps of one further, instrumented launch (layout: panel_mfma.hip).
int gpmi_probe_panel(gpmi_ctx* c, int kind, int64_t m, int reps, double* out_us, uint64_t* stamps_out) {
    if (!c || !out_us || reps < 1 || kind < 0 || kind > 1 || (kind == 1 && (m <= 0 || m % TILE)))
        return fail_arg("gpmi_probe_panel: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    Tuning tn = c->tune;
    TuneScope tune_scope(&tn);
    hipStream_t s = c->stream;
    const int64_t ld = 128 + 32, rows = 128 + (kind == 1 ? m : 0);
    DevBuf A, st, info;
    HIP_TRY(A.ensure((size_t)rows * ld * 8));
    HIP_TRY(st.ensure(64 * 8));
    HIP_TRY(info.ensure(8));
    std::vector<double> h((size_t)rows * ld, 0.0);
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < 128; ++j)
            h[(size_t)(i * ld + j)] = (i < 128) ? ((i == j) ? 2.0 : 0.5 / (1.0 + (double)(i > j ? i - j : j - i)))
                                                : 0.001 * (double)((i * 7 + j * 13) % 97);
    HIP_TRY(hipMemcpy(A.p, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(st.p, 0, 64 * 8));
    double* Ad = A.as<double>();
    DevBuf pristine;
    HIP_TRY(pristine.ensure((size_t)128 * ld * 8));
    HIP_TRY(hipMemcpy(pristine.p, A.p, (size_t)128 * ld * 8, hipMemcpyDeviceToDevice));
    auto restore = [&]() -> hipError_t {     // potrf128 works in place: every launch gets the same SPD block
        return kind == 0 ? hipMemcpyAsync(A.p, pristine.p, (size_t)128 * ld * 8, hipMemcpyDeviceToDevice, s) : hipSuccess;
    };
    auto once = [&]() -> hipError_t {
        return kind == 0 ? launch_potrf128(s, Ad, ld, 0, info.as<int64_t>()) : launch_trsm128(s, Ad, ld, Ad + 128 * ld, ld, m);
    };
    if (kind == 1) HIP_TRY(launch_potrf128(s, Ad, ld, 0, info.as<int64_t>()));
    HIP_TRY(restore());
    HIP_TRY(once());
    hipEvent_t ea, eb, ec;
    HIP_TRY(hipEventCreate(&ea));
    HIP_TRY(hipEventCreate(&eb));
    HIP_TRY(hipEventCreate(&ec));
    HIP_TRY(hipEventRecord(ea, s));
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps && e == hipSuccess; ++r) { e = restore(); if (e == hipSuccess) e = once(); }
    HIP_TRY(hipEventRecord(eb, s));
    for (int r = 0; r < reps && e == hipSuccess; ++r) e = restore();
    HIP_TRY(hipEventRecord(ec, s));
    HIP_TRY(hipEventSynchronize(ec));
    float ms = 0.f, ms_copy = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ea, eb));
    HIP_TRY(hipEventElapsedTime(&ms_copy, eb, ec));
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb); (void)hipEventDestroy(ec);
    if (e != hipSuccess) return fail_runtime(e, "panel kernel");
    *out_us = (ms - ms_copy) * 1e3 / reps;
    HIP_TRY(restore());
    if (stamps_out) {
        tn.panel_stamps = st.as<unsigned long long>();
        HIP_TRY(once());
        tn.panel_stamps = nullptr;
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(stamps_out, st.p, 64 * 8, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipStreamSynchronize(s));
    A.release(); st.release(); info.release(); pristine.release();
    return GPMI_OK;
}

// The give-up path of the one-launch backward solve (solve.hip: trsv_lt_chain_kernel): on a trivially solvable system
// (L = I, n unknowns) the bottom block is deliberately left unsolved, so every wait runs into its bound (wait_ms instead
// of the product's 10 s).  *err_out = the kernel's error word (1 expected), *elapsed_ms = how long the launch took:
// every wave must LEAVE the kernel, with NaN in place of the entries it waited for.
int gpmi_probe_trsv_giveup(gpmi_ctx* c, int64_t n, double wait_ms, int* err_out, double* elapsed_ms, double* x_out) {
    if (!c || !err_out || !elapsed_ms || !x_out) return fail_arg("gpmi_probe_trsv_giveup: null argument");
    if (n < 256 || n % TILE || n > 8192 || !(wait_ms > 0.0) || wait_ms > 5000.0) return fail_arg("gpmi_probe_trsv_giveup: n in 256..8192 (multiple of 128), wait_ms in (0, 5000]");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf L, vs, m, x, err;
    HIP_TRY(L.ensure((size_t)n * n * 8));
    HIP_TRY(vs.ensure((size_t)n * 128 * 8));
    HIP_TRY(m.ensure((size_t)n * 8));
    HIP_TRY(x.ensure((size_t)n * 8));
    HIP_TRY(err.ensure(64));
    HIP_TRY(hipMemsetAsync(L.p, 0, (size_t)n * n * 8, s));
    HIP_TRY(hipMemsetAsync(vs.p, 0, (size_t)n * 128 * 8, s));
    HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
    HIP_TRY(launch_set_identity_diag(s, L.as<double>(), n, n));
    std::vector<double> hv((size_t)n * 128, 0.0), hm((size_t)n, 1.0);
    for (int64_t r = 0; r < n; ++r) hv[(size_t)r * 128 + (size_t)(r % 128)] = 1.0;      // V_kk = I for every block
    HIP_TRY(hipMemcpyAsync(vs.p, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(m.p, hm.data(), hm.size() * 8, hipMemcpyHostToDevice, s));
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    hipError_t e = launch_trsv_lt_chain(s, L.as<double>(), n, vs.as<double>(), m.as<double>(), x.as<double>(), n, err.as<int>(),
                                        1, wait_ms);
    (void)hipEventRecord(b, s);
    hipError_t e2 = hipStreamSynchronize(s);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (e != hipSuccess) return fail_runtime(e, "trsv_lt_chain launch");
    if (e2 != hipSuccess) return fail_runtime(e2, "trsv_lt_chain sync");
    HIP_TRY(hipMemcpy(err_out, err.p, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(x_out, x.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    *elapsed_ms = ms;
    L.release(); vs.release(); m.release(); x.release(); err.release();
    return GPMI_OK;
}

// One resident workgroup that does nothing: `threads` threads and `lds_bytes` of LDS it never touches, asleep for
// `milliseconds` on a stream of its own (high_priority != 0: the device's highest stream priority).  Returns at once; time
// something else (gpmi_probe_gemm) while it is resident to see what a workgroup that merely HOLDS a CU costs the rest of the
// chip (the resident potrf128 chain).
__global__ void probe_sleeper_kernel(unsigned long long ticks, const unsigned long long* poll, int poll_sleep, int fences) {
    const unsigned long long t0 = wall_clock64();
    if (!poll) {
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
        return;
    }
    // the way a flag-chained resident kernel waits: thread 0 re-reads a device flag with agent-scope atomic loads
    // (poll_sleep = argument of s_sleep between two reads); fences & 1: an agent-scope acquire + release pair every
    // ~30 us, as a kernel would issue around each unit of work it is released for; fences & 2: the OTHER waves of the
    // workgroup do not leave but wait at a workgroup barrier for thread 0 (as the waves of a server workgroup wait for
    // their next job) -- round 4: THAT is what costs a concurrent GEMM 14 %
    if (threadIdx.x == 0) {
        unsigned long long acc = 0, last = t0;
        while (wall_clock64() - t0 < ticks) {
            acc += __hip_atomic_load(poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (poll_sleep == 1) __builtin_amdgcn_s_sleep(1);
            else if (poll_sleep <= 2) __builtin_amdgcn_s_sleep(2);
            else if (poll_sleep <= 16) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(64);
            if ((fences & 1) && wall_clock64() - last > 3000) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                last = wall_clock64();
            }
        }
        if (acc == 0x123456789ull) __builtin_trap();
    }
    if (fences & 2) __syncthreads();
}

// the same sleeper holding ~130 vector registers per lane (64 doubles kept live across the sleep): what a resident
// workgroup's REGISTER footprint does to a concurrent GEMM, apart from everything else about it (fences & 4 selects it)
// park: 0 every wave sleeps in a loop; 1 wave 0 sleeps in a loop, the others wait for it at a workgroup barrier; 2 wave 0
// sleeps in a loop, the others poll an LDS word it sets at the end (s_sleep between reads)
}  // extern "C" (a template needs C++ linkage)
template <int ND>
__global__ __launch_bounds__(512) void probe_sleeper_fat_kernel(unsigned long long ticks, double* sink, int park) {
    __shared__ volatile int done;
    double v[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) v[i] = (double)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < ND; ++i) asm volatile("" : "+v"(v[i]));
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (park == 0 || threadIdx.x < 64) {
        while (wall_clock64() - t0 < ticks) {
            __builtin_amdgcn_s_sleep(64);
#pragma unroll
            for (int i = 0; i < ND; ++i) asm volatile("" : "+v"(v[i]));
        }
        if (threadIdx.x == 0) done = 1;
    }
    if (park == 1) __syncthreads();
    if (park == 2 && threadIdx.x >= 64) {
        while (!done) __builtin_amdgcn_s_sleep(64);
    }
    double t = 0.;
#pragma unroll
    for (int i = 0; i < ND; ++i) t += v[i];
    if (t == 123.456) sink[0] = t;
}
extern "C" {

// streams of sleepers that may still be running: destroyed by a LATER call.  hipStreamDestroy WAITS for the stream's work on
// this runtime -- until round 4 this probe destroyed the sleeper's stream right after the launch, i.e. it returned when the
// sleeper was gone, and everything timed "beside" it ran alone (profiles/r03_resident_workgroup_cost.txt is void).
static std::vector<hipStream_t> g_sleeper_streams;

int gpmi_probe_resident(gpmi_ctx* c, int high_priority, int lds_bytes, int threads, double milliseconds, int poll_sleep,
                        int fences) {
    if (!c || lds_bytes < 0 || lds_bytes > 160 * 1024 || threads < 64 || threads > 1024 || threads % 64 ||
        !(milliseconds > 0.0) || milliseconds > 5000.0)
        return fail_arg("gpmi_probe_resident: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    // (poll_sleep < 0: keep the earlier sleepers -- several resident at once, each on a stream of its own)
    if (poll_sleep >= 0) {
        for (hipStream_t old : g_sleeper_streams) (void)hipStreamDestroy(old);   // waits for sleepers of earlier calls
        g_sleeper_streams.clear();
    } else {
        poll_sleep = 0;
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? hi : lo));
    hipError_t e = hipFuncSetAttribute((const void*)probe_sleeper_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) {
        const unsigned long long* flag = nullptr;
        if (poll_sleep > 0) {
            if (c->red.ensure(16 * 8) != hipSuccess) { (void)hipStreamDestroy(st); return fail_arg("gpmi_probe_resident: no scratch"); }
            flag = reinterpret_cast<const unsigned long long*>(c->red.as<double>() + 12);
        }
        if (fences & 4) {
            // fences bits 5..6: register footprint -- 0 ~130 per lane, 1 ~138, 2 ~146, 3 ~106
            const int fat = (fences >> 5) & 3;
            const void* fn = fat == 1 ? (const void*)probe_sleeper_fat_kernel<68> : fat == 2 ? (const void*)probe_sleeper_fat_kernel<72>
                           : fat == 3 ? (const void*)probe_sleeper_fat_kernel<52> : (const void*)probe_sleeper_fat_kernel<64>;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e == hipSuccess && c->red.ensure(16 * 8) != hipSuccess) e = hipErrorOutOfMemory;
            if (e == hipSuccess) {
                const unsigned long long tk = (unsigned long long)(milliseconds * 1e5);
                double* sink = c->red.as<double>();
                const int park = (fences >> 3) & 3;
                if (fat == 1) hipLaunchKernelGGL(probe_sleeper_fat_kernel<68>, dim3(1), dim3((unsigned)threads), (size_t)lds_bytes, st, tk, sink, park);
                else if (fat == 2) hipLaunchKernelGGL(probe_sleeper_fat_kernel<72>, dim3(1), dim3((unsigned)threads), (size_t)lds_bytes, st, tk, sink, park);
                else if (fat == 3) hipLaunchKernelGGL(probe_sleeper_fat_kernel<52>, dim3(1), dim3((unsigned)threads), (size_t)lds_bytes, st, tk, sink, park);
                else hipLaunchKernelGGL(probe_sleeper_fat_kernel<64>, dim3(1), dim3((unsigned)threads), (size_t)lds_bytes, st, tk, sink, park);
            }
        } else {
            hipLaunchKernelGGL(probe_sleeper_kernel, dim3(1), dim3((unsigned)threads), (size_t)lds_bytes, st,
                               (unsigned long long)(milliseconds * 1e5), flag, poll_sleep, fences);      // wall_clock64: 100 MHz
        }
        if (e == hipSuccess) e = hipGetLastError();
    }
    g_sleeper_streams.push_back(st);     // NOT destroyed here: that would wait for the sleeper
    if (e != hipSuccess) return fail_runtime(e, "probe_sleeper launch");
    return GPMI_OK;
}

// A storm of tiny kernels on a stream of its own: `count` launches of a one-wave kernel, `sleep_us` microseconds long
// each (kind 0: it only sleeps; 1: an agent-scope release + acquire fence pair as well; 2: an agent-scope atomic store
// as well).  Returns at once; time something else (gpmi_probe_gemm) meanwhile to see what the KERNEL BOUNDARIES of a
// busy second stream -- the runtime brackets every kernel with cache maintenance -- cost a long-running update GEMM
// (the resident panel chain of round 3 doubled the launches on the panel stream).
__global__ void probe_tiny_kernel(unsigned long long ticks, int kind, unsigned long long* flag) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (kind == 1 && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (kind == 2 && threadIdx.x == 0) __hip_atomic_store(flag, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int gpmi_probe_launch_storm(gpmi_ctx* c, int high_priority, int count, double sleep_us, int kind) {
    if (!c || count < 1 || count > 200000 || sleep_us < 0.0 || sleep_us > 1000.0 || kind < 0 || kind > 2)
        return fail_arg("gpmi_probe_launch_storm: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->red.ensure(16 * 8));
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? hi : lo));
    hipError_t e = hipSuccess;
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->red.as<double>() + 12);
    for (int i = 0; i < count && e == hipSuccess; ++i) {
        hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)(sleep_us * 100.0), kind, flag);
        e = hipGetLastError();
    }
    (void)hipStreamDestroy(st);          // released when the kernels have finished
    if (e != hipSuccess) return fail_runtime(e, "probe_tiny launch");
    return GPMI_OK;
}

// How many streams of a priority really run side by side?  n_high streams at the device's highest priority and n_norm at
// the default one, each handed ONE one-wave kernel that sleeps `milliseconds`; *wall_ms = time until all have finished.
// All concurrent: ~milliseconds; streams that share a hardware queue run one after the other: a multiple of it.  The
// runtime maps streams onto a small pool of hardware queues as they are first used (a second
// DistGP instance's fresh pair of auxiliary streams landed on one queue).
int gpmi_probe_stream_overlap(gpmi_ctx* c, int n_high, int n_norm, double milliseconds, double* wall_ms) {
    if (!c || !wall_ms || n_high < 0 || n_norm < 0 || n_high + n_norm < 1 || n_high + n_norm > 32 || !(milliseconds > 0.0) ||
        milliseconds > 1000.0)
        return fail_arg("gpmi_probe_stream_overlap: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::vector<hipStream_t> st((size_t)(n_high + n_norm), nullptr);
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < st.size() && e == hipSuccess; ++i)
        e = hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (int)i < n_high ? hi : lo);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < st.size() && e == hipSuccess; ++i) {
        hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, st[i], (unsigned long long)(milliseconds * 1e5), 0,
                           (unsigned long long*)nullptr);
        e = hipGetLastError();
    }
    for (size_t i = 0; i < st.size(); ++i)
        if (st[i]) { if (e == hipSuccess) e = hipStreamSynchronize(st[i]); }
    *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (hipStream_t s : st) if (s) (void)hipStreamDestroy(s);
    if (e != hipSuccess) return fail_runtime(e, "gpmi_probe_stream_overlap");
    return GPMI_OK;
}

// The resident potrf128 server itself (panel_mfma.hip, experiment) beside a GEMM and nothing else: starts it on the context's
// server stream with the given mode bits, times gpmi_probe_gemm's launches, stops it.  out as gpmi_probe_gemm.
int gpmi_probe_gemm_beside_server(gpmi_ctx* c, int64_t M, int64_t N, int64_t K, int lower, int variant, int reps, int mode,
                                  double* out) {
    if (!c || !out) return fail_arg("gpmi_probe_gemm_beside_server: null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->pmail.p) {
        HIP_TRY(c->pmail.ensure(sizeof(PotrfMail)));
        HIP_TRY(hipMemset(c->pmail.p, 0, sizeof(PotrfMail)));
        c->pserver.mail = c->pmail.as<PotrfMail>();
    }
    if (!c->sstream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_TRY(hipStreamCreateWithPriority(&c->sstream, hipStreamNonBlocking, hi));
    }
    c->pserver.mode = mode;
    if (mode) HIP_TRY(potrf_server_start(&c->pserver, c->sstream));
    const int rc = gpmi_probe_gemm(c, M, N, K, lower, variant, reps, out);
    if (mode) {
        HIP_TRY(potrf_server_stop(&c->pserver, c->pstream));
        HIP_TRY(hipStreamSynchronize(c->pstream));
        HIP_TRY(hipStreamSynchronize(c->sstream));
    }
    return rc;
}

int gpmi_probe_hbm_write(gpmi_ctx* c, int64_t bytes, double* gbps) {
    return gpmi_probe_hbm_ex(c, bytes, 0, 2048, gbps);
}

}  // extern "C"
