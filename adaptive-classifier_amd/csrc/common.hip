// Error string, version and device info for libacamd.so.
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace ac {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int g_persistent = [] {
    int m = 3;
    if (const char* e = getenv("AC_HEAD_PERSISTENT")) if (atoi(e) == 0) m &= ~1;
    if (const char* e = getenv("AC_BERT_SMALL")) if (atoi(e) == 0) m &= ~2;
    return m;
}();
bool test_hooks_enabled() {
    static const bool on = [] { const char* e = getenv("AC_TEST_HOOKS"); return e && atoi(e) != 0; }();
    return on;
}
CallOpts& call_opts() {
    static thread_local CallOpts o;
    return o;
}
CallScope::CallScope(int arith_opt, int ln_fusion_opt, int one_launch_opt) : saved(call_opts()) {
    CallOpts& o = call_opts();
    if (arith_opt > 0) o.arith = arith_opt - 1;
    if (ln_fusion_opt > 0) o.ln_fusion = ln_fusion_opt - 1;
    if (one_launch_opt > 0) o.one_launch = one_launch_opt - 1;
}
CallScope::~CallScope() { call_opts() = saved; }
int persistent_mask() {
    const int o = call_opts().one_launch;              // per-call: bit 1 (bert_small.hip) only
    return o < 0 ? g_persistent : ((g_persistent & ~2) | (o ? 2 : 0));
}
int set_persistent_mask(int m) { const int old = g_persistent; if (m >= 0) g_persistent = m & 3; return old; }

// ---- how many CUs can this process's workgroups actually land on? ----
// Every "one residency round" decision of this library -- the grids of the persistent kernels (head_epoch.hip, bert_small.hip:
// grid barriers), the LayerNorm exchange between the tiles of a row panel and the one-round tile choice (gemm_pipe.hip), the
// sweeps' grids (knn_l2.hip) -- is a statement about co-residency: grid <= workgroups per CU x CUs.  hipDeviceProp's
// multiProcessorCount is the CHIP's CU count; under HSA_CU_MASK / ROC_GLOBAL_CU_MASK the process's queues reach fewer, and a
// grid sized for 256 would wait for workgroups that are not resident (the bounded waits then give up into NaN rows and the host
// repeats the call unfused: correct, but discovered by failure).  So, when the environment carries a CU mask, the count is
// MEASURED once per device: a probe launch of 8 x CUs short workgroups (<= 4 resident per CU by LDS) each reports the (XCC, SE,
// SH, CU) it ran on; the number of distinct places is what dev_info().cus reports and what every such decision uses.  A mask on a
// caller's own stream (hipExtStreamCreateWithCUMask) cannot be seen from here: such a caller sets AC_ACTIVE_CUS.
namespace {
__global__ __launch_bounds__(256) void cu_probe_kernel(unsigned* __restrict__ out) {
    __shared__ unsigned pad[10 * 1024];                                  // 40 KB: at most 4 of these per CU, so the grid spreads out
    pad[threadIdx.x] = threadIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;       // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);               // HW_REG_HW_ID[15:0]: ..., CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(64);            // ~ 30 k cycles: the first wave of workgroups is still there when the last is placed
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = 0x80000000u | (xcc << 8) | ((hw >> 8) & 0xffu) | (pad[1] & 0u);
}
std::atomic<int> g_active_cus[64];                                       // per device, 0 = not measured yet (process-wide)

int measure_active_cus(int hw_cus) {
    if (const char* e = getenv("AC_ACTIVE_CUS")) { const int v = atoi(e); if (v >= 1) return v < hw_cus ? v : hw_cus; }
    // Only a process whose queues are restricted needs the measurement; without a mask in the environment the chip's count stands.
    // (A probe that shares the device with work already in flight -- the caller's own kernels on another stream -- can find CUs
    //  full and undercount: seen once as 255 of 256 next to a 30 GB generator kernel.  So the masked case also waits for the device
    //  to drain first; it is a one-time cost per process and device.)
    if (!getenv("HSA_CU_MASK") && !getenv("ROC_GLOBAL_CU_MASK")) return hw_cus;
    (void)hipDeviceSynchronize();
    const int blocks = 8 * hw_cus;
    unsigned* d = nullptr;
    hipStream_t st = nullptr;
    int best = 0;
    if (hipMalloc(&d, (size_t)blocks * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return hw_cus; }
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        unsigned* h = (unsigned*)malloc((size_t)blocks * sizeof(unsigned));
        for (int rep = 0; rep < 2 && h; ++rep) {                         // (twice: the larger count stands)
            if (hipMemsetAsync(d, 0, (size_t)blocks * sizeof(unsigned), st) != hipSuccess) break;
            hipLaunchKernelGGL(cu_probe_kernel, dim3(blocks), dim3(256), 0, st, d);
            if (hipMemcpyAsync(h, d, (size_t)blocks * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipStreamSynchronize(st) != hipSuccess) break;
            unsigned char seen[16 * 256 / 8] = {0};
            int n = 0;
            for (int i = 0; i < blocks; ++i)
                if (h[i] & 0x80000000u) {
                    const unsigned key = h[i] & 0xfffu;
                    if (!(seen[key >> 3] & (1u << (key & 7)))) { seen[key >> 3] |= (unsigned char)(1u << (key & 7)); ++n; }
                }
            if (n > best) best = n;
        }
        free(h);
        (void)hipStreamDestroy(st);
    }
    (void)hipFree(d);
    (void)hipGetLastError();
    return best >= 1 && best <= hw_cus ? best : hw_cus;                  // (a probe that could not run: the chip's count)
}
}  // namespace

const DevInfo& dev_info() {
    static thread_local DevInfo info = {0, 0, 0, 0};
    static thread_local int cached_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (cached_dev != dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
            info.hw_cus = p.multiProcessorCount;
            info.lds_per_block = (int)p.sharedMemPerBlock;
            info.hbm_bytes = p.totalGlobalMem;
            int act = (dev >= 0 && dev < 64) ? g_active_cus[dev].load(std::memory_order_acquire) : 0;
            if (act <= 0) {
                act = measure_active_cus(info.hw_cus);
                if (dev >= 0 && dev < 64) g_active_cus[dev].store(act, std::memory_order_release);
            }
            info.cus = act;
        } else {
            info.cus = info.hw_cus = 256;
            info.lds_per_block = 65536;
            info.hbm_bytes = 0;
        }
        cached_dev = dev;
    }
    return info;
}

bool first_call_on_device(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;      // unknown device: do the work every time
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_relaxed) & bit) return false;
    done.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

}  // namespace ac

extern "C" const char* ac_last_error(void) { return ac::g_err; }

extern "C" int ac_version(void) { return 1; }

extern "C" int ac_device_info(int* cu_count, int* lds_bytes_per_block, size_t* hbm_bytes) {
    const ac::DevInfo& d = ac::dev_info();
    if (cu_count) *cu_count = d.cus;
    if (lds_bytes_per_block) *lds_bytes_per_block = d.lds_per_block;
    if (hbm_bytes) *hbm_bytes = d.hbm_bytes;
    return AC_OK;
}

extern "C" int ac_persistent_launches(int64_t* head_epoch, int64_t* bert_small) {
    if (head_epoch) *head_epoch = (int64_t)ac::head_epoch_launches();
    if (bert_small) *bert_small = (int64_t)ac::bert_small_launches();
    return AC_OK;
}

extern "C" int ac_device_cus(int* chip_cus, int* active_cus) {
    const ac::DevInfo& d = ac::dev_info();
    if (chip_cus) *chip_cus = d.hw_cus;
    if (active_cus) *active_cus = d.cus;
    return AC_OK;
}

extern "C" int ac_set_persistent_kernels(int mask) {
    if (mask >= 0 && !ac::test_hooks_enabled()) {        // (a query, mask < 0, is always answered)
        ac::set_error("ac_set_persistent_kernels: process-wide switches are test hooks; this process did not enable them (AC_TEST_HOOKS=1)");
        return ac::set_persistent_mask(-1);
    }
    return ac::set_persistent_mask(mask);
}

// ---- ac_clock_stamp: (shader clock, 100 MHz real-time clock) of every XCD, for "what clock did this timed region run at" ----
namespace {
__global__ void clock_stamp_kernel(unsigned long long* out) {
    // HW_REG_XCC_ID (hwreg 20), bits [3:0] = the XCD this workgroup runs on (MI355X_MICROARCH.md); 64 workgroups cover all 8
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
    if (threadIdx.x == 0) {
        const unsigned long long c = __builtin_readcyclecounter();      // s_memtime: one tick per shader cycle
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
        out[2 * xcc] = c;                                               // (several workgroups of an XCD write near-equal pairs;
        out[2 * xcc + 1] = r;                                           //  a torn pair would need two writers 2^32 ticks apart)
    }
}
}  // namespace

extern "C" int ac_clock_stamp(unsigned long long* d_out16, ac_stream_t stream_) {
    AC_REQUIRE(d_out16 != nullptr, AC_EINVAL, "clock_stamp: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    AC_HIP_CHECK(hipMemsetAsync(d_out16, 0, 16 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(64), dim3(64), 0, stream, d_out16);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
