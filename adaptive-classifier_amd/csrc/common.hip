// Error string, version and device info for libacamd.so.
#include "common.h"

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

const DevInfo& dev_info() {
    static thread_local DevInfo info = {0, 0, 0};
    static thread_local int cached_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (cached_dev != dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
            info.cus = p.multiProcessorCount;
            info.lds_per_block = (int)p.sharedMemPerBlock;
            info.hbm_bytes = p.totalGlobalMem;
        } else {
            info.cus = 256;
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

extern "C" int ac_set_persistent_kernels(int mask) { return ac::set_persistent_mask(mask); }

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
