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
int persistent_mask() { return g_persistent; }
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
