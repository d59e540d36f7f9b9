// Scratch probe: CPU enqueue cost + wall time of a 15-kernel dependent chain, direct launches vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k(float* x, int n, int work) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = x[i]; for (int j = 0; j < work; ++j) v = fmaf(v, 1.0001f, 0.5f); x[i] = v; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int NK = 15, REP = 2000;
    for (int work : {1, 600}) {       // ~2 us and ~6 us kernels
        auto chain = [&](hipStream_t st) { for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, d, 1 << 16, work); };
        for (int i = 0; i < 50; ++i) chain(s);
        hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < REP; ++i) chain(s);
        double t1 = now(); hipStreamSynchronize(s); double t2 = now();
        printf("work=%d direct : enqueue %.1f us/chain, wall %.1f us/chain\n", work, (t1 - t0) / REP * 1e6, (t2 - t0) / REP * 1e6);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); chain(s); hipStreamEndCapture(s, &g);
        double c0 = now(); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); double c1 = now();
        for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = now();
        for (int i = 0; i < REP; ++i) hipGraphLaunch(ge, s);
        t1 = now(); hipStreamSynchronize(s); t2 = now();
        printf("work=%d graph  : enqueue %.1f us/chain, wall %.1f us/chain (instantiate %.0f us)\n", work, (t1 - t0) / REP * 1e6, (t2 - t0) / REP * 1e6, (c1 - c0) * 1e6);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
