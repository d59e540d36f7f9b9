// Standalone A/B harness for the planes GEMM variants (no torch: runs in seconds on the GPU box).
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_bench.hip -o tools/ab/gemm_bench \
//         -Ladaptive-classifier_amd/adaptive_classifier -lacamd -Wl,-rpath,'$ORIGIN/../../adaptive-classifier_amd/adaptive_classifier'
//   tools/ab/gemm_bench <variants, e.g. 1,2231,2261> <reps> <rounds> [M,N,K,act,res,cplanes ...]
// For every shape: R interleaved rounds over all variants (ac_gemm_set_variant) of ac_linear_bf16x3 with pre-split
// operands; prints median and min time per variant and compares every variant's output with the first variant's bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../include/acamd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define AC(x) do { int r_ = (x); if (r_ != 0) { printf("acamd error %d: %s at line %d\n", r_, ac_last_error(), __LINE__); exit(1); } } while (0)

__global__ void fill(float* x, size_t n, uint64_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    x[i] = ((float)(z >> 40) * (1.0f / 16777216.0f) - 0.5f) * 2.f * scale;
}

struct Shape { int M, N, K, act, res, cplanes; };

int main(int argc, char** argv) {
    setenv("AC_TEST_HOOKS", "1", 0);        // ac_gemm_set_variant / ac_gemm_debug_stamps are test hooks (include/acamd.h)
    std::vector<int> variants = {1};
    int reps = 20, rounds = 3;
    std::vector<Shape> shapes = {{5141, 2304, 768, 0, 0, 0}, {5141, 768, 768, 0, 1, 0}, {5141, 3072, 768, 2, 0, 1},
                                 {5141, 768, 3072, 0, 1, 0}};
    if (argc > 1) { variants.clear(); char* s = strdup(argv[1]); for (char* t = strtok(s, ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
    if (argc > 2) reps = atoi(argv[2]);
    if (argc > 3) rounds = atoi(argv[3]);
    if (argc > 4) {
        shapes.clear();
        for (int i = 4; i < argc; ++i) { Shape s{0, 0, 0, 0, 0, 0}; sscanf(argv[i], "%d,%d,%d,%d,%d,%d", &s.M, &s.N, &s.K, &s.act, &s.res, &s.cplanes); shapes.push_back(s); }
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const size_t MA = (size_t)sh.M * sh.K, MW = (size_t)sh.N * sh.K, MC = (size_t)sh.M * sh.N;
        float *A, *W, *bias, *R, *C; uint16_t *Ap, *Wp, *Cp;
        CK(hipMalloc(&A, MA * 4)); CK(hipMalloc(&W, MW * 4)); CK(hipMalloc(&bias, sh.N * 4)); CK(hipMalloc(&R, MC * 4));
        CK(hipMalloc(&C, MC * 4));
        CK(hipMalloc(&Ap, MA * 6)); CK(hipMalloc(&Wp, MW * 6)); CK(hipMalloc(&Cp, MC * 6));
        hipLaunchKernelGGL(fill, dim3((MA + 255) / 256), dim3(256), 0, st, A, MA, 1, 1.0f);
        hipLaunchKernelGGL(fill, dim3((MW + 255) / 256), dim3(256), 0, st, W, MW, 2, 0.05f);
        hipLaunchKernelGGL(fill, dim3((sh.N + 255) / 256), dim3(256), 0, st, bias, (size_t)sh.N, 3, 0.1f);
        hipLaunchKernelGGL(fill, dim3((MC + 255) / 256), dim3(256), 0, st, R, MC, 4, 1.0f);
        AC(ac_split_bf16x3(A, sh.K, sh.M, sh.K, Ap, st));
        AC(ac_split_bf16x3(W, sh.K, sh.N, sh.K, Wp, st));
        CK(hipStreamSynchronize(st));
        printf("M=%d N=%d K=%d act=%d res=%d cplanes=%d\n", sh.M, sh.N, sh.K, sh.act, sh.res, sh.cplanes);
        const size_t cbytes = sh.cplanes ? MC * 6 : MC * 4;
        std::vector<uint8_t> ref(cbytes), got(cbytes);
        auto run = [&]() -> int { return ac_linear_bf16x3(A, sh.K, Ap, W, sh.K, Wp, bias, sh.res ? R : nullptr, sh.N, sh.cplanes ? nullptr : C, sh.N,
                                                         sh.cplanes ? Cp : nullptr, sh.M, sh.N, sh.K, sh.act, st); };
        std::vector<std::vector<double>> t(variants.size());
        std::vector<int> ok(variants.size(), 1);
        std::vector<size_t> ndiff(variants.size(), 0);
        std::vector<double> maxd(variants.size(), 0.0);
        // correctness pass (and warm-up)
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            AC(ac_gemm_set_variant(variants[vi]));
            CK(hipMemsetAsync(sh.cplanes ? (void*)Cp : (void*)C, 0xff, cbytes, st));
            if (run() != 0) { ok[vi] = 0; continue; }
            run(); run();
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy((vi == 0 ? ref : got).data(), sh.cplanes ? (void*)Cp : (void*)C, cbytes, hipMemcpyDeviceToHost));
            if (vi > 0) {
                if (sh.cplanes) { for (size_t i = 0; i < cbytes; ++i) ndiff[vi] += ref[i] != got[i]; }
                else {
                    const float* a = (const float*)ref.data(); const float* b = (const float*)got.data();
                    for (size_t i = 0; i < MC; ++i) { if (!(a[i] == b[i])) { ++ndiff[vi]; double d = fabs((double)a[i] - b[i]); if (!(d <= maxd[vi])) maxd[vi] = d; } }
                }
            }
        }
        for (int r = 0; r < rounds; ++r)
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                if (!ok[vi]) continue;
                AC(ac_gemm_set_variant(variants[vi]));
                run();
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[vi].push_back(ms * 1e3 / reps);
            }
        static const bool want_stamps = getenv("GEMM_BENCH_STAMPS") != nullptr;
        std::vector<std::string> stamp_line(variants.size());
        if (want_stamps) {
            const int64_t cap = 1 << 16;
            unsigned long long* d_st; CK(hipMalloc(&d_st, cap * 4 * sizeof(unsigned long long)));
            std::vector<unsigned long long> h(cap * 4);
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                if (!ok[vi] || variants[vi] < 1000) continue;
                AC(ac_gemm_set_variant(variants[vi]));
                run(); run();
                CK(hipMemsetAsync(d_st, 0, cap * 4 * sizeof(unsigned long long), st));
                AC(ac_gemm_debug_stamps(d_st, cap));
                run();
                CK(hipStreamSynchronize(st));
                AC(ac_gemm_debug_stamps(nullptr, 0));
                CK(hipMemcpy(h.data(), d_st, cap * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::vector<double> pro, loop, epi, t0s, t3s;
                unsigned long long mn = ~0ull, mx = 0;
                for (int64_t b = 0; b < cap; ++b) {
                    const unsigned long long* q = &h[b * 4];
                    if (!q[0] || !q[3]) continue;
                    pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                    mn = std::min(mn, q[0]); mx = std::max(mx, q[3]);
                }
                if (pro.empty()) continue;
                for (int64_t b = 0; b < cap; ++b) { const unsigned long long* q = &h[b * 4]; if (q[0] && q[3]) { t0s.push_back((double)(q[0] - mn)); t3s.push_back((double)(mx - q[3])); } }
                auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                auto p95 = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[(size_t)(v.size() * 0.95)]; };
                char buf[512];
                snprintf(buf, sizeof buf, "      stamps (shader cycles, %zu workgroups): span %.0f | median prologue %.0f  k-loop %.0f  epilogue %.0f | start p95 %.0f  early-finish p95 %.0f",
                         pro.size(), (double)(mx - mn), med(pro), med(loop), med(epi), p95(t0s), p95(t3s));
                stamp_line[vi] = buf;
            }
            hipFree(d_st);
        }
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            if (!ok[vi]) { printf("  variant %5d: not built / not applicable (%s)\n", variants[vi], ac_last_error()); continue; }
            std::sort(t[vi].begin(), t[vi].end());
            const double med = t[vi][t[vi].size() / 2], mn = t[vi][0];
            const double tf = 2.0 * sh.M * sh.N * (double)sh.K / (med * 1e-6) / 1e12;
            printf("  variant %5d: med %8.1f us  min %8.1f us  %6.1f TF fp32-equiv (%.3f of 416.7)", variants[vi], med, mn, tf, tf / 416.7);
            if (vi > 0) printf("   vs first: %zu differing, max |diff| %.3g", ndiff[vi], maxd[vi]);
            printf("\n");
            if (!stamp_line[vi].empty()) printf("%s\n", stamp_line[vi].c_str());
        }
        fflush(stdout);
        hipFree(A); hipFree(W); hipFree(bias); hipFree(R); hipFree(C); hipFree(Ap); hipFree(Wp); hipFree(Cp);
    }
    AC(ac_gemm_set_variant(0));
    return 0;
}
