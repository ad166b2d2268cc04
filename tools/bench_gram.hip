// Where does k_gram_mfma's time go?  Same kernel with parts compiled out (results are wrong for EXP != 0):
//   EXP 0 the kernel; 1 no global loads inside the K loop; 2 also no LDS traffic (MFMA + barrier);
//   3 MFMA only (no barrier).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I infercnvpy_amd/csrc tools/bench_gram.hip -o tools/bench_gram.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "icv_corr.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int EXP>
static double run(const float* z, int64_t n, int kz, float* out, const double* norm, int reps) {
    const int64_t ss = 1024, ns = (n + ss - 1) / ss;
    std::vector<icv::GramSuper> v;
    for (int64_t sy = 0; sy < ns; ++sy)
        for (int64_t sx = sy; sx < ns; ++sx) v.push_back({(int)(sy * ss), (int)(sx * ss), sy * ss * n + sx * ss, sx * ss * n + sy * ss});
    char* buf;
    hipMalloc((void**)&buf, v.size() * sizeof(icv::GramSuper));
    hipMemcpy(buf, v.data(), v.size() * sizeof(icv::GramSuper), hipMemcpyHostToDevice);
    icv::GramJob J{reinterpret_cast<const icv::GramSuper*>(buf), (int)v.size(), n, n, out, n, out, n};
    const unsigned grid = 512u * (unsigned)((v.size() + 7) / 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((icv::k_gram_mfma<true, true, EXP>), dim3(grid), dim3(256), 0, 0, z, kz, norm, J);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((icv::k_gram_mfma<true, true, EXP>), dim3(grid), dim3(256), 0, 0, z, kz, norm, J);
    }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipFree(buf);
    const double tf = (double)n * (double)(n + 128) * kz * reps / (ms * 1e-3) / 1e12;
    printf("EXP %d: n %lld kz %d  %.2f ms per launch  %.1f TFLOP/s executed (%.3f of 157.3)\n", EXP, (long long)n, kz, ms / reps, tf, tf / 157.3);
    return tf;
}

// MFMA issue rate alone: W wavefronts per SIMD, four independent 32x32x2 accumulators each
__global__ void __launch_bounds__(256) k_mfma_only(float* out, int iters) {
    icv::f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const float x = (float)threadIdx.x, y = (float)blockIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    float s = 0;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}
static void mfma_only(float* out, int wgs_per_cu, int iters, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_mfma_only, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double flop = 256.0 * wgs_per_cu * 4 * iters * 4 * 4096.0;
    printf("mfma only, %d wavefront(s) per SIMD, %s: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, what, ms, flop / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 32768;
    const int kz = argc > 2 ? atoi(argv[2]) : 5008;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    float *z, *out;
    double* norm;
    CK(hipMalloc((void**)&z, (size_t)n * kz * 4));
    CK(hipMalloc((void**)&out, (size_t)n * n * 4));
    CK(hipMalloc((void**)&norm, (size_t)n * 8));
    std::vector<float> h((size_t)n * kz);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMemcpy(z, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(norm, 0, (size_t)n * 8));
    mfma_only(out, 1, 1000, "first launch");
    mfma_only(out, 1, 10000, "~2 ms");
    mfma_only(out, 2, 5000, "~2 ms");
    mfma_only(out, 1, 500000, "~100 ms");
    mfma_only(out, 2, 250000, "~100 ms");
    mfma_only(out, 2, 5000, "~2 ms after load");
    run<0>(z, n, kz, out, norm, reps);
    run<1>(z, n, kz, out, norm, reps);
    run<2>(z, n, kz, out, norm, reps);
    run<3>(z, n, kz, out, norm, reps);
    run<0>(z, n, kz, out, norm, reps);
    return 0;
}
