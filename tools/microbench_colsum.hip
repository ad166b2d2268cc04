// Column-sum access patterns over a row-major float32 matrix (100 000 x 20 000 = 8 GB): which shape streams HBM fastest?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_colsum.bin tools/microbench_colsum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V0: the shipped kernel's shape: block = 256 columns x `rps` rows, one column per thread, 8 rows in flight
template <int UNROLL, bool NT_>
__global__ void __launch_bounds__(256) k_cols1(const float* x, int64_t n_rows, int64_t ld, int n_cols, int rps, double* partial) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rps;
    int64_t r1 = r0 + rps; if (r1 > n_rows) r1 = n_rows;
    double acc = 0.0;
    if (col < n_cols) {
        int64_t r = r0;
        for (; r + UNROLL <= r1; r += UNROLL) {
            float v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = NT_ ? __builtin_nontemporal_load(&x[(r + u) * ld + col]) : x[(r + u) * ld + col];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += (double)v[u];
        }
        for (; r < r1; ++r) acc += (double)x[r * ld + col];
        partial[(int64_t)blockIdx.y * n_cols + col] = acc;
    }
}
// V1: four columns per thread (16-byte loads): block = 1024 columns x rps rows
template <int UNROLL, bool NT_, bool SWAP>
__global__ void __launch_bounds__(256) k_cols4(const float* x, int64_t n_rows, int64_t ld, int n_cols, int rps, double* partial) {
    const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
    const int col = (bx * 256 + threadIdx.x) * 4;
    const int64_t r0 = (int64_t)by * rps;
    int64_t r1 = r0 + rps; if (r1 > n_rows) r1 = n_rows;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (col < n_cols) {
        int64_t r = r0;
        for (; r + UNROLL <= r1; r += UNROLL) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const f4* p = reinterpret_cast<const f4*>(x + (r + u) * ld + col);
                v[u] = NT_ ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
        }
        for (; r < r1; ++r) { const f4 q = *reinterpret_cast<const f4*>(x + r * ld + col); a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w; }
        double* o = partial + (int64_t)by * n_cols + col;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}
// V2: persistent: one workgroup walks whole rows (4 columns per thread, 1024 columns per pass), `rps` rows per workgroup,
// float32 accumulation per pass of 8 rows then widened (fewer float64 adds)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_rows(const float* x, int64_t n_rows, int64_t ld, int n_cols, int rps, double* partial) {
    const int64_t r0 = (int64_t)blockIdx.x * rps;
    int64_t r1 = r0 + rps; if (r1 > n_rows) r1 = n_rows;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int c0 = threadIdx.x * 4; c0 < n_cols; c0 += 1024) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int64_t r = r0;
        for (; r + UNROLL <= r1; r += UNROLL) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const f4*>(x + (r + u) * ld + c0);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
        }
        double* o = partial + (int64_t)blockIdx.x * n_cols + c0;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}

int main(int argc, char** argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 100000;
    const int n_cols = 20000;
    float* x; double* partial;
    CHECK(hipMalloc(&x, n_rows * n_cols * sizeof(float)));
    CHECK(hipMalloc(&partial, (size_t)4096 * n_cols * sizeof(double)));
    CHECK(hipMemset(x, 0, n_rows * n_cols * sizeof(float)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int it = 0; it < 10; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms; if (ms < best) best = ms;
        }
        CHECK(hipGetLastError());
        const double gb = (double)n_rows * n_cols * 4 / 1e9;
        printf("%-72s mean %.3f ms %.0f GB/s   best %.3f ms %.0f GB/s\n", name, tot / 10, gb / (tot / 10) * 1e3, best, gb / best * 1e3);
    };
    for (int rps : {128, 256, 512, 1024}) {
        const int ns = (int)((n_rows + rps - 1) / rps);
        char nm[128];
        snprintf(nm, sizeof nm, "1 col/thread, 256-col tiles, %d rows/slab, 8 in flight (shipped at 256)", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols1<8, false>), dim3((n_cols + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "1 col/thread, %d rows/slab, 16 in flight", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols1<16, false>), dim3((n_cols + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "1 col/thread, %d rows/slab, 8 in flight, nontemporal", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols1<8, true>), dim3((n_cols + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "4 cols/thread (16 B), 1024-col tiles, %d rows/slab, 4 in flight", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols4<4, false, false>), dim3((n_cols / 4 + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "4 cols/thread (16 B), %d rows/slab, 8 in flight", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols4<8, false, false>), dim3((n_cols / 4 + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "4 cols/thread (16 B), %d rows/slab, 8 in flight, nontemporal", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols4<8, true, false>), dim3((n_cols / 4 + 255) / 256, ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
        snprintf(nm, sizeof nm, "4 cols/thread (16 B), %d rows/slab, 8 in flight, slab index fastest", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_cols4<8, false, true>), dim3(ns, (n_cols / 4 + 255) / 256), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
    }
    for (int rps : {48, 96, 196}) {
        const int ns = (int)((n_rows + rps - 1) / rps);
        char nm[128];
        snprintf(nm, sizeof nm, "whole rows per workgroup (5 passes of 1024 columns), %d rows each, 8 in flight", rps);
        run(nm, [&] { hipLaunchKernelGGL((k_rows<8>), dim3(ns), dim3(256), 0, 0, x, n_rows, (int64_t)n_cols, n_cols, rps, partial); });
    }
    return 0;
}
