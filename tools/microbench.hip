// Micro-benchmarks behind the open questions of DESIGN.md §7 (gfx950).  Stand-alone: not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/icv_microbench tools/microbench.hip
//   /tmp/icv_microbench            (on the GPU box; prints one line per measurement)
//
// What it measures, and which design decision each number feeds:
//   chain_*      issue-to-issue cost of DEPENDENT float64 operations (v_add_f64, v_fma_f64, cvt+add) with 1, 2 and 4
//                wavefronts per SIMD: is the S / W chain of k_smooth_ws latency- or issue-bound?
//   lds_*        ds_read_b128 dependent latency and streaming rate; random ds_write_b32 scatter (bank conflicts)
//   barrier_*    cost of a workgroup barrier for 512- and 1024-thread workgroups (k_smooth_sp / a 1024-thread ws)
//   hbm_*        streaming reads with a bounded number of bytes in flight per CU: loaded latency and the
//                bandwidth reachable with 80 / 160 / 320 KB in flight per CU (rows prefetched ahead)
//   sphase_*, wphase_*   the S and W loops of k_smooth_ws in isolation (synthetic LDS contents), 8 and 16 wavefronts
//                per CU, alone on the CU: the floor of the serial chain of one cell
// Times come from s_memtime (shader clock) inside the kernels and from hipEvents around them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            std::exit(1);                                                                    \
        }                                                                                    \
    } while (0)

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------
// dependent float64 chains: every wavefront runs N dependent operations; cycles per operation as seen by
// wavefront 0 of the workgroup.  WAVES wavefronts per workgroup, one workgroup per CU -> WAVES / 4 per SIMD.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ void k_chain(double* out, u64* cyc, int n, double seed) {
    double a = seed + threadIdx.x, b = 1.0 + 1e-9 * threadIdx.x;
    float f = (float)seed;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) a = a + b;                    // v_add_f64
        if (KIND == 1) a = fma(a, b, b);             // v_fma_f64
        if (KIND == 2) {                             // v_cvt_f64_f32 + v_add_f64 (S phase pattern)
            a = a + (double)f;
            f = (float)a * 0.5f;                     // keeps the conversion on the chain
        }
        if (KIND == 3) {                             // two independent chains per lane
            a = a + b;
            b = fma(b, 1.0000001, 1e-12);
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + f;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// LDS: dependent ds_read_b128 (pointer chase), streaming ds_read_b128 (16 B lane stride), random ds_write_b32
// ------------------------------------------------------------------------------------------------
__global__ void k_lds_chase(int* out, u64* cyc, int n) {
    __shared__ int4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_int4((i * 37 + 11) & 1023, 0, 0, 0);
    __syncthreads();
    int p = threadIdx.x & 1023;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) p = buf[p].x;
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_lds_stream(double* out, u64* cyc, int n) {
    extern __shared__ double2 sbuf[];
    const int nelem = 4096;  // 64 KB
    for (int i = threadIdx.x; i < nelem; i += blockDim.x) sbuf[i] = make_double2(i, 1.0);
    __syncthreads();
    double acc = 0.0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        const int base = (threadIdx.x + i * 64) & (nelem - 16);
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sbuf[(base + u) & (nelem - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// every lane writes 4 B to a pseudo-random slot of a 80 KB row (RANDOM) or to consecutive slots
template <bool RANDOM>
__global__ void k_lds_scatter(float* out, u64* cyc, int n) {
    extern __shared__ float frow[];
    const int nslot = 20000;
    unsigned s = threadIdx.x * 2654435761u + 12345u;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const int slot = RANDOM ? (int)((s >> 8) % nslot) : (int)((threadIdx.x + (i * 8 + u) * blockDim.x) % nslot);
            frow[slot] = (float)u;
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = frow[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// workgroup barriers
// ------------------------------------------------------------------------------------------------
__global__ void k_barrier(int* out, u64* cyc, int n) {
    int x = threadIdx.x;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        x = x * 3 + 1;
        __syncthreads();
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// HBM: persistent 512-thread workgroups stream rows of 80 000 B; VEC x 16 B per thread requested at once, then
// consumed (one add per dword) before the next request -> VEC * 8 KB in flight per workgroup, no overlap between
// requests of a workgroup (the worst case: latency fully exposed), WGS workgroups per CU overlap each other.
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(512) k_hbm(const float* x, long n_rows, long ld, float* out, u64* cyc) {
    float acc = 0.0f;
    u64 wait = 0;
    for (long row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const float4* p = reinterpret_cast<const float4*>(x + row * ld);
        for (int c0 = 0; c0 < 5000; c0 += VEC * 512) {  // 5000 float4 per row
            float4 v[VEC];
            const u64 t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                const int c = c0 + u * 512 + threadIdx.x;
                v[u] = c < 5000 ? p[c] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < VEC; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            wait += __builtin_amdgcn_s_memtime() - t0;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = wait;
}

// ------------------------------------------------------------------------------------------------
// the S phase (block sums of 10 genes, float64) and the W phase (10 {S0,S1} pairs per window, serial float64
// chain + 3-FMA division) of k_smooth_ws on synthetic LDS contents; NTHR threads, NB = 2000 blocks, W = 1802
// ------------------------------------------------------------------------------------------------
template <int NTHR>
__global__ void __launch_bounds__(NTHR) k_sphase(double* out, u64* cyc, int n) {
    extern __shared__ float srow[];
    for (int i = threadIdx.x; i < 20000; i += NTHR) srow[i] = 1e-3f * (i % 97);
    __syncthreads();
    double tot = 0.0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        for (int b = threadIdx.x; b < 2000; b += NTHR) {
            const float2* rp = reinterpret_cast<const float2*>(srow + b * 10);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int r = 0; r < 10; r += 2) {
                const float2 v = rp[r >> 1];
                s0 = s0 + (double)v.x;
                s1 = fma((double)r, (double)v.x, s1);
                s0 = s0 + (double)v.y;
                s1 = fma((double)(r + 1), (double)v.y, s1);
            }
            tot += s0 + s1;
        }
        __syncthreads();
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NTHR + threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NTHR, int GW>
__global__ void __launch_bounds__(NTHR) k_wphase(double* out, u64* cyc, int n) {
    extern __shared__ double2 s01[];
    for (int i = threadIdx.x; i < 2048; i += NTHR) s01[i] = make_double2(1e-3 * (i % 89), 1e-4 * (i % 13));
    __syncthreads();
    double tot = 0.0;
    const double den = 2550.0, rcp = 1.0 / 2550.0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        for (int j0 = threadIdx.x; j0 < 1802; j0 += NTHR * GW) {
            double v[GW];
            const double2* sp[GW];
#pragma unroll
            for (int g = 0; g < GW; ++g) {
                const int j = j0 + g * NTHR < 1802 ? j0 + g * NTHR : j0;
                sp[g] = s01 + j;
                v[g] = 0.0;
            }
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                double2 sv[GW];
#pragma unroll
                for (int g = 0; g < GW; ++g) sv[g] = sp[g][m];
#pragma unroll
                for (int g = 0; g < GW; ++g) v[g] = fma((double)(m < 5 ? m * 10 + 1 : 100 - m * 10), sv[g].x, v[g]);
#pragma unroll
                for (int g = 0; g < GW; ++g) v[g] = m < 5 ? v[g] + sv[g].y : v[g] - sv[g].y;
            }
#pragma unroll
            for (int g = 0; g < GW; ++g) {
                const double q = v[g] * rcp;
                const double r = fma(-q, den, v[g]);
                tot += fma(r, rcp, q);
            }
        }
        __syncthreads();
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NTHR + threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
static double mean_cycles(const u64* d_cyc, int n) {
    std::vector<u64> h(n);
    CHECK(hipMemcpy(h.data(), d_cyc, n * sizeof(u64), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)h[i];
    return s / n;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
    void* d_out;
    u64* d_cyc;
    CHECK(hipMalloc(&d_out, 64 << 20));
    CHECK(hipMalloc((void**)&d_cyc, 1 << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // ---- dependent chains: one workgroup per CU with 256 / 512 / 1024 threads = 1 / 2 / 4 wavefronts per SIMD
    const int n_chain = 4096;
    const char* chain_name[4] = {"add_f64", "fma_f64", "cvt+add_f64", "2 chains (add, fma)"};
    for (int threads : {256, 512, 1024}) {
        for (int kind = 0; kind < 4; ++kind) {
            void (*k)(double*, u64*, int, double) =
                kind == 0 ? k_chain<0> : kind == 1 ? k_chain<1> : kind == 2 ? k_chain<2> : k_chain<3>;
            hipLaunchKernelGGL(k, dim3(n_cu), dim3(threads), 0, 0, (double*)d_out, d_cyc, n_chain, 1.5);
            CHECK(hipDeviceSynchronize());
            std::printf("chain_%-22s %d waves/SIMD: %.2f cycles per dependent step\n", chain_name[kind], threads / 256,
                        mean_cycles(d_cyc, n_cu) / n_chain);
        }
    }
    // ---- LDS
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k_lds_chase, dim3(n_cu), dim3(threads), 0, 0, (int*)d_out, d_cyc, 2048);
        CHECK(hipDeviceSynchronize());
        std::printf("lds_chase_b128   %4d threads/CU: %.1f cycles per dependent ds_read_b128\n", threads,
                    mean_cycles(d_cyc, n_cu) / 2048);
        hipLaunchKernelGGL(k_lds_stream, dim3(n_cu), dim3(threads), 65536, 0, (double*)d_out, d_cyc, 512);
        CHECK(hipDeviceSynchronize());
        const double cyc = mean_cycles(d_cyc, n_cu);
        std::printf("lds_stream_b128  %4d threads/CU: %.1f B/cycle/CU\n", threads, 512.0 * 8 * 16 * threads / cyc);
        hipLaunchKernelGGL(k_lds_scatter<true>, dim3(n_cu), dim3(threads), 80016, 0, (float*)d_out, d_cyc, 256);
        CHECK(hipDeviceSynchronize());
        const double cr = mean_cycles(d_cyc, n_cu);
        hipLaunchKernelGGL(k_lds_scatter<false>, dim3(n_cu), dim3(threads), 80016, 0, (float*)d_out, d_cyc, 256);
        CHECK(hipDeviceSynchronize());
        const double cs = mean_cycles(d_cyc, n_cu);
        std::printf("lds_scatter_b32  %4d threads/CU: random %.2f, consecutive %.2f cycles per wavefront write\n", threads,
                    cr / (256.0 * 8 * threads / 64), cs / (256.0 * 8 * threads / 64));
    }
    // ---- barriers
    for (int threads : {512, 1024}) {
        for (int wgs : {1, 2}) {
            if (threads * wgs > 2048) continue;
            hipLaunchKernelGGL(k_barrier, dim3(n_cu * wgs), dim3(threads), 0, 0, (int*)d_out, d_cyc, 4096);
            CHECK(hipDeviceSynchronize());
            std::printf("barrier          %4d threads x %d WG/CU: %.1f cycles per barrier\n", threads, wgs,
                        mean_cycles(d_cyc, n_cu * wgs) / 4096);
        }
    }
    // ---- HBM with bounded bytes in flight
    {
        const long n_rows = 100000, ld = 20000;
        float* d_x;
        CHECK(hipMalloc((void**)&d_x, n_rows * ld * sizeof(float)));
        CHECK(hipMemset(d_x, 0, n_rows * ld * sizeof(float)));
        for (int wgs : {1, 2, 4}) {
            for (int vec : {2, 5, 10}) {
                void (*k)(const float*, long, long, float*, u64*) = vec == 2 ? k_hbm<2> : vec == 5 ? k_hbm<5> : k_hbm<10>;
                hipLaunchKernelGGL(k, dim3(n_cu * wgs), dim3(512), 0, 0, d_x, n_rows, ld, (float*)d_out, d_cyc);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(n_cu * wgs), dim3(512), 0, 0, d_x, n_rows, ld, (float*)d_out, d_cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double reqs = (double)n_rows / (n_cu * wgs) * ((5000 + vec * 512 - 1) / (vec * 512));
                std::printf("hbm  %d WG/CU x %3d KB in flight each: %.2f TB/s, %.0f cycles per request round trip\n", wgs,
                            vec * 8, n_rows * ld * 4.0 / (ms * 1e-3) / 1e12, mean_cycles(d_cyc, n_cu * wgs) / reqs);
            }
        }
        CHECK(hipFree(d_x));
    }
    // ---- S and W phases alone
    {
        hipLaunchKernelGGL(k_sphase<512>, dim3(n_cu), dim3(512), 80016, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("sphase  512 threads alone: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
        hipLaunchKernelGGL(k_sphase<1024>, dim3(n_cu), dim3(1024), 80016, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("sphase 1024 threads alone: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
        hipLaunchKernelGGL(k_sphase<512>, dim3(2 * n_cu), dim3(512), 80016, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("sphase  512 threads, 2 WG/CU: %.0f cycles per cell of one workgroup\n", mean_cycles(d_cyc, 2 * n_cu) / 64);
        hipLaunchKernelGGL((k_wphase<512, 1>), dim3(n_cu), dim3(512), 32768, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("wphase  512 threads, 1 window at a time: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
        hipLaunchKernelGGL((k_wphase<512, 2>), dim3(n_cu), dim3(512), 32768, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("wphase  512 threads, 2 windows interleaved: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
        hipLaunchKernelGGL((k_wphase<512, 4>), dim3(n_cu), dim3(512), 32768, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("wphase  512 threads, 4 windows interleaved: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
        hipLaunchKernelGGL((k_wphase<1024, 2>), dim3(n_cu), dim3(1024), 32768, 0, (double*)d_out, d_cyc, 64);
        CHECK(hipDeviceSynchronize());
        std::printf("wphase 1024 threads, 2 windows interleaved: %.0f cycles per cell\n", mean_cycles(d_cyc, n_cu) / 64);
    }
    std::printf("done\n");
    return 0;
}
