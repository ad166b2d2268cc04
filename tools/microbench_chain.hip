// Reference-order column sums (csrc/icv_kernel_chain.hpp): correctness against a one-thread-per-column chain and timing.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I infercnvpy_amd/csrc -o tools/microbench_chain.bin tools/microbench_chain.hip
//   tools/microbench_chain.bin [n_rows] [n_cols] [ld]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "icv_kernel_chain.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_cu = 256, g_lds = 0;
template <typename T>
__global__ void k_fill(T* x, int64_t n, unsigned seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
        x[i] = (h & 7u) < 5u ? (T)0 : (T)(u * 3.7f + 0.01f);  // ~62 % zeros, the rest 0.01 .. 3.71
    }
}
template <typename T>
__global__ void k_naive(const T* x, int64_t ld, int n_cols, const int32_t* sel, int64_t n_sel, T* acc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    T a = acc[c];
    for (int64_t i = 0; i < n_sel; ++i) {
        const int64_t r = sel ? sel[i] : i;
        a = a + x[r * ld + c];
    }
    acc[c] = a;
}

template <typename T>
int run(int64_t n_rows, int n_cols, int64_t ld, bool list) {
    T *x, *acc, *acc_ref;
    CHECK(hipMalloc(&x, (size_t)n_rows * ld * sizeof(T)));
    CHECK(hipMalloc(&acc, n_cols * sizeof(T)));
    CHECK(hipMalloc(&acc_ref, n_cols * sizeof(T)));
    hipLaunchKernelGGL(k_fill<T>, dim3(4096), dim3(256), 0, 0, x, n_rows * ld, 12345u);
    std::vector<int32_t> h_sel;
    int32_t* sel = nullptr;
    int64_t n_sel = n_rows;
    if (list) {
        for (int64_t r = 0; r < n_rows; ++r)
            if ((r * 2654435761u >> 7) % 10 < 3) h_sel.push_back((int32_t)r);
        n_sel = (int64_t)h_sel.size();
        CHECK(hipMalloc(&sel, (n_sel + 1) * sizeof(int32_t)));
        CHECK(hipMemcpy(sel, h_sel.data(), n_sel * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    constexpr int EPL = 16 / (int)sizeof(T);
    const icv::ChainLaunch L(n_cols, (int)sizeof(T), g_cu);
    const int lds = g_lds ? g_lds : L.lds_bytes;
    // the buffer's last row goes through the guarded tail when a 16-byte segment could run past the end
    int64_t tail = -1, n_dma = n_sel;
    const int64_t last = list ? (n_sel ? h_sel.back() : -1) : n_rows - 1;
    if (n_sel && last == n_rows - 1 && (n_cols + EPL - 1) / EPL * EPL > ld) { tail = last; n_dma = n_sel - 1; }
    auto kern = list ? icv::k_colchain<T, true> : icv::k_colchain<T, false>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, icv::kChLdsFull));
    auto launch = [&] {
        hipLaunchKernelGGL(kern, dim3(L.grid), dim3(icv::kChThreads), lds, 0, x, ld, n_cols, L.n_lines, lds, sel, n_dma, tail, acc);
    };
    CHECK(hipMemset(acc, 0, n_cols * sizeof(T)));
    CHECK(hipMemset(acc_ref, 0, n_cols * sizeof(T)));
    launch();
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_naive<T>, dim3((n_cols + 63) / 64), dim3(64), 0, 0, x, ld, n_cols, sel, n_sel, acc_ref);
    CHECK(hipDeviceSynchronize());
    std::vector<T> a(n_cols), b(n_cols);
    CHECK(hipMemcpy(a.data(), acc, n_cols * sizeof(T), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), acc_ref, n_cols * sizeof(T), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int c = 0; c < n_cols; ++c)
        if (std::memcmp(&a[c], &b[c], sizeof(T)) != 0 && bad++ < 5) printf("  col %d: %.9g vs %.9g\n", c, (double)a[c], (double)b[c]);
    printf("grid %d lds %d | %s rows %lld (selected %lld) cols %d ld %lld tail %lld: %s (%d columns differ), acc[0]=%.9g\n", L.grid, lds, sizeof(T) == 4 ? "f32" : "f64",
           (long long)n_rows, (long long)n_sel, n_cols, (long long)ld, (long long)tail, bad ? "MISMATCH" : "bit-exact", bad, (double)a[0]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    for (int it = 0; it < 10; ++it) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms; if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double gb = (double)n_sel * n_cols * sizeof(T) / 1e9;
    printf("  mean %.3f ms %.0f GB/s   best %.3f ms %.0f GB/s\n", tot / 10, gb / (tot / 10) * 1e3, best, gb / best * 1e3);
    hipFree(x); hipFree(acc); hipFree(acc_ref); if (sel) hipFree(sel);
    return bad;
}

// ---- CSR: every row K entries at ascending pseudo-random columns (one per stripe of n_cols / K columns) ----------------
__global__ void k_fill_csr(int64_t n_rows, int n_cols, int K, int64_t* indptr, int32_t* indices, float* vals) {
    const int stripe = n_cols / K;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * K; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / K;
        const int j = (int)(e - r * K);
        unsigned h = (unsigned)r * 2654435761u ^ (unsigned)j * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        indices[e] = j * stripe + (int)(h % (unsigned)stripe);
        vals[e] = 0.01f + (float)(h >> 8) * (3.7f / 16777216.0f);
        if (j == 0) indptr[r] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) indptr[n_rows] = n_rows * K;
}
__global__ void k_naive_csr(const float* vals, const int64_t* indptr, const int32_t* indices, int64_t n_rows, float scale, float* acc) {
    // one thread per row would race: one thread per COLUMN, binary search of the column in every row (slow, exact)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    float a = acc[c];
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t lo = indptr[r], hi = indptr[r + 1];
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (indices[mid] < c) lo = mid + 1; else hi = mid; }
        if (lo < indptr[r + 1] && indices[lo] == c) a = a + vals[lo] * scale;
    }
    acc[c] = a;
}
__global__ void k_line_tiles(int n_lines, int grid, uint16_t* line_tile) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= grid) return;
    const int l0 = (int)((int64_t)t * n_lines / grid), l1 = (int)((int64_t)(t + 1) * n_lines / grid);
    for (int l = l0; l < l1; ++l) line_tile[l] = (uint16_t)t;
}
int run_csr(int64_t n_rows, int n_cols, int K, bool verify) {
    int64_t* indptr; int32_t* indices; float *vals, *acc, *acc_ref;
    CHECK(hipMalloc(&indptr, (n_rows + 1) * 8));
    CHECK(hipMalloc(&indices, n_rows * K * 4));
    CHECK(hipMalloc(&vals, n_rows * K * 4));
    CHECK(hipMalloc(&acc, n_cols * 4));
    CHECK(hipMalloc(&acc_ref, n_cols * 4));
    hipLaunchKernelGGL(k_fill_csr, dim3(4096), dim3(256), 0, 0, n_rows, n_cols, K, indptr, indices, vals);
    const icv::ChainLaunch L(n_cols, 4, g_cu);
    uint16_t* lt; uint32_t* bounds;
    const int64_t n_blk = (n_rows + icv::kCcBlock - 1) / icv::kCcBlock;
    CHECK(hipMalloc(&lt, L.n_lines * 2));
    CHECK(hipMalloc(&bounds, (size_t)n_blk * (L.grid + 1) * icv::kCcBlock * 4));
    hipLaunchKernelGGL(k_line_tiles, dim3((L.grid + 255) / 256), dim3(256), 0, 0, L.n_lines, L.grid, lt);
    auto kern = icv::k_colchain_csr<float, false>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, icv::kChLdsFull));
    const float scale = 1.0f / (float)n_rows;
    auto bnd = [&] { hipLaunchKernelGGL(icv::k_csr_tile_bounds<false>, dim3((unsigned)n_blk), dim3(256), 0, 0, indptr, indices, (const int32_t*)nullptr, n_rows, lt, 2, L.grid, bounds); };
    auto chn = [&] { hipLaunchKernelGGL(kern, dim3(L.grid), dim3(icv::kChThreads), L.lds_bytes, 0, vals, indptr, indices, n_rows, (const int32_t*)nullptr, n_rows, n_cols, L.n_lines, L.lds_bytes, bounds, scale, acc); };
    CHECK(hipMemset(acc, 0, n_cols * 4));
    CHECK(hipMemset(acc_ref, 0, n_cols * 4));
    bnd(); chn();
    CHECK(hipDeviceSynchronize());
    int bad = 0;
    if (verify) {
        hipLaunchKernelGGL(k_naive_csr, dim3((n_cols + 63) / 64), dim3(64), 0, 0, vals, indptr, indices, n_rows, scale, acc_ref);
        CHECK(hipDeviceSynchronize());
        std::vector<float> a(n_cols), b(n_cols);
        CHECK(hipMemcpy(a.data(), acc, n_cols * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), acc_ref, n_cols * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < n_cols; ++c) if (std::memcmp(&a[c], &b[c], 4) != 0 && bad++ < 5) printf("  col %d: %.9g vs %.9g\n", c, a[c], b[c]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float tb = 0, tc = 0;
    for (int it = 0; it < 5; ++it) {
        float ms;
        hipEventRecord(e0); bnd(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tb += ms;
        hipEventRecord(e0); chn(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tc += ms;
    }
    CHECK(hipGetLastError());
    printf("csr rows %lld cols %d K %d grid %d lds %d: %s | bounds %.3f ms  chain %.3f ms  (%.0f GB/s of 8 B per entry)\n", (long long)n_rows, n_cols, K,
           L.grid, L.lds_bytes, verify ? (bad ? "MISMATCH" : "bit-exact") : "-", tb / 5, tc / 5, (double)n_rows * K * 8 / 1e9 / ((tb + tc) / 5) * 1e3);
    hipFree(indptr); hipFree(indices); hipFree(vals); hipFree(acc); hipFree(acc_ref); hipFree(lt); hipFree(bounds);
    return bad;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "csr") == 0) {
        const int64_t n = argc > 2 ? atoll(argv[2]) : 500000;
        const int K = argc > 3 ? atoi(argv[3]) : 1400;
        int bad = run_csr(3000, 20000, 1400, true);
        bad += run_csr(n, 20000, K, false);
        return bad != 0;
    }
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 100000;
    const int n_cols = argc > 2 ? atoi(argv[2]) : 20000;
    const int64_t ld = argc > 3 ? atoll(argv[3]) : n_cols;
    if (argc > 4) g_cu = atoi(argv[4]);
    if (argc > 5) g_lds = atoi(argv[5]);
    int bad = 0;
    bad += run<float>(n_rows, n_cols, ld, false);
    bad += run<float>(n_rows, n_cols, ld, true);
    bad += run<double>(n_rows / 2, n_cols, ld, false);
    bad += run<float>(1237, 20002, 20002, false);
    bad += run<float>(1237, 20003, 20003, true);
    bad += run<double>(999, 20001, 20001, false);
    bad += run<float>(61, 130, 131, false);
    bad += run<float>(1, 3, 3, false);
    bad += run<float>(29, 5, 5, false);
    bad += run<float>(5000, 60001, 60001, false);
    bad += run<double>(3000, 60001, 60001, true);
    bad += run<float>(4000, 9000, 9000, true);
    bad += run<double>(4001, 3000, 3000, false);
    return bad != 0;
}
