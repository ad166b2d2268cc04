// Issue cost of single VALU / MFMA instructions on gfx950 (round 2: which instructions of the smoothing kernel are
// expensive?).  Stand-alone, not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench_ops.bin tools/microbench_ops.hip
//
// Every wavefront issues N x 8 instances of one instruction on 8 independent register sets; WAVES wavefronts per
// workgroup, one workgroup per CU.  Printed: shader cycles per wave-instruction PER SIMD (elapsed cycles of the
// workgroup / (N * 8 * wavefronts per SIMD)) -- the reciprocal throughput of the pipe the instruction uses.
// The last section runs a float64-VALU wavefront next to a float64-MFMA wavefront on every SIMD: do they share a pipe?
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

typedef unsigned long long u64;
typedef double double4_t __attribute__((ext_vector_type(4)));

#define TIMED_LOOP(BODY)                                              \
    __syncthreads();                                                  \
    const u64 t0 = __builtin_amdgcn_s_memtime();                      \
    for (int i = 0; i < n; ++i) {                                     \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) { BODY }        \
    }                                                                 \
    __syncthreads();                                                  \
    const u64 t1 = __builtin_amdgcn_s_memtime();                      \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;

#define OP_F32_2(NAME, ASM)                                                                        \
    __global__ void NAME(float* out, u64* cyc, int n) {                                            \
        float a[8], b = 1.0f + threadIdx.x * 1e-7f;                                                \
        for (int k = 0; k < 8; ++k) a[k] = k + threadIdx.x;                                        \
        TIMED_LOOP(asm volatile(ASM : "+v"(a[k]) : "v"(b));)                                       \
        float s = 0;                                                                               \
        for (int k = 0; k < 8; ++k) s += a[k];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }
#define OP_F64_2(NAME, ASM)                                                                        \
    __global__ void NAME(float* out, u64* cyc, int n) {                                            \
        double a[8], b = 1.0 + threadIdx.x * 1e-9;                                                 \
        for (int k = 0; k < 8; ++k) a[k] = k + threadIdx.x;                                        \
        TIMED_LOOP(asm volatile(ASM : "+v"(a[k]) : "v"(b));)                                       \
        double s = 0;                                                                              \
        for (int k = 0; k < 8; ++k) s += a[k];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                                     \
    }

OP_F32_2(k_add_f32, "v_add_f32 %0, %0, %1")
OP_F32_2(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
OP_F32_2(k_med3_f32, "v_med3_f32 %0, %0, %1, %1")
OP_F32_2(k_lshl_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
OP_F32_2(k_and_b32, "v_and_b32 %0, %0, %1")
OP_F32_2(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP_F32_2(k_add_u32_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OP_F32_2(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP_F64_2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OP_F64_2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
OP_F64_2(k_add_f64, "v_add_f64 %0, %0, %1")
OP_F64_2(k_mul_f64, "v_mul_f64 %0, %0, %1")
OP_F64_2(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
OP_F64_2(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")

__global__ void k_cvt_f64_f32(float* out, u64* cyc, int n) {
    double a[8];
    float f[8];
    for (int k = 0; k < 8; ++k) f[k] = k + threadIdx.x;
    TIMED_LOOP(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));)
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_cvt_f32_f64(float* out, u64* cyc, int n) {
    double a[8];
    float f[8];
    for (int k = 0; k < 8; ++k) a[k] = k + threadIdx.x;
    TIMED_LOOP(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));)
    float s = 0;
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_readlane(float* out, u64* cyc, int n) {
    int a[8], s[8];
    for (int k = 0; k < 8; ++k) a[k] = k + threadIdx.x;
    TIMED_LOOP(asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[k]) : "v"(a[k]));)
    int r = 0;
    for (int k = 0; k < 8; ++k) r += s[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)r;
}
// float64 matrix instructions: 16x16x4 (4 accumulator registers of 2 dwords per lane) and 4x4x4 x 4 blocks
__global__ void k_mfma_f64_16(float* out, u64* cyc, int n) {
    double4_t acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = double4_t{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k & 3], 0, 0, 0);
    }
    __syncthreads();
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
}
__global__ void k_mfma_f64_4(float* out, u64* cyc, int n) {
    double acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[k], 0, 0, 0);
    }
    __syncthreads();
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    double s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
// wavefronts 0..3 (one per SIMD): float64 VALU FMAs; wavefronts 4..7: float64 MFMA 4x4x4; both timed separately
__global__ void k_mix_f64(float* out, u64* cyc, int n, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_valu = (mode == 0) || (mode == 2 && wave < 4);
    const bool do_mfma = (mode == 1) || (mode == 2 && wave >= 4);
    double a[8], b = 1.0 + threadIdx.x * 1e-9;
    for (int k = 0; k < 8; ++k) a[k] = k;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    if (do_valu) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
        }
    }
    if (do_mfma) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, a[k], 0, 0, 0);
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

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
    std::printf("device %s, %d CUs\n", prop.gcnArchName, n_cu);
    float* d_out;
    u64* d_cyc;
    CHECK(hipMalloc((void**)&d_out, 64 << 20));
    CHECK(hipMalloc((void**)&d_cyc, 1 << 20));
    struct Op {
        const char* name;
        void (*k)(float*, u64*, int);
    };
    const Op ops[] = {
        {"v_add_f32", k_add_f32},         {"v_fma_f32", k_fma_f32},       {"v_med3_f32", k_med3_f32},
        {"v_lshlrev_b32_sdwa", k_lshl_sdwa}, {"v_and_b32", k_and_b32},    {"v_mov_b32_dpp", k_mov_dpp},
        {"v_add_u32_dpp", k_add_u32_dpp}, {"v_cndmask_b32", k_cndmask},   {"v_pk_add_f32", k_pk_add_f32},
        {"v_pk_fma_f32", k_pk_fma_f32},   {"v_add_f64", k_add_f64},       {"v_mul_f64", k_mul_f64},
        {"v_fma_f64", k_fma_f64},         {"v_cmp_lt_f64", k_cmp_f64},    {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_readlane_b32", k_readlane}, {"v_mfma_f64_16x16x4", k_mfma_f64_16},
        {"v_mfma_f64_4x4x4_4b", k_mfma_f64_4},
    };
    const int n = 512;
    for (const Op& op : ops) {
        std::printf("%-22s", op.name);
        for (int threads : {256, 512, 1024}) {
            hipLaunchKernelGGL(op.k, dim3(n_cu), dim3(threads), 0, 0, d_out, d_cyc, n);
            CHECK(hipDeviceSynchronize());
            const double c = mean_cycles(d_cyc, n_cu);
            std::printf("  %d w/SIMD: %6.2f", threads / 256, c / (n * 8.0 * (threads / 256)));
        }
        std::printf("   cycles per wave-instruction per SIMD\n");
    }
    const char* mode_name[3] = {"8 wavefronts all v_fma_f64", "8 wavefronts all mfma_f64_4x4x4",
                                "4 wavefronts v_fma_f64 + 4 wavefronts mfma_f64_4x4x4"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_mix_f64, dim3(n_cu), dim3(512), 0, 0, d_out, d_cyc, n, mode);
        CHECK(hipDeviceSynchronize());
        std::vector<u64> h(n_cu * 8);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
        double lo = 0, hi = 0;
        for (int b = 0; b < n_cu; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
        std::printf("mix: %-55s wavefronts 0-3: %7.2f  wavefronts 4-7: %7.2f cycles per own instruction\n", mode_name[mode],
                    lo / (n_cu * 4) / (n * 8.0), hi / (n_cu * 4) / (n * 8.0));
    }
    std::printf("done\n");
    return 0;
}
