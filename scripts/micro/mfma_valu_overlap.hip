// Micro-benchmark (gfx950): how many independent vector instructions fit behind a DEPENDENT v_mfma_f32_32x32x16_f16 before they cost time?
// One wave per SIMD (256 threads, launch_bounds(256, 1)) or two (NW = 2); a chain of MFMAs into the same accumulator with NV independent
// v_fma_f32 between consecutive MFMAs, fenced by sched_barrier.  Prints cycles per MFMA for NV = 0 .. 16.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu scripts/micro/mfma_valu_overlap.hip && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.1f + i;
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) v[j] = __builtin_fmaf(v[j], m, c);                      // independent chains, one op each per gap
                else v[j & 1] = __builtin_fmaf(v[j & 1], m, c);                       // two dependent chains
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int KIND>
static void run(int wgs_per_cu_hint, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256 * wgs_per_cu_hint), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256 * wgs_per_cu_hint), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  NV = %2d  %s  %6.1f cycles per MFMA\n", NV, KIND ? "2 dependent chains " : "independent ops    ", (double)c / (iters * 8.0));
}

// two waves per SIMD: waves 0-3 run the MFMA chain alone, waves 4-7 nothing but v_fma_f32 (16 independent chains); do they overlap?
template <int MODE>   // 0: both kinds, 1: MFMA waves only, 2: vector waves only
__global__ __launch_bounds__(512, 1) void k2(float *out, unsigned long long *cyc, int iters)
{
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.1f + i;
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (MODE != 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    } else if (MODE != 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], m, c);      // 64 vector instructions per iteration
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

template <int MODE>
static void run2(float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long c[2] = {0, 0};
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("  %s: MFMA wave %6.1f cycles per MFMA (8 per iteration), vector wave %6.2f cycles per v_fma_f32 (64 per iteration)\n",
           MODE == 0 ? "both kinds of waves   " : MODE == 1 ? "MFMA waves only       " : "vector waves only     ", (double)c[0] / (iters * 8.0),
           (double)c[1] / (iters * 64.0));
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4 * 2);
    hipMalloc(&cyc, 16);
    printf("dependent v_mfma_f32_32x32x16_f16 chain, NV v_fma_f32 behind each MFMA, one wave per SIMD, every CU busy:\n");
    run<0, 0>(1, out, cyc); run<2, 0>(1, out, cyc); run<4, 0>(1, out, cyc); run<5, 0>(1, out, cyc); run<6, 0>(1, out, cyc); run<7, 0>(1, out, cyc);
    run<8, 0>(1, out, cyc); run<10, 0>(1, out, cyc); run<12, 0>(1, out, cyc); run<16, 0>(1, out, cyc);
    run<4, 1>(1, out, cyc); run<8, 1>(1, out, cyc);
    printf("two waves per SIMD, one kind of instruction each:\n");
    run2<1>(out, cyc); run2<2>(out, cyc); run2<0>(out, cyc);
    return 0;
}
