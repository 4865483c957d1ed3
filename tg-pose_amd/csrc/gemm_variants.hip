// Development-only micro-benchmark kernels for the fp32 MFMA GEMM main loop (not part of the public ABI,
// not used by the product path).  scripts/gemm_variants.py times them against each other in one process
// to decide what the production kernel in gemm.hip should look like.  Plain epilogue (C = A W^T), M, N
// multiples of the tile, K multiple of BK.
#include "tgp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct VarParams {
    const float *A;
    const float *W;
    float *C;
    int lda, ldw, ldc, M, N, K;
    unsigned long long *stamps; // per block: {memtime start, memtime end, realtime start, realtime end}
};

template <int BM, int BN, int NWM, int NWN, int BK, bool DBUF, int MINW>
__global__ __launch_bounds__(64 * NWM * NWN, MINW) void gemm_var_kernel(VarParams p)
{
    constexpr int THREADS = 64 * NWM * NWN;
    constexpr int LD = BK + 4;
    constexpr int WTM = BM / NWM, WTN = BN / NWN; // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int F4 = BK / 4;                    // float4 per row
    constexpr int RPP = THREADS / F4;             // rows per staging pass
    constexpr int PA = (BM + RPP - 1) / RPP, PW = (BN + RPP - 1) / RPP; // a partial pass when the tile has fewer rows
    constexpr int BUF = (BM + BN) * LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    unsigned long long t0 = 0, r0t = 0;
    if (p.stamps && threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0t = __builtin_amdgcn_s_memrealtime();
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kq = tid % F4, r0 = tid / F4;
    float4 ra[PA], rw[PW];

    auto load_tile = [&](int kt) {
        const int kcol = kt * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (r0 + RPP * i < BM)
                ra[i] = *reinterpret_cast<const float4 *>(p.A + (int64_t)(m0 + r0 + RPP * i) * p.lda + kcol);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            if (r0 + RPP * i < BN)
                rw[i] = *reinterpret_cast<const float4 *>(p.W + (int64_t)(n0 + r0 + RPP * i) * p.ldw + kcol);
    };
    auto store_tile = [&](int buf) {
        float *as = smem + buf * BUF;
        float *ws = as + BM * LD;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (r0 + RPP * i < BM) *reinterpret_cast<float4 *>(as + (r0 + RPP * i) * LD + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < PW; ++i)
            if (r0 + RPP * i < BN) *reinterpret_cast<float4 *>(ws + (r0 + RPP * i) * LD + kq * 4) = rw[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int numK = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < numK; ++kt) {
        const bool more = (kt + 1) < numK;
        if (more) load_tile(kt + 1);
        const int cur = DBUF ? (kt & 1) : 0;
        const float *as = smem + cur * BUF + (wm * WTM + r) * LD;
        const float *ws = smem + cur * BUF + BM * LD + (wn * WTN + r) * LD;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LD + kk * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(ws + j * 32 * LD + kk * 8 + h * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (DBUF) {
            if (more) store_tile((kt + 1) & 1);
            __syncthreads();
        } else {
            __syncthreads();
            if (more) store_tile(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + r;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                p.C[(int64_t)row * p.ldc + col] = acc[i][j][e];
            }
    }
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long *s = p.stamps + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        s[0] = t0;
        s[1] = __builtin_amdgcn_s_memtime();
        s[2] = r0t;
        s[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int BM, int BN, int NWM, int NWN, int BK, bool DBUF, int MINW>
static int launch_var(const VarParams &p, hipStream_t stream)
{
    if (p.M % BM || p.N % BN || p.K % BK) return TGP_EUNSUPPORTED;
    const size_t lds = (size_t)(DBUF ? 2 : 1) * (BM + BN) * (BK + 4) * sizeof(float);
    auto fn = gemm_var_kernel<BM, BN, NWM, NWN, BK, DBUF, MINW>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fn, dim3(p.N / BN, p.M / BM), dim3(64 * NWM * NWN), lds, stream, p);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_debug_gemm_variant(const float *A, const float *W, float *C, int M, int N, int K, int ld, int variant,
                                      unsigned long long *stamps, tgp_stream_t stream)
{
    VarParams p = {A, W, C, ld, ld, N, M, N, K, stamps};
    hipStream_t s = tgp_hs(stream);
    switch (variant) {
    case 0: return launch_var<128, 128, 2, 2, 32, true, 1>(p, s);   // production shape: 2 x 73.7 KB per CU
    case 1: return launch_var<128, 128, 2, 2, 32, false, 1>(p, s);  // single LDS buffer: 4 x 36.9 KB per CU
    case 2: return launch_var<128, 128, 2, 2, 16, true, 1>(p, s);   // BK 16, double buffer: 3-4 per CU
    case 3: return launch_var<256, 128, 4, 2, 32, true, 1>(p, s);   // 512 threads, 110 KB, 1 per CU
    case 4: return launch_var<128, 256, 2, 4, 32, true, 1>(p, s);   // 512 threads
    case 5: return launch_var<256, 128, 4, 2, 32, false, 1>(p, s);  // 512 threads single buffer: 2 per CU
    case 6: return launch_var<128, 128, 2, 2, 64, false, 1>(p, s);  // BK 64 single buffer: 69.6 KB, 2 per CU
    case 7: return launch_var<256, 256, 4, 4, 32, false, 1>(p, s);  // 1024 threads, 73.7 KB single buffer: 2 per CU
    case 8: return launch_var<128, 128, 2, 2, 16, false, 1>(p, s);  // BK 16 single buffer: 20 KB
    case 9: return launch_var<256, 256, 4, 4, 32, true, 1>(p, s);   // 1024 threads double buffer: 147 KB, 1 per CU
    case 10: return launch_var<256, 256, 4, 2, 32, false, 1>(p, s); // 512 threads, wave tile 64x128
    case 11: return launch_var<256, 256, 4, 4, 16, true, 1>(p, s);  // 1024 threads, BK16 dbuf: 82 KB
    case 12: return launch_var<256, 128, 4, 2, 16, true, 1>(p, s);  // 512 threads BK16 dbuf: 61 KB, 2 per CU
    case 13: return launch_var<256, 256, 4, 4, 16, false, 1>(p, s); // 1024 threads BK16 single: 41 KB, 2 per CU (thread cap)
    case 14: return launch_var<128, 128, 4, 4, 16, true, 1>(p, s);  // 1024 threads on a 128x128 tile (tail candidate)
    case 15: return launch_var<256, 128, 4, 2, 16, false, 1>(p, s); // 512 threads BK16 single
    case 16: return launch_var<128, 256, 2, 4, 16, true, 1>(p, s);  // 512 threads BK16 dbuf
    case 17: return launch_var<256, 256, 8, 2, 16, true, 1>(p, s);  // 1024 threads, wave tile 32x128
    case 18: return launch_var<256, 256, 2, 8, 16, true, 1>(p, s);  // 1024 threads, wave tile 128x32
    case 19: return launch_var<128, 128, 2, 2, 16, true, 2>(p, s);  // 256 threads BK16 dbuf, min 2 waves/SIMD
    case 20: return launch_var<256, 128, 4, 4, 16, true, 1>(p, s);  // 1024 threads, wave tile 64x32
    default: return TGP_EUNSUPPORTED;
    }
}
