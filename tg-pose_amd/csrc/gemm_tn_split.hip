// dW = dy^T x of the large layers on the fp16 matrix cores WITHOUT transposed copies (round 3).
//
// The weight gradient reduces over the ROWS of both operands (dy (rows, N), x (rows, K), both row-major), while a
// v_mfma_f32_32x32x16_f16 operand wants eight consecutive reduction indices per lane.  Until now dy^T and x^T were written out
// (tgp_transpose_scaled, tgp_transpose_split_f16: 0.83 ms of a 19.4 ms training step, twice the operands' bytes through HBM) so that
// the NT tile kernel could run on them.  gfx950's ds_read_b64_tr_b16 delivers a 4 x 16 block of 16-bit LDS elements column-major:
// the operand tiles are staged as they lie in memory -- [row][column], converted to fp16 hi / lo planes (dy times its power-of-two
// scale) on the way -- and two transposed reads give a lane its eight reduction indices of one column.
//
// 256 x 256 output tile per 1024-thread workgroup (4 x 4 waves, 64 x 64 per wave, three split terms), 32 rows per step, two LDS
// stages (one barrier per step), the next step's global loads in flight behind the MFMAs; the reduction is cut into Z chunks
// (gridDim.y) whose partial tiles go to slabs that tgp_sum_slabs adds in order -- as the transposed form did.
#include "tgp_common.h"

typedef _Float16 ts_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ts_f16x4 __attribute__((ext_vector_type(4)));
typedef float ts_f32x4 __attribute__((ext_vector_type(4)));
typedef float ts_f32x16 __attribute__((ext_vector_type(16)));
typedef short ts_s4 __attribute__((ext_vector_type(4)));

#define TS_BM 32                               // reduction rows per step
#define TS_T 256                               // output tile edge
#define TS_ROWB (TS_T * 2 + 64)                // LDS row of a plane: 256 fp16 + 64 B (the four rows of a transposed read land 16 banks apart)
#define TS_PLANE (TS_BM * TS_ROWB)
#define TS_STAGE (4 * TS_PLANE)                // dy hi | dy lo | x hi | x lo

__device__ __forceinline__ void ts_split(const float4 v, uint2 &hi, uint2 &lo)
{
    const ts_f32x4 x = {v.x, v.y, v.z, v.w};
    const ts_f16x4 h = __builtin_convertvector(x, ts_f16x4);
    const ts_f32x4 rest = x - __builtin_convertvector(h, ts_f32x4);
    const ts_f16x4 l = __builtin_convertvector(rest, ts_f16x4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

// eight consecutive rows of one column of a [row][column] fp16 image as an MFMA operand fragment: two transposed reads
__device__ __forceinline__ ts_f16x8 ts_frag(const char *p)
{
    const ts_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts_s4 *)(p));
    const ts_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts_s4 *)(p + 4 * TS_ROWB));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(ts_f16x8, v);
}

__global__ __launch_bounds__(1024) void gemm_tn_split_kernel(const float *__restrict__ a, int lda, const float *__restrict__ b, int ldb,
                                                             int rows, int N, int K, const float *__restrict__ scale, int chunk,
                                                             float *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    const int tiles_k = (K + TS_T - 1) / TS_T;
    const int n0 = (blockIdx.x / tiles_k) * TS_T, k0 = (blockIdx.x % tiles_k) * TS_T;
    const int z = blockIdx.y;
    const int m_begin = z * chunk, m_end = min(rows, m_begin + chunk);
    const int steps = (m_end - m_begin + TS_BM - 1) / TS_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5, r = lane & 31;
    const float asc = scale[0];

    // staging: thread -> row tid / 64 (+ 16 on the second pass), four columns from (tid % 64) * 4
    const int srow = tid >> 6, scol = (tid & 63) * 4;
    float4 ra[2], rb[2];
    auto load = [&](int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = m_begin + s * TS_BM + srow + 16 * p;
            const bool mok = m < m_end;
            ra[p] = (mok && n0 + scol < N) ? *reinterpret_cast<const float4 *>(a + (int64_t)m * lda + n0 + scol) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[p] = (mok && k0 + scol < K) ? *reinterpret_cast<const float4 *>(b + (int64_t)m * ldb + k0 + scol) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&](int st) {
        char *base = ts_smem + st * TS_STAGE + scol * 2;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            char *row = base + (srow + 16 * p) * TS_ROWB;
            float4 v = ra[p];
            v.x *= asc, v.y *= asc, v.z *= asc, v.w *= asc;          // exact: a power of two
            uint2 hi, lo;
            ts_split(v, hi, lo);
            *reinterpret_cast<uint2 *>(row) = hi;
            *reinterpret_cast<uint2 *>(row + TS_PLANE) = lo;
            ts_split(rb[p], hi, lo);
            *reinterpret_cast<uint2 *>(row + 2 * TS_PLANE) = hi;
            *reinterpret_cast<uint2 *>(row + 3 * TS_PLANE) = lo;
        }
    };

    ts_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transposed read: lane 4 q + p of a 16-lane group supplies the address of block row q, columns 4 p .. 4 p + 3, and receives
    // column (lane % 16) of the four rows.  Group g = lane / 16: columns 16 (g % 2) .. + 15 of the wave's 32, reduction rows 8 (g / 2) ..
    const int g16 = lane >> 4, i16 = lane & 15;
    const int frow = 8 * (g16 >> 1) + (i16 >> 2);
    const int fcol = 16 * (g16 & 1) + 4 * (i16 & 3);
    const int aoff = frow * TS_ROWB + (wm * 64 + fcol) * 2;
    const int boff = 2 * TS_PLANE + frow * TS_ROWB + (wn * 64 + fcol) * 2;

    if (steps > 0) load(0);
    for (int s = 0; s < steps; ++s) {
        const int st = s & 1;
        store(st);
        __syncthreads();
        if (s + 1 < steps) load(s + 1);
        const char *sb = ts_smem + st * TS_STAGE;
#pragma unroll
        for (int ks = 0; ks < TS_BM / 16; ++ks) {
            ts_f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = ts_frag(sb + aoff + ks * 16 * TS_ROWB + i * 64);
                al[i] = ts_frag(sb + aoff + ks * 16 * TS_ROWB + i * 64 + TS_PLANE);
                bh[i] = ts_frag(sb + boff + ks * 16 * TS_ROWB + i * 64);
                bl[i] = ts_frag(sb + boff + ks * 16 * TS_ROWB + i * 64 + TS_PLANE);
            }
            // smallest terms first, as in the NT tile kernel
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    // accumulator element e of lane (r, h): row 4 h + (e & 3) + 8 (e >> 2), column r of the 32 x 32 block
    float *out = parts + (int64_t)z * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = n0 + wm * 64 + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
                if (row < N && col < K) out[(int64_t)row * K + col] = acc[i][j][e];
            }
        }
}

// parts (Z, N, K): partial products of the Z row chunks (chunk rows each), to be added and unscaled by tgp_sum_slabs.
// N % 4 == 0, K % 4 == 0, lda % 4 == 0, ldb % 4 == 0, a / b 16-byte aligned, else TGP_EUNSUPPORTED.
extern "C" int tgp_gemm_tn_split(const float *a, int lda, const float *b, int ldb, int rows, int N, int K, const float *scale, int Z,
                                 int chunk, float *parts, tgp_stream_t stream)
{
    TGP_REQUIRE(a && b && scale && parts && rows > 0 && N > 0 && K > 0 && Z > 0 && chunk > 0 && lda >= N && ldb >= K);
    TGP_REQUIRE((int64_t)Z * chunk >= rows);
    if ((N & 3) || (K & 3) || (lda & 3) || (ldb & 3) || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15))
        return TGP_EUNSUPPORTED;
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(gemm_tn_split_kernel), 2 * TS_STAGE)) return e;
    const dim3 grid(tgp_cdiv(N, TS_T) * tgp_cdiv(K, TS_T), Z);
    hipLaunchKernelGGL(gemm_tn_split_kernel, grid, dim3(1024), 2 * TS_STAGE, tgp_hs(stream), a, lda, b, ldb, rows, N, K, scale, chunk, parts);
    return TGP_LAUNCH_RESULT();
}
