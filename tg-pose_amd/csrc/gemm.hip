// Dense per-point layers (Conv1d kernel-1 / Linear on channel-last rows) for gfx950, fp32 in /
// fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//
//   C[m,n] = act((sum_k A[m,k] W[n,k] + bias[n] + rowbias[obj(m),n] + res1[m,n] + res2[m,n]) * scale[n] + shift[n])
//
// Tiled kernel: 256 threads = 4 waves in a 2x2 grid, block tile BM x BN (128x128 or 64x64), each
// wave (BM/2)x(BN/2) as 32x32 MFMA tiles, BK = 32.  A and W are both K-contiguous, so both are
// staged global -> registers -> LDS as float4 rows padded to 36 floats (conflict-free
// ds_read_b128 of 4 consecutive k per lane); the loads of tile t+1 are in flight while tile t is
// multiplied; one barrier per K-tile, two LDS buffers.
//   * default k order inside a 8-wide group is (0,4,1,5,2,6,3,7): lane-half h supplies k = 4h+t
//     to MFMA step t.  Fine for layers compared at 1e-4.
//   * NATURAL_K: ascending k (lane-half h supplies k = 2s+h to step s) -- bit-identical to a CPU
//     sgemm FMA chain; used for the feature-space distance matrix, whose rounding decides the
//     neighbour order (DIST epilogue: fl(fl(-2*acc + q_col) + q_row)).
// Skinny kernel: M <= 32 rows (per-object vectors: PH_Predictor linears, head conv3/conv4):
// weight-streaming, one wave per 4 output columns, lanes split K, butterfly reduction.
#include "tgp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float *A;
    const float *W;
    float *C;
    int lda, ldw, ldc, M, N, K;
    const float *bias;
    const float *rowbias;
    int ldrb, rows_per_obj;
    const float *res1;
    int ldr1;
    const float *res2;
    int ldr2;
    const float *scale;
    const float *shift;
    const float *slope_vec;
    int act;
    float slope;
    uint32_t *cm;
    int ldcm;
    int cm_cols;     // colmax covers columns [0, cm_cols)
    int c_col0;      // C is stored for columns >= c_col0, at C[row * ldc + col - c_col0]
    int batch;
    int64_t sA, sW, sC, sV, sCM; // per-batch element strides (A, W, C, per-column vectors, colmax keys)
    const float *qrow;           // DIST epilogue
    const float *qcol;
    int64_t sq;
    // tile schedule of the main kernel: per batch, M-tile rows [0, mt_big) use 128x128 tiles, the rest 64x64
    int mt_big, tiles_n_big, tiles_big, tiles_m_small, tiles_n_small;
};

#define GEMM_BK 32
#define GEMM_LD 36

// One BM x BN output tile at (m0, n0) of batch z.  256 threads.
template <int BM, int BN, bool NATURAL_K, bool DIST>
__device__ __forceinline__ void gemm_tile(const GemmParams &p, const int m0, const int n0, const int z, float *smem)
{
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PA = BM / 32, PW = BN / 32;
    constexpr int BUF = (BM + BN) * GEMM_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const float *A = p.A + (int64_t)z * p.sA;
    const float *W = p.W + (int64_t)z * p.sW;

    const int kq = tid & 7, r0 = tid >> 3;
    float4 ra[PA], rw[PW];

    // Guards without branches: every lane loads from a clamped (always valid) address, then selects.
    auto load_tile = [&](int kt) {
        const int kcol = kt * GEMM_BK + kq * 4;
        const bool kok = kcol < p.K;
        const int kc = kok ? kcol : 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = m0 + r0 + 32 * i;
            const bool ok = kok && row < p.M;
            const float4 v = *reinterpret_cast<const float4 *>(A + (int64_t)(row < p.M ? row : p.M - 1) * p.lda + kc);
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = n0 + r0 + 32 * i;
            const bool ok = kok && row < p.N;
            const float4 v = *reinterpret_cast<const float4 *>(W + (int64_t)(row < p.N ? row : p.N - 1) * p.ldw + kc);
            rw[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float *as = smem + buf * BUF;
        float *ws = as + BM * GEMM_LD;
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<float4 *>(as + (r0 + 32 * i) * GEMM_LD + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < PW; ++i) *reinterpret_cast<float4 *>(ws + (r0 + 32 * i) * GEMM_LD + kq * 4) = rw[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int numK = (p.K + GEMM_BK - 1) / GEMM_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < numK; ++kt) {
        const bool more = (kt + 1) < numK;
        if (more) load_tile(kt + 1);
        const float *as = smem + (kt & 1) * BUF + (wm * (BM / 2) + r) * GEMM_LD;
        const float *ws = smem + (kt & 1) * BUF + BM * GEMM_LD + (wn * (BN / 2) + r) * GEMM_LD;
        if constexpr (NATURAL_K) {
#pragma unroll
            for (int s = 0; s < GEMM_BK / 2; ++s) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = as[i * 32 * GEMM_LD + 2 * s + h];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = ws[j * 32 * GEMM_LD + 2 * s + h];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < GEMM_BK / 8; ++kk) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(as + i * 32 * GEMM_LD + kk * 8 + h * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(ws + j * 32 * GEMM_LD + kk * 8 + h * 4);
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
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + r;
        const bool colok = col < p.N;
        if constexpr (DIST) {
            const float *qrow = p.qrow + (int64_t)z * p.sq;
            const float *qcol = p.qcol + (int64_t)z * p.sq;
            float *C = p.C + (int64_t)z * p.sC;
            const float qc = colok ? qcol[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (colok && row < p.M) {
                        const float t1 = acc[i][j][e] * -2.0f;
                        const float t2 = t1 + qc;
                        C[(int64_t)row * p.ldc + col] = t2 + qrow[row];
                    }
                }
        } else {
            const int64_t vo = (int64_t)z * p.sV;
            const float bias = (colok && p.bias) ? p.bias[vo + col] : 0.f;
            const float sc = (colok && p.scale) ? p.scale[vo + col] : 1.f;
            const float sh = (colok && p.shift) ? p.shift[vo + col] : 0.f;
            const float slope = (colok && p.slope_vec) ? p.slope_vec[vo + col] : p.slope;
            float *C = (p.C && col >= p.c_col0) ? p.C + (int64_t)z * p.sC + (col - p.c_col0) : nullptr;
            uint32_t *cm = (p.cm && col < p.cm_cols) ? p.cm + (int64_t)z * p.sCM + col : nullptr;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * h;
                int obj = 0, bound = 0x7fffffff;
                if (p.rowbias || p.cm) {
                    obj = rbase / p.rows_per_obj;
                    bound = (obj + 1) * p.rows_per_obj;
                }
                uint32_t run_key = 0;
                int run_obj = -1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rbase + (e & 3) + 8 * (e >> 2);
                    if (!(colok && row < p.M)) continue;
                    while (row >= bound) {
                        ++obj;
                        bound += p.rows_per_obj;
                    }
                    float v = acc[i][j][e] + bias;
                    if (p.rowbias) v += p.rowbias[(int64_t)obj * p.ldrb + col];
                    if (p.res1) v += p.res1[(int64_t)row * p.ldr1 + col];
                    if (p.res2) v += p.res2[(int64_t)row * p.ldr2 + col];
                    if (p.scale) v = v * sc + sh;
                    if (p.act == 1) v = v > 0.f ? v : v * slope;
                    if (C) C[(int64_t)row * p.ldc] = v;
                    if (cm) {
                        const uint32_t key = tgp_float_key(v);
                        if (obj != run_obj) {
                            if (run_obj >= 0) atomicMax(cm + (int64_t)run_obj * p.ldcm, run_key);
                            run_obj = obj;
                            run_key = key;
                        } else {
                            run_key = key > run_key ? key : run_key;
                        }
                    }
                }
                if (cm && run_obj >= 0) atomicMax(cm + (int64_t)run_obj * p.ldcm, run_key);
            }
        }
    }
}

// Main kernel: blocks [0, tiles_big) take 128x128 tiles (N fastest, then M, then batch); the remaining
// blocks cover the leftover M rows with 64x64 tiles.  M = B*1028 is 257 tiles of 128 (257 is prime), so a
// plain grid leaves e.g. 2056 tiles for 512 resident workgroups = 4.02 rounds -> 5; giving the last
// M-tile rows to quarter-size tiles, dispatched last, fills the tail round instead (4.25).
__global__ __launch_bounds__(256) void gemm_main_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int L = blockIdx.x;
    if (L < p.tiles_big) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<128, 128, false, false>(p, (L / p.tiles_n_big) * 128, (L % p.tiles_n_big) * 128, z, smem);
    } else {
        L -= p.tiles_big;
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<64, 64, false, false>(p, p.mt_big * 128 + (L / p.tiles_n_small) * 64, (L % p.tiles_n_small) * 64, z, smem);
    }
}

template <bool NAT, bool DIST>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_tile<64, 64, NAT, DIST>(p, blockIdx.y * 64, blockIdx.x * 64, blockIdx.z, smem);
}

// ---------------------------------------------------------------------------------------------------
// M <= 32 rows (per-object vectors): weight streaming.  A workgroup owns 4 output columns; its 4 waves
// split K (wave w takes k = 1024*i + 256*w + 4*lane), so N waves are in flight for N columns and every
// W element is read exactly once with 16-byte lane loads.  Per wave a 6-step butterfly sums the lanes,
// then the 4 waves are combined through LDS in fixed order (deterministic).
#define SKINNY_COLS 4
__global__ __launch_bounds__(256) void skinny_gemm_kernel(GemmParams p)
{
    __shared__ float red[4][SKINNY_COLS][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = blockIdx.x * SKINNY_COLS;
    float acc[SKINNY_COLS][32];
#pragma unroll
    for (int c = 0; c < SKINNY_COLS; ++c)
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[c][m] = 0.f;

    for (int k0 = wave * 256 + lane * 4; k0 < p.K; k0 += 1024) {
        float4 w[SKINNY_COLS];
#pragma unroll
        for (int c = 0; c < SKINNY_COLS; ++c)
            w[c] = (nb + c < p.N) ? *reinterpret_cast<const float4 *>(p.W + (int64_t)(nb + c) * p.ldw + k0)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (m < p.M) {
                const float4 a = *reinterpret_cast<const float4 *>(p.A + (int64_t)m * p.lda + k0);
#pragma unroll
                for (int c = 0; c < SKINNY_COLS; ++c) {
                    float s = acc[c][m];
                    s = fmaf(a.x, w[c].x, s);
                    s = fmaf(a.y, w[c].y, s);
                    s = fmaf(a.z, w[c].z, s);
                    s = fmaf(a.w, w[c].w, s);
                    acc[c][m] = s;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < SKINNY_COLS; ++c) {
        float mine = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float s = acc[c][m];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == m) mine = s;
        }
        if (lane < 32) red[wave][c][lane] = mine;
    }
    __syncthreads();
    // 128 threads finish: thread -> (column c, row m)
    if (threadIdx.x < SKINNY_COLS * 32) {
        const int c = threadIdx.x >> 5, m = threadIdx.x & 31;
        const int col = nb + c;
        if (m < p.M && col < p.N) {
            float v = ((red[0][c][m] + red[1][c][m]) + red[2][c][m]) + red[3][c][m];
            v += (p.bias ? p.bias[col] : 0.f);
            if (p.scale) v = v * p.scale[col] + (p.shift ? p.shift[col] : 0.f);
            if (p.act == 1) v = v > 0.f ? v : v * p.slope;
            p.C[(int64_t)m * p.ldc + col] = v;
        }
    }
}

static int set_lds_limit(const void *fn, size_t lds)
{
    if (lds <= 64 * 1024) return 0; // > 64 KiB of dynamic LDS needs the opt-in
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

static int resident_slots(void)
{
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus; // two 73.7 KB workgroups per CU
    }
    return slots;
}

template <bool NAT, bool DIST>
static int launch_small(const GemmParams &p, hipStream_t stream)
{
    const size_t lds = (size_t)2 * (64 + 64) * GEMM_LD * sizeof(float);
    hipLaunchKernelGGL((gemm_small_kernel<NAT, DIST>), dim3(tgp_cdiv(p.N, 64), tgp_cdiv(p.M, 64), p.batch), dim3(256), lds,
                       stream, p);
    return TGP_LAUNCH_RESULT();
}

// Split the M-tile rows between 128x128 tiles and a tail of 64x64 tiles so that the tail round is filled.
static int launch_main(GemmParams &p, hipStream_t stream)
{
    const int S = resident_slots();
    const int tiles_m = tgp_cdiv(p.M, 128), tiles_n = tgp_cdiv(p.N, 128);
    const int64_t T = (int64_t)tiles_m * tiles_n * p.batch;
    const double small_cost = 0.3; // a 64x64 tile in units of a 128x128 tile (measured ~0.27-0.3)
    // candidate (a): every row on big tiles
    int best_mt = tiles_m;
    double best = (double)((T + S - 1) / S);
    // candidate (b): big tiles for whole M-tile rows up to the last full round, small tiles for the rest
    const int64_t cap = (T / S) * S;
    const int mt_b = (int)(cap / ((int64_t)tiles_n * p.batch));
    if (mt_b < tiles_m) {
        const int64_t tb = (int64_t)mt_b * tiles_n * p.batch;
        const int rows_left = p.M - mt_b * 128;
        const int64_t ts = (int64_t)tgp_cdiv(rows_left, 64) * tgp_cdiv(p.N, 64) * p.batch;
        const double est = (double)((tb + S - 1) / S) + small_cost * (double)((ts + S - 1) / S);
        if (est < best) best = est, best_mt = mt_b;
    }
    p.mt_big = best_mt;
    p.tiles_n_big = tiles_n;
    p.tiles_big = best_mt * tiles_n * p.batch;
    const int rows_left = p.M - best_mt * 128;
    p.tiles_m_small = rows_left > 0 ? tgp_cdiv(rows_left, 64) : 0;
    p.tiles_n_small = tgp_cdiv(p.N, 64);
    const int total = p.tiles_big + p.tiles_m_small * p.tiles_n_small * p.batch;
    const size_t lds = (size_t)2 * (128 + 128) * GEMM_LD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        const int e = set_lds_limit(reinterpret_cast<const void *>(gemm_main_kernel), lds);
        if (e) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_main_kernel, dim3(total), dim3(256), lds, stream, p);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_gemm_f32(const tgp_gemm_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->A && a->W && (a->C || a->colmax_keys));
    TGP_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0);
    TGP_REQUIRE((a->K & 3) == 0 && (a->lda & 3) == 0 && (a->ldw & 3) == 0 && a->lda >= a->K && a->ldw >= a->K);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(a->A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->W) & 15) == 0);
    TGP_REQUIRE(a->c_col0 >= 0 && a->c_col0 <= a->N && (!a->C || a->ldc >= a->N - a->c_col0));
    TGP_REQUIRE(!(a->rowbias || a->colmax_keys) || a->rows_per_obj > 0);
    TGP_REQUIRE(a->act == 0 || a->act == 1);
    TGP_REQUIRE(a->batch >= 0 && a->cm_cols >= 0 && a->cm_cols <= a->N);
    GemmParams p = {};
    p.A = a->A, p.W = a->W, p.C = a->C;
    p.lda = a->lda, p.ldw = a->ldw, p.ldc = a->ldc, p.M = a->M, p.N = a->N, p.K = a->K;
    p.bias = a->bias, p.rowbias = a->rowbias, p.ldrb = a->ldrb, p.rows_per_obj = a->rows_per_obj > 0 ? a->rows_per_obj : 1;
    p.res1 = a->res1, p.ldr1 = a->ldr1, p.res2 = a->res2, p.ldr2 = a->ldr2;
    p.scale = a->scale, p.shift = a->shift, p.slope_vec = a->slope_vec, p.act = a->act, p.slope = a->slope;
    p.cm = a->colmax_keys, p.ldcm = a->ldcm;
    p.cm_cols = a->cm_cols > 0 ? a->cm_cols : a->N;
    p.c_col0 = a->c_col0;
    p.batch = a->batch > 0 ? a->batch : 1;
    p.sA = a->batch_stride_a, p.sW = a->batch_stride_w, p.sC = a->batch_stride_c, p.sV = a->batch_stride_vec;
    p.sCM = a->batch_stride_colmax;
    const bool plain = !a->rowbias && !a->res1 && !a->res2 && !a->colmax_keys && !a->slope_vec && a->c_col0 == 0 &&
                       p.batch == 1;
    if (a->M <= 32 && a->C && plain) {
        hipLaunchKernelGGL(skinny_gemm_kernel, dim3(tgp_cdiv(a->N, SKINNY_COLS)), dim3(256), 0, tgp_hs(stream), p);
        return TGP_LAUNCH_RESULT();
    }
    const int64_t big_tiles = (int64_t)tgp_cdiv(a->M, 128) * tgp_cdiv(a->N, 128) * p.batch;
    if (big_tiles >= resident_slots() / 2 && a->N > 64) return launch_main(p, tgp_hs(stream));
    return launch_small<false, false>(p, tgp_hs(stream));
}

// feature-space distance matrix for tgp_knn_feat (knn.hip)
int tgp_launch_dist_gemm(const float *x, int ld, const float *q, int B, int n, int d, float *D, hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return TGP_EINVAL;
    GemmParams p = {};
    p.A = x, p.W = x, p.C = D;
    p.lda = ld, p.ldw = ld, p.ldc = n, p.M = n, p.N = n, p.K = d;
    p.rows_per_obj = 1;
    p.batch = B;
    p.sA = p.sW = (int64_t)n * ld;
    p.sC = (int64_t)n * n;
    p.qrow = q, p.qcol = q, p.sq = n;
    return launch_small<true, true>(p, stream);
}

// ---------------------------------------------------------------------------------------------------
__global__ void colmax_decode_kernel(const uint32_t *__restrict__ keys, int ldk, int rows, int N,
                                     float *__restrict__ out, int ldo, float *__restrict__ out2)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * N) return;
    const int rr = (int)(t / N), c = (int)(t - (int64_t)rr * N);
    const float v = tgp_key_float(keys[(int64_t)rr * ldk + c]);
    out[(int64_t)rr * ldo + c] = v;
    if (out2) out2[(int64_t)rr * ldo + c] = v;
}

extern "C" int tgp_colmax_decode(const uint32_t *keys, int ldk, int rows, int N, float *out, int ldo, float *out2,
                                 tgp_stream_t stream)
{
    TGP_REQUIRE(keys && out && rows > 0 && N > 0 && ldk >= N && ldo >= N);
    hipLaunchKernelGGL(colmax_decode_kernel, dim3(tgp_cdiv((int64_t)rows * N, 256)), dim3(256), 0, tgp_hs(stream), keys,
                       ldk, rows, N, out, ldo, out2);
    return TGP_LAUNCH_RESULT();
}

// out[b,c] = max_i x[b,i,c]; block = (b, 64 columns) x 4 row-slices, combined through LDS
__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ x, int ld, int n, int C,
                                                     float *__restrict__ out)
{
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    float m = -INFINITY;
    if (c < C) {
        const float *xb = x + (int64_t)b * n * ld + c;
        for (int i = slice; i < n; i += 4) m = fmaxf(m, xb[(int64_t)i * ld]);
    }
    part[slice][threadIdx.x & 63] = m;
    __syncthreads();
    if (slice == 0 && c < C) {
        m = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]), fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
        out[(int64_t)b * C + c] = m;
    }
}

extern "C" int tgp_colmax(const float *x, int ld, int B, int n, int C, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(x && out && B > 0 && n > 0 && C > 0 && ld >= C);
    hipLaunchKernelGGL(colmax_kernel, dim3(tgp_cdiv(C, 64), B), dim3(256), 0, tgp_hs(stream), x, ld, n, C, out);
    return TGP_LAUNCH_RESULT();
}

__global__ void sigmoid_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) y[t] = 1.0f / (1.0f + expf(-x[t]));
}

extern "C" int tgp_sigmoid(const float *x, float *y, int64_t count, tgp_stream_t stream)
{
    TGP_REQUIRE(x && y && count > 0);
    hipLaunchKernelGGL(sigmoid_kernel, dim3(tgp_cdiv(count, 256)), dim3(256), 0, tgp_hs(stream), x, y, count);
    return TGP_LAUNCH_RESULT();
}

__global__ void head_post_kernel(const float *__restrict__ green, const float *__restrict__ red,
                                 const float *__restrict__ ts, const float *__restrict__ mean, int B,
                                 float *__restrict__ pg, float *__restrict__ pr, float *__restrict__ fg,
                                 float *__restrict__ fr, float *__restrict__ pT, float *__restrict__ ps)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *g = green + b * 4, *rd = red + b * 4;
    // torch.norm(v, dim=1): sqrt of the sum of squares; PoseNet9D.py:57-58 divides by (norm + 1e-6)
    float ng = sqrtf((g[1] * g[1] + g[2] * g[2]) + g[3] * g[3]) + 1e-6f;
    float nr = sqrtf((rd[1] * rd[1] + rd[2] * rd[2]) + rd[3] * rd[3]) + 1e-6f;
    for (int c = 0; c < 3; ++c) {
        pg[b * 3 + c] = g[1 + c] / ng;
        pr[b * 3 + c] = rd[1 + c] / nr;
        pT[b * 3 + c] = ts[b * 6 + c] + mean[b * 3 + c];
        ps[b * 3 + c] = ts[b * 6 + 3 + c];
    }
    fg[b] = 1.0f / (1.0f + expf(-g[0]));
    fr[b] = 1.0f / (1.0f + expf(-rd[0]));
}

extern "C" int tgp_head_post(const float *green, const float *red, const float *ts, const float *mean, int B,
                             float *p_green, float *p_red, float *f_green, float *f_red, float *pred_T, float *pred_s,
                             tgp_stream_t stream)
{
    TGP_REQUIRE(green && red && ts && mean && p_green && p_red && f_green && f_red && pred_T && pred_s && B > 0);
    hipLaunchKernelGGL(head_post_kernel, dim3(tgp_cdiv(B, 64)), dim3(64), 0, tgp_hs(stream), green, red, ts, mean, B,
                       p_green, p_red, f_green, f_red, pred_T, pred_s);
    return TGP_LAUNCH_RESULT();
}

__global__ void add_mean_kernel(float *__restrict__ recon, const float *__restrict__ mean, int n, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t pt = t / 3;
    const int c = (int)(t - pt * 3);
    recon[t] = recon[t] + mean[(pt / n) * 3 + c];
}

extern "C" int tgp_add_mean(float *recon, const float *mean, int B, int n, tgp_stream_t stream)
{
    TGP_REQUIRE(recon && mean && B > 0 && n > 0);
    const int64_t total = (int64_t)B * n * 3;
    hipLaunchKernelGGL(add_mean_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), recon, mean, n, total);
    return TGP_LAUNCH_RESULT();
}
