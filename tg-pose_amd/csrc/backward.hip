// Backward-pass kernels of the dense per-point layers (Conv1d k=1 / Linear + BatchNorm(train) + activation + max over
// points), gfx950.  What torch autograd runs for nn.Conv1d / nn.BatchNorm1d / F.relu / torch.max in the reference's
// training step (trainer/RL_TDA.py:205-224 calls loss.backward() on the graph PoseNet9D.forward built).
//
//   y = act(BN(x)),  x = a W^T + b        dW = dx^T a   (tgp_gemm_tn_f32: reduction over the rows)
//                                          da = dx W     (tgp_gemm_f32 on the transposed weight)
//                                          db = sum_rows dx (tgp_colsum)
//   BN(train):  dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)),  dz = dy * act'(z)
//               dgamma = sum dz * xhat,  dbeta = sum dz
// All reductions over rows are chunked and summed in chunk order (deterministic).
#include "tgp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// C[n][k] = sum_m A[m][n] * B[m][k]   (A: rows x N, B: rows x K, both row-major, reduction over the rows)
// v_mfma_f32_32x32x2_f32 takes its operands one element per lane: lane (i = lane % 32, kk = lane / 32) supplies
// A^T[i][kk] = A[m + kk][n0 + i] -- 32 consecutive floats of one row for 32 consecutive lanes, so both operands are read
// straight from global memory, coalesced, with no LDS transpose.  A workgroup (4 waves, 2 x 2) owns a 128 x 128 tile of C
// and one slice of the rows; slices are written to a workspace and summed in order by gemm_tn_reduce_kernel.
#define TN_TILE 128
#define TN_UNROLL 8

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                      int64_t rows, int N, int K, int64_t rows_per_slice,
                                                      float *__restrict__ C, int ldc, int64_t slice_stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int nb = blockIdx.y * TN_TILE + (wave >> 1) * 64, kb = blockIdx.x * TN_TILE + (wave & 1) * 64;
    const int64_t m0 = (int64_t)blockIdx.z * rows_per_slice;
    const int64_t m1 = m0 + rows_per_slice < rows ? m0 + rows_per_slice : rows;
    // clamped column offsets (always valid addresses); out-of-range columns are zeroed by the select below
    int acol[2], bcol[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        aok[t] = nb + 32 * t + i < N, bok[t] = kb + 32 * t + i < K;
        acol[t] = aok[t] ? nb + 32 * t + i : 0, bcol[t] = bok[t] ? kb + 32 * t + i : 0;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

    for (int64_t m = m0; m < m1; m += 2 * TN_UNROLL) {
        float a[TN_UNROLL][2], b[TN_UNROLL][2];
#pragma unroll
        for (int u = 0; u < TN_UNROLL; ++u) {
            const int64_t row = m + 2 * u + kk;
            const bool rok = row < m1;
            const int64_t rc = rok ? row : m1 - 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float av = A[rc * lda + acol[t]], bv = B[rc * ldb + bcol[t]];
                a[u][t] = (rok && aok[t]) ? av : 0.f;
                b[u][t] = (rok && bok[t]) ? bv : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < TN_UNROLL; ++u)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][x], b[u][y], acc[x][y], 0, 0, 0);
    }
    float *out = C + (int64_t)blockIdx.z * slice_stride;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = kb + 32 * y + i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = nb + 32 * x + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (row < N && col < K) out[(int64_t)row * ldc + col] = acc[x][y][e];
            }
        }
}

// The same product with the row slabs staged through LDS (aligned operands: lda, ldb, N, K multiples of 4, 16-byte aligned
// bases): 16 rows x 128 columns of A and of B per step, loaded once per workgroup with 16-byte lane loads (the direct form
// above re-reads every operand element per wave and is bound by the L1), double buffered.  The MFMA operand reads are
// ds_read_b32 of 32 consecutive floats per half wave: conflict free.
#define TN_ROWS 16

__global__ __launch_bounds__(256) void gemm_tn_lds_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                          int64_t rows, int N, int K, int64_t rows_per_slice,
                                                          float *__restrict__ C, int ldc, int64_t slice_stride)
{
    __shared__ __attribute__((aligned(16))) float sa[2][TN_ROWS][TN_TILE];
    __shared__ __attribute__((aligned(16))) float sb[2][TN_ROWS][TN_TILE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int n0 = blockIdx.y * TN_TILE, k0 = blockIdx.x * TN_TILE;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int64_t m0 = (int64_t)blockIdx.z * rows_per_slice;
    const int64_t m1 = m0 + rows_per_slice < rows ? m0 + rows_per_slice : rows;
    // staging: thread -> (row r and r + 8, float4 column c4)
    const int r = tid >> 5, c4 = (tid & 31) * 4;
    const bool a_ok = n0 + c4 < N, b_ok = k0 + c4 < K;      // N, K multiples of 4: a quad is wholly inside or outside
    const float *ap = A + (a_ok ? n0 + c4 : 0), *bp = B + (b_ok ? k0 + c4 : 0);
    float4 ra[2], rb[2];
    auto load = [&](int64_t m) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t row = m + r + 8 * t;
            const int64_t rc = row < m1 ? row : m1 - 1;
            const float4 av = *reinterpret_cast<const float4 *>(ap + rc * lda);
            const float4 bv = *reinterpret_cast<const float4 *>(bp + rc * ldb);
            const float fa = (row < m1 && a_ok) ? 1.f : 0.f, fb = (row < m1 && b_ok) ? 1.f : 0.f;
            // multiply by a 0/1 mask instead of selecting: keeps the loads free of branches (and of early waits)
            ra[t] = make_float4(av.x * fa, av.y * fa, av.z * fa, av.w * fa);
            rb[t] = make_float4(bv.x * fb, bv.y * fb, bv.z * fb, bv.w * fb);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            *reinterpret_cast<float4 *>(&sa[buf][r + 8 * t][c4]) = ra[t];
            *reinterpret_cast<float4 *>(&sb[buf][r + 8 * t][c4]) = rb[t];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

    load(m0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m = m0; m < m1; m += TN_ROWS) {
        const bool more = m + TN_ROWS < m1;
        if (more) load(m + TN_ROWS);
#pragma unroll
        for (int u = 0; u < TN_ROWS / 2; ++u) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = sa[buf][2 * u + kk][wn + 32 * t + i], b[t] = sb[buf][2 * u + kk][wk + 32 * t + i];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float *out = C + (int64_t)blockIdx.z * slice_stride;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = k0 + wk + 32 * y + i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = n0 + wn + 32 * x + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (row < N && col < K) out[(int64_t)row * ldc + col] = acc[x][y][e];
            }
        }
}

// One operand a few columns wide (the decoder's 128 -> 3 head: dW is 3 x 128; the surface layer's STE on xyz: 128 x 4): the MFMA tile
// kernels spend a 128 x 128 tile on it (159 / 66 us per launch at 32896 rows).  Here a thread owns one column of the WIDE operand and
// keeps the <= 8 products with the narrow one in registers; the narrow operand's row is the same address for the whole workgroup.
// part[slice][w][c] = sum over the slice's rows of narrow[r][w] * wide[r][c]
#define TNS_MAXW 8
// 64 columns of the wide operand x 4 row lanes per workgroup; a row lane takes every fourth row of the slice, the four partial sums
// are combined through LDS in lane order (deterministic)
__global__ __launch_bounds__(256) void gemm_tn_skinny_kernel(const float *__restrict__ nar, int ldn, int Wn, const float *__restrict__ wide,
                                                             int ldw, int Wd, int64_t rows, int64_t rows_per_slice, float *__restrict__ part)
{
    __shared__ float red[4][TNS_MAXW][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t m1 = m0 + rows_per_slice < rows ? m0 + rows_per_slice : rows;
    float acc[TNS_MAXW];
#pragma unroll
    for (int w = 0; w < TNS_MAXW; ++w) acc[w] = 0.f;
    const int cc = c < Wd ? c : 0;
#pragma unroll 8
    for (int64_t m = m0 + rl; m < m1; m += 4) {
        const float x = wide[m * ldw + cc];
#pragma unroll
        for (int w = 0; w < TNS_MAXW; ++w)
            if (w < Wn) acc[w] = fmaf(nar[m * ldn + w], x, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < TNS_MAXW; ++w) red[rl][w][cl] = acc[w];
    __syncthreads();
    if (rl == 0 && c < Wd)
#pragma unroll
        for (int w = 0; w < TNS_MAXW; ++w)
            if (w < Wn) part[((int64_t)blockIdx.y * Wn + w) * Wd + c] = ((red[0][w][cl] + red[1][w][cl]) + red[2][w][cl]) + red[3][w][cl];
}

// out[n][k] (+)= sum over slices, in slice order (transposed: the slices hold [k][n], the skinny kernel's layout when B is the narrow one)
__global__ void gemm_tn_reduce_kernel(const float *__restrict__ part, int slices, int64_t slice_stride, int N, int K,
                                      float *__restrict__ out, int ldo, int accumulate, int transposed = 0)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * K) return;
    const int n = (int)(t / K), k = (int)(t - (int64_t)n * K);
    float s = 0.f;
    if (transposed) {
        for (int z = 0; z < slices; ++z) s += part[(int64_t)z * slice_stride + (int64_t)k * N + n];
    } else
    for (int z = 0; z < slices; ++z) s += part[(int64_t)z * slice_stride + t];
    float *o = out + (int64_t)n * ldo + k;
    *o = accumulate ? *o + s : s;
}

static int tn_slices(int64_t rows, int N, int K)
{
    const int64_t tiles = (int64_t)tgp_cdiv(N, TN_TILE) * tgp_cdiv(K, TN_TILE);
    int64_t s = tgp_cdiv((int64_t)1024, tiles);           // ~4 workgroups per CU in flight
    const int64_t max_s = tgp_cdiv(rows, (int64_t)256);   // at least 256 rows per slice
    if (s > max_s) s = max_s;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t tgp_gemm_tn_workspace_floats(int64_t rows, int N, int K)
{
    if (rows <= 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)tn_slices(rows, N, K) * N * K;
}

extern "C" int tgp_gemm_tn_f32(const float *A, int lda, const float *B, int ldb, int64_t rows, int N, int K, float *C, int ldc,
                               int accumulate, float *workspace, tgp_stream_t stream)
{
    TGP_REQUIRE(A && B && C && workspace && rows > 0 && N > 0 && K > 0 && lda >= N && ldb >= K && ldc >= K);
    const int slices = tn_slices(rows, N, K);
    int64_t per = tgp_cdiv(rows, (int64_t)slices);
    per = (per + 2 * TN_UNROLL - 1) / (2 * TN_UNROLL) * (2 * TN_UNROLL);
    const int used = (int)tgp_cdiv(rows, per);
    const int64_t stride = (int64_t)N * K;
    if ((N <= TNS_MAXW || K <= TNS_MAXW) && rows >= 1024) {
        const bool a_narrow = N <= K;                          // the narrow operand: A (N columns) or B (K columns)
        const int Wn = a_narrow ? N : K, Wd = a_narrow ? K : N;
        hipLaunchKernelGGL(gemm_tn_skinny_kernel, dim3(tgp_cdiv(Wd, 64), used), dim3(256), 0, tgp_hs(stream), a_narrow ? A : B,
                           a_narrow ? lda : ldb, Wn, a_narrow ? B : A, a_narrow ? ldb : lda, Wd, rows, per, workspace);
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(tgp_cdiv(stride, (int64_t)256)), dim3(256), 0, tgp_hs(stream), workspace, used,
                           stride, N, K, C, ldc, accumulate, a_narrow ? 0 : 1);
        return TGP_LAUNCH_RESULT();
    }
    const bool aligned = !((lda | ldb | N | K) & 3) && !((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15);
    const dim3 grid(tgp_cdiv(K, TN_TILE), tgp_cdiv(N, TN_TILE), used);
    if (aligned)
        hipLaunchKernelGGL(gemm_tn_lds_kernel, grid, dim3(256), 0, tgp_hs(stream), A, lda, B, ldb, rows, N, K, per, workspace, K, stride);
    else
        hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, tgp_hs(stream), A, lda, B, ldb, rows, N, K, per, workspace, K, stride);
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(tgp_cdiv(stride, (int64_t)256)), dim3(256), 0, tgp_hs(stream), workspace, used,
                       stride, N, K, C, ldc, accumulate);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------
// Column sums (bias gradients) and the two BatchNorm-backward sums, chunked like bn.hip.
#define BW_CHUNK 1024

__device__ __forceinline__ float act_grad(float z, int act, float slope) { return (act == 1 && !(z > 0.f)) ? slope : 1.f; }

// MODE 0: partial[chunk][c] = sum dy;  MODE 1: s1 = sum dz, s2 = sum dz * xhat (dz = dy * act'(z), z = xhat * gamma + beta)
template <int MODE>
__global__ __launch_bounds__(256) void bw_partial_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ld,
                                                         int64_t rows, int C, const float *__restrict__ mean,
                                                         const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int act, float slope,
                                                         const float *__restrict__ slope_vec, float *__restrict__ p1,
                                                         float *__restrict__ p2)
{
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * BW_CHUNK;
    const int64_t r1 = r0 + BW_CHUNK < rows ? r0 + BW_CHUNK : rows;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        float mu = 0.f, inv = 0.f, g = 0.f, b = 0.f, sl = 0.f;
        if (MODE == 1) mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps), g = gamma[c], b = beta[c], sl = slope_vec ? slope_vec[c] : slope;
#pragma unroll 8
    for (int64_t r = r0 + slice; r < r1; r += 4) {
            float d = dy[r * lddy + c];
            if (MODE == 1) {
                const float xh = (x[r * ld + c] - mu) * inv;
                d *= act_grad(xh * g + b, act, sl);
                s2 += d * xh;
            }
            s1 += d;
        }
    }
    red[0][slice][threadIdx.x & 63] = s1;
    red[1][slice][threadIdx.x & 63] = s2;
    __syncthreads();
    if (slice == 0 && c < C) {
        const int l = threadIdx.x;
        p1[(int64_t)blockIdx.y * C + c] = ((red[0][0][l] + red[0][1][l]) + red[0][2][l]) + red[0][3][l];
        if (MODE == 1) p2[(int64_t)blockIdx.y * C + c] = ((red[1][0][l] + red[1][1][l]) + red[1][2][l]) + red[1][3][l];
    }
}

// (round 3) the same sums with a column quad per thread: 16-byte loads, 256 columns per workgroup, BWV_CHUNK = 256 rows per
// workgroup.  (First attempt: the scalar kernels' 1024-row chunks -- a quarter of the workgroups, 8 waves per CU -- and every
// 16-byte kernel was SLOWER than its 4-byte twin, bn_bwd_apply 405 -> 949 us per step: these passes live on latency hiding.)
// Deterministic: fixed slice and chunk order (not the scalar kernels' order: the chunks are shorter).
#define BWV_CHUNK 256
template <int MODE>
__global__ __launch_bounds__(256) void bw_partial_v4_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ld,
                                                            int64_t rows, int C, const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int act, float slope,
                                                            const float *__restrict__ slope_vec, float *__restrict__ p1,
                                                            float *__restrict__ p2)
{
    __shared__ float4 red[2][4][64];
    const int quad = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + quad * 4;
    const int64_t r0 = (int64_t)blockIdx.y * BWV_CHUNK;
    const int64_t r1 = r0 + BWV_CHUNK < rows ? r0 + BWV_CHUNK : rows;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, inv[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f},
              sl[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mu[q] = mean[c + q], inv[q] = 1.0f / sqrtf(var[c + q] + eps), g[q] = gamma[c + q], b[q] = beta[c + q],
                sl[q] = slope_vec ? slope_vec[c + q] : slope;
        }
#pragma unroll 4
        for (int64_t r = r0 + slice; r < r1; r += 4) {
            const float4 dv = *reinterpret_cast<const float4 *>(dy + r * lddy + c);
            float d[4] = {dv.x, dv.y, dv.z, dv.w};
            if (MODE == 1) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + r * ld + c);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xh = (xs[q] - mu[q]) * inv[q];
                    d[q] *= act_grad(xh * g[q] + b[q], act, sl[q]);
                    s2[q] += d[q] * xh;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s1[q] += d[q];
        }
    }
    red[0][slice][quad] = make_float4(s1[0], s1[1], s1[2], s1[3]);
    red[1][slice][quad] = make_float4(s2[0], s2[1], s2[2], s2[3]);
    __syncthreads();
    if (slice == 0 && c < C) {
        auto comb = [&](int w) {
            const float4 a0 = red[w][0][quad], a1 = red[w][1][quad], a2 = red[w][2][quad], a3 = red[w][3][quad];
            return make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z,
                               ((a0.w + a1.w) + a2.w) + a3.w);
        };
        *reinterpret_cast<float4 *>(p1 + (int64_t)blockIdx.y * C + c) = comb(0);
        if (MODE == 1) *reinterpret_cast<float4 *>(p2 + (int64_t)blockIdx.y * C + c) = comb(1);
    }
}

// both BatchNorm sums finished by one launch (was two): 64 columns x 4 chunk groups per workgroup, groups added in order
__global__ __launch_bounds__(256) void bw_finish2_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int chunks, int C,
                                                         float *__restrict__ o1, float *__restrict__ o2, int accumulate)
{
    __shared__ float sa[4][64], sb[4][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    const int per = (chunks + 3) / 4;
    float a = 0.f, b = 0.f;
    if (c < C) {
        const int k1 = (g + 1) * per < chunks ? (g + 1) * per : chunks;
#pragma unroll 4
        for (int k = g * per; k < k1; ++k) {
            a += p1[(int64_t)k * C + c];
            if (p2) b += p2[(int64_t)k * C + c];
        }
    }
    sa[g][l] = a, sb[g][l] = b;
    __syncthreads();
    if (g == 0 && c < C) {
        a = ((sa[0][l] + sa[1][l]) + sa[2][l]) + sa[3][l];
        o1[c] = accumulate ? o1[c] + a : a;
        if (p2) o2[c] = ((sb[0][l] + sb[1][l]) + sb[2][l]) + sb[3][l];
    }
}

static bool bw_vec_ok(const void *a, int lda, const void *b, int ldb, int C)
{
#ifdef TGP_BN_SCALAR        // measurement builds only
    return false;
#endif
    return (C & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

__global__ void bw_finish_kernel(const float *__restrict__ partial, int chunks, int C, float *__restrict__ out, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(int64_t)k * C + c];
    out[c] = accumulate ? out[c] + s : s;
}

extern "C" int64_t tgp_bw_workspace_floats(int64_t rows, int C) { return rows > 0 && C > 0 ? 2 * (int64_t)tgp_cdiv(rows, BWV_CHUNK) * C : 0; }

extern "C" int tgp_colsum(const float *dy, int lddy, int64_t rows, int C, float *out, int accumulate, float *workspace,
                          tgp_stream_t stream)
{
    TGP_REQUIRE(dy && out && workspace && rows > 0 && C > 0 && lddy >= C);
    const int chunks = tgp_cdiv(rows, BW_CHUNK);
    if (bw_vec_ok(dy, lddy, workspace, 0, C)) {
        const int vchunks = tgp_cdiv(rows, BWV_CHUNK);
        hipLaunchKernelGGL(bw_partial_v4_kernel<0>, dim3(tgp_cdiv(C, 256), vchunks), dim3(256), 0, tgp_hs(stream), dy, lddy,
                           (const float *)nullptr, 0, rows, C, (const float *)nullptr, (const float *)nullptr, 0.f,
                           (const float *)nullptr, (const float *)nullptr, 0, 0.f, (const float *)nullptr, workspace, (float *)nullptr);
        hipLaunchKernelGGL(bw_finish2_kernel, dim3(tgp_cdiv(C, 64)), dim3(256), 0, tgp_hs(stream), workspace, (const float *)nullptr, vchunks, C,
                           out, (float *)nullptr, accumulate);
        return TGP_LAUNCH_RESULT();
    }
    hipLaunchKernelGGL(bw_partial_kernel<0>, dim3(tgp_cdiv(C, 64), chunks), dim3(256), 0, tgp_hs(stream), dy, lddy,
                       (const float *)nullptr, 0, rows, C, (const float *)nullptr, (const float *)nullptr, 0.f,
                       (const float *)nullptr, (const float *)nullptr, 0, 0.f, (const float *)nullptr, workspace, (float *)nullptr);
    hipLaunchKernelGGL(bw_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), workspace, chunks, C, out, accumulate);
    return TGP_LAUNCH_RESULT();
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ld,
                                                           int64_t rows, int C, const float *__restrict__ mean,
                                                           const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float slope,
                                                           const float *__restrict__ slope_vec, const float *__restrict__ s1,
                                                           const float *__restrict__ s2, float *__restrict__ dx, int lddx)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    const float mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps), g = gamma[c], b = beta[c];
    const float sl = slope_vec ? slope_vec[c] : slope;
    const float inv_n = (float)(1.0 / (double)rows);
    const float m1 = s1[c] * inv_n, m2 = s2[c] * inv_n, scale = g * inv;
    const int64_t r0 = (int64_t)blockIdx.y * BW_CHUNK;
    const int64_t r1 = r0 + BW_CHUNK < rows ? r0 + BW_CHUNK : rows;
#pragma unroll 8
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const float xh = (x[r * ld + c] - mu) * inv;
        const float dz = dy[r * lddy + c] * act_grad(xh * g + b, act, sl);
        dx[r * lddx + c] = scale * ((dz - m1) - xh * m2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ld,
                                                              int64_t rows, int C, const float *__restrict__ mean,
                                                              const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int act, float slope,
                                                              const float *__restrict__ slope_vec, const float *__restrict__ s1,
                                                              const float *__restrict__ s2, float *__restrict__ dx, int lddx,
                                                              uint32_t *__restrict__ amax_bits)
{
    const int c0 = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int slice = threadIdx.x >> 6;
    const bool live = c0 < C;                                  // (no early return: the wave reduces max |dx| together at the end)
    const int c = live ? c0 : 0;
    float amax = 0.f;          // max |dx| of this thread (as bits a NaN sorts above every number: the scale kernel then answers 1)
    float poison = 0.f;
    const float inv_n = (float)(1.0 / (double)rows);
    float mu[4], inv[4], g[4], b[4], sl[4], m1[4], m2[4], scale[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[c + q], inv[q] = 1.0f / sqrtf(var[c + q] + eps), g[q] = gamma[c + q], b[q] = beta[c + q];
        sl[q] = slope_vec ? slope_vec[c + q] : slope;
        m1[q] = s1[c + q] * inv_n, m2[q] = s2[c + q] * inv_n, scale[q] = g[q] * inv[q];
    }
    const int64_t r0 = (int64_t)blockIdx.y * BWV_CHUNK;
    const int64_t r1 = live ? (r0 + BWV_CHUNK < rows ? r0 + BWV_CHUNK : rows) : r0;
#pragma unroll 4
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + r * ld + c);
        const float4 dv = *reinterpret_cast<const float4 *>(dy + r * lddy + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xs[q] - mu[q]) * inv[q];
            const float dz = ds[q] * act_grad(xh * g[q] + b[q], act, sl[q]);
            o[q] = scale[q] * ((dz - m1[q]) - xh * m2[q]);
            amax = fmaxf(amax, fabsf(o[q]));
            poison = fmaf(0.f, o[q], poison);                  // NaN once a NaN or an infinity went by (fmaxf drops NaNs); no branch
        }
        *reinterpret_cast<float4 *>(dx + r * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (amax_bits) {
        // max |dx| for the next layer's fp16 scale (tgp_absmax_scale_from_bits): the consumer of dx is a linear layer's backward,
        // which otherwise reads all of dx once more just for this number.  A maximum is order-free: bit-repeatable.
        uint32_t m = poison != poison ? 0x7fc00000u : __float_as_uint(amax);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o2 = (uint32_t)__shfl_xor((int)m, off, 64);
            m = o2 > m ? o2 : m;
        }
        __shared__ uint32_t s_m[4];
        if ((threadIdx.x & 63) == 0) s_m[slice] = m;
        __syncthreads();
        // one word per workgroup, no atomics (a thousand device-scope atomicMax on one address cost this pass 60 % more time)
        if (threadIdx.x == 0) {
            const uint32_t a = s_m[0] > s_m[1] ? s_m[0] : s_m[1], b2 = s_m[2] > s_m[3] ? s_m[2] : s_m[3];
            amax_bits[blockIdx.y * gridDim.x + blockIdx.x] = a > b2 ? a : b2;
        }
    }
}

// BatchNorm(train) + activation backward.  dgamma = sum dz * xhat, dbeta = sum dz are returned (they double as the
// batch sums of the dx formula).  dx may alias dy.
extern "C" int tgp_bn_bwd(const float *dy, int lddy, const float *x, int ld, int64_t rows, int C, const float *mean,
                          const float *var, float eps, const float *gamma, const float *beta, int act, float slope,
                          const float *slope_vec, float *dx, int lddx, float *dgamma, float *dbeta, float *workspace,
                          uint32_t *absmax_bits, tgp_stream_t stream)
{
    TGP_REQUIRE(dy && x && mean && var && gamma && beta && dx && dgamma && dbeta && workspace && rows > 0 && C > 0);
    TGP_REQUIRE(lddy >= C && ld >= C && lddx >= C && (act == 0 || act == 1));
    const int chunks = tgp_cdiv(rows, BW_CHUNK);
    float *p1 = workspace, *p2 = workspace + (int64_t)chunks * C;
    const dim3 grid(tgp_cdiv(C, 64), chunks), block(256);
    if (bw_vec_ok(dy, lddy, x, ld, C) && bw_vec_ok(dx, lddx, workspace, 0, C)) {
        const int vchunks = tgp_cdiv(rows, BWV_CHUNK);
        const dim3 grid4(tgp_cdiv(C, 256), vchunks);
        p2 = workspace + (int64_t)vchunks * C;
        hipLaunchKernelGGL(bw_partial_v4_kernel<1>, grid4, block, 0, tgp_hs(stream), dy, lddy, x, ld, rows, C, mean, var, eps, gamma, beta,
                           act, slope, slope_vec, p1, p2);
        hipLaunchKernelGGL(bw_finish2_kernel, dim3(tgp_cdiv(C, 64)), dim3(256), 0, tgp_hs(stream), p1, p2, vchunks, C, dbeta, dgamma, 0);
        hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, grid4, block, 0, tgp_hs(stream), dy, lddy, x, ld, rows, C, mean, var, eps, gamma, beta,
                           act, slope, slope_vec, dbeta, dgamma, dx, lddx, absmax_bits);
        return TGP_LAUNCH_RESULT();
    }
    if (absmax_bits) return TGP_EUNSUPPORTED;                  // (collected by the 16-byte form only; the caller asks for it there only)
    hipLaunchKernelGGL(bw_partial_kernel<1>, grid, block, 0, tgp_hs(stream), dy, lddy, x, ld, rows, C, mean, var, eps, gamma, beta,
                       act, slope, slope_vec, p1, p2);
    hipLaunchKernelGGL(bw_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), p1, chunks, C, dbeta, 0);
    hipLaunchKernelGGL(bw_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), p2, chunks, C, dgamma, 0);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, block, 0, tgp_hs(stream), dy, lddy, x, ld, rows, C, mean, var, eps, gamma, beta,
                       act, slope, slope_vec, dbeta, dgamma, dx, lddx);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------
// Layers whose output only feeds a max over each object's points (conv_5, the heads' conv2): the incoming gradient is
// (objects x C) and lands on one row per (object, channel) -- the argmax the forward recorded.
// sums: s1[c] = sum_b dz(b, c), s2[c] = sum_b dz(b, c) * xhat(argmax(b, c), c); one thread per channel, objects in order.
__global__ void bn_bwd_pooled_sums_kernel(const float *__restrict__ dpool, int ldp, const int *__restrict__ argrow, int lda,
                                          const float *__restrict__ x, int ld, int objects, int C, const float *__restrict__ mean,
                                          const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                          const float *__restrict__ beta, int act, float slope, const float *__restrict__ slope_vec,
                                          float *__restrict__ s1, float *__restrict__ s2)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps), g = gamma[c], b = beta[c];
    const float sl = slope_vec ? slope_vec[c] : slope;
    float a1 = 0.f, a2 = 0.f;
    for (int o = 0; o < objects; ++o) {
        const int64_t r = argrow[(int64_t)o * lda + c];
        const float xh = (x[r * ld + c] - mu) * inv;
        const float dz = dpool[(int64_t)o * ldp + c] * act_grad(xh * g + b, act, sl);
        a1 += dz, a2 += dz * xh;
    }
    s1[c] = a1, s2[c] = a2;
}

__global__ __launch_bounds__(256) void bn_bwd_pooled_apply_kernel(const float *__restrict__ dpool, int ldp,
                                                                  const int *__restrict__ argrow, int lda,
                                                                  const float *__restrict__ x, int ld, int64_t rows, int rows_per_obj,
                                                                  int C, const float *__restrict__ mean, const float *__restrict__ var,
                                                                  float eps, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, int act, float slope,
                                                                  const float *__restrict__ slope_vec, const float *__restrict__ s1,
                                                                  const float *__restrict__ s2, float *__restrict__ dx, int lddx)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    const float mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps), g = gamma[c], b = beta[c];
    const float sl = slope_vec ? slope_vec[c] : slope;
    const float inv_n = (float)(1.0 / (double)rows);
    const float m1 = s1[c] * inv_n, m2 = s2[c] * inv_n, scale = g * inv;
    const int64_t r0 = (int64_t)blockIdx.y * BW_CHUNK;
    const int64_t r1 = r0 + BW_CHUNK < rows ? r0 + BW_CHUNK : rows;
#pragma unroll 8
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const float xh = (x[r * ld + c] - mu) * inv;
        const int64_t o = r / rows_per_obj;
        float dz = 0.f;
        if (argrow[o * lda + c] == r) dz = dpool[o * ldp + c] * act_grad(xh * g + b, act, sl);
        dx[r * lddx + c] = scale * ((dz - m1) - xh * m2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_pooled_apply_v4_kernel(const float *__restrict__ dpool, int ldp,
                                                                     const int *__restrict__ argrow, int lda,
                                                                     const float *__restrict__ x, int ld, int64_t rows, int rows_per_obj,
                                                                     int C, const float *__restrict__ mean, const float *__restrict__ var,
                                                                     float eps, const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, int act, float slope,
                                                                     const float *__restrict__ slope_vec, const float *__restrict__ s1,
                                                                     const float *__restrict__ s2, float *__restrict__ dx, int lddx)
{
    const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    const float inv_n = (float)(1.0 / (double)rows);
    float mu[4], inv[4], g[4], b[4], sl[4], m1[4], m2[4], scale[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[c + q], inv[q] = 1.0f / sqrtf(var[c + q] + eps), g[q] = gamma[c + q], b[q] = beta[c + q];
        sl[q] = slope_vec ? slope_vec[c + q] : slope;
        m1[q] = s1[c + q] * inv_n, m2[q] = s2[c + q] * inv_n, scale[q] = g[q] * inv[q];
    }
    const int64_t r0 = (int64_t)blockIdx.y * BWV_CHUNK;
    const int64_t r1 = r0 + BWV_CHUNK < rows ? r0 + BWV_CHUNK : rows;
#pragma unroll 4
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + r * ld + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const int64_t o = r / rows_per_obj;
        const int4 ar = *reinterpret_cast<const int4 *>(argrow + o * lda + c);
        const int as[4] = {ar.x, ar.y, ar.z, ar.w};
        float ov[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xs[q] - mu[q]) * inv[q];
            float dz = 0.f;
            if (as[q] == r) dz = dpool[o * ldp + c + q] * act_grad(xh * g[q] + b[q], act, sl[q]);
            ov[q] = scale[q] * ((dz - m1[q]) - xh * m2[q]);
        }
        *reinterpret_cast<float4 *>(dx + r * lddx + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
}

extern "C" int tgp_bn_bwd_pooled(const float *dpool, int ldp, const int *argrow, int lda, const float *x, int ld, int objects,
                                 int rows_per_obj, int C, const float *mean, const float *var, float eps, const float *gamma,
                                 const float *beta, int act, float slope, const float *slope_vec, float *dx, int lddx,
                                 float *dgamma, float *dbeta, tgp_stream_t stream)
{
    TGP_REQUIRE(dpool && argrow && x && mean && var && gamma && beta && dx && dgamma && dbeta);
    TGP_REQUIRE(objects > 0 && rows_per_obj > 0 && C > 0 && ldp >= C && lda >= C && ld >= C && lddx >= C && (act == 0 || act == 1));
    const int64_t rows = (int64_t)objects * rows_per_obj;
    hipLaunchKernelGGL(bn_bwd_pooled_sums_kernel, dim3(tgp_cdiv(C, 64)), dim3(64), 0, tgp_hs(stream), dpool, ldp, argrow, lda, x, ld,
                       objects, C, mean, var, eps, gamma, beta, act, slope, slope_vec, dbeta, dgamma);
    if (bw_vec_ok(x, ld, dx, lddx, C) && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(argrow) & 15) == 0) {
        hipLaunchKernelGGL(bn_bwd_pooled_apply_v4_kernel, dim3(tgp_cdiv(C, 256), tgp_cdiv(rows, (int64_t)BWV_CHUNK)), dim3(256), 0,
                           tgp_hs(stream), dpool, ldp, argrow, lda, x, ld, rows, rows_per_obj, C, mean, var, eps, gamma, beta, act,
                           slope, slope_vec, dbeta, dgamma, dx, lddx);
        return TGP_LAUNCH_RESULT();
    }
    hipLaunchKernelGGL(bn_bwd_pooled_apply_kernel, dim3(tgp_cdiv(C, 64), tgp_cdiv(rows, (int64_t)BW_CHUNK)), dim3(256), 0,
                       tgp_hs(stream), dpool, ldp, argrow, lda, x, ld, rows, rows_per_obj, C, mean, var, eps, gamma, beta, act,
                       slope, slope_vec, dbeta, dgamma, dx, lddx);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------
// max over each object's points with the winning row (first row on ties, as torch.max) of y = act(BN(x)) computed on the
// fly from the raw layer output (mean == NULL: y = x).  CW channels x 1024 / CW row slices per workgroup: slice s scans the
// contiguous rows [s n / S, (s + 1) n / S) of the object, the (value, row) candidates are merged in slice order with a strict
// '>' so the first row wins ties.  (Round 3: was 64 channels x 4 slices in 256 threads -- 128 workgroups for a 256-channel
// layer at B = 32, half a wave per SIMD walking 257 rows each: 87 us for 34 MB.)
template <int CW>
__global__ __launch_bounds__(1024) void colmax_arg_kernel(const float *__restrict__ x, int ld, int n, int C, const float *__restrict__ mean,
                                  const float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, int act, float slope, const float *__restrict__ slope_vec,
                                  float *__restrict__ out, int ldo, int *__restrict__ argrow, int lda)
{
    constexpr int S = 1024 / CW;
    __shared__ float s_best[S][CW];
    __shared__ int s_arg[S][CW];
    const int lane = threadIdx.x % CW;
    const int c = blockIdx.x * CW + lane;
    const int slice = threadIdx.x / CW;
    const int o = blockIdx.y;
    float mu = 0.f, a = 1.f, b = 0.f, sl = 0.f;
    if (c < C && mean) mu = mean[c], a = gamma[c] / sqrtf(var[c] + eps), b = beta[c], sl = slope_vec ? slope_vec[c] : slope;
    const int64_t r0 = (int64_t)o * n;
    const int64_t ra = r0 + (int64_t)n * slice / S, rb = r0 + (int64_t)n * (slice + 1) / S;
    float best = 0.f;
    int64_t arg = -1;
    if (c < C) {
#pragma unroll 8
        for (int64_t r = ra; r < rb; ++r) {
            float v = x[r * ld + c];
            if (mean) {
                v = (v - mu) * a + b;      // same expression as bn_apply_kernel: the pooled value equals the stored activation
                if (act == 1) v = v > 0.f ? v : v * sl;
            }
            if (arg < 0 || v > best || (v != v && best == best)) best = v, arg = r;   // NaN propagates like torch.max
        }
    }
    s_best[slice][lane] = best;
    s_arg[slice][lane] = (int)arg;
    __syncthreads();
    if (slice == 0 && c < C) {
        for (int s = 1; s < S; ++s) {
            const float v = s_best[s][lane];
            const int r = s_arg[s][lane];
            if (r >= 0 && (arg < 0 || v > best || (v != v && best == best))) best = v, arg = r;
        }
        out[(int64_t)o * ldo + c] = best;
        argrow[(int64_t)o * lda + c] = (int)arg;
    }
}

// narrow layers get 32-channel workgroups (twice the workgroups, 32 row slices each)
static inline bool per_object_narrow(int C, int objects) { return (int64_t)tgp_cdiv(C, 64) * objects < 512; }

extern "C" int tgp_colmax_arg(const float *x, int ld, int objects, int n, int C, const float *mean, const float *var, float eps,
                              const float *gamma, const float *beta, int act, float slope, const float *slope_vec, float *out,
                              int ldo, int *argrow, int lda, tgp_stream_t stream)
{
    TGP_REQUIRE(x && out && argrow && objects > 0 && n > 0 && C > 0 && ld >= C && ldo >= C && lda >= C);
    TGP_REQUIRE(!mean || (var && gamma && beta));
    TGP_REQUIRE((int64_t)objects * n < 0x7fffffff);
    if (per_object_narrow(C, objects))
        hipLaunchKernelGGL(colmax_arg_kernel<32>, dim3(tgp_cdiv(C, 32), objects), dim3(1024), 0, tgp_hs(stream), x, ld, n, C, mean, var,
                           eps, gamma, beta, act, slope, slope_vec, out, ldo, argrow, lda);
    else
        hipLaunchKernelGGL(colmax_arg_kernel<64>, dim3(tgp_cdiv(C, 64), objects), dim3(1024), 0, tgp_hs(stream), x, ld, n, C, mean, var,
                           eps, gamma, beta, act, slope, slope_vec, out, ldo, argrow, lda);
    return TGP_LAUNCH_RESULT();
}

// out[o][c] = sum over the object's n rows of dy: the gradient of a per-object bias that was broadcast over the object's points
// (ORL_forward's global half, gcn3d.py:108-112; the PH back-projection in front of the decoder, FaceRecon.py:165).  CW channels x
// 1024 / CW contiguous row slices per workgroup, the partial sums combined in slice order: deterministic.  (torch's own sum over the
// point dimension is a multi-block reduction with semaphores, which did not survive hipGraph replay reliably: the backward of the
// captured step keeps to this library's kernels.)
template <int CW>
__global__ __launch_bounds__(1024) void colsum_objects_kernel(const float *__restrict__ dy, int ld, int n, int C, float *__restrict__ out, int ldo)
{
    constexpr int S = 1024 / CW;
    __shared__ float part[S][CW];
    const int lane = threadIdx.x % CW;
    const int c = blockIdx.x * CW + lane;
    const int slice = threadIdx.x / CW;
    const int o = blockIdx.y;
    const int64_t r0 = (int64_t)o * n;
    const int64_t ra = r0 + (int64_t)n * slice / S, rb = r0 + (int64_t)n * (slice + 1) / S;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int64_t r = ra; r < rb; ++r) acc += dy[r * ld + c];
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && c < C) {
        for (int s = 1; s < S; ++s) acc += part[s][lane];
        out[(int64_t)o * ldo + c] = acc;
    }
}

extern "C" int tgp_colsum_objects(const float *dy, int ld, int objects, int n, int C, float *out, int ldo, tgp_stream_t stream)
{
    TGP_REQUIRE(dy && out && objects > 0 && n > 0 && C > 0 && ld >= C && ldo >= C);
    if (per_object_narrow(C, objects))
        hipLaunchKernelGGL(colsum_objects_kernel<32>, dim3(tgp_cdiv(C, 32), objects), dim3(1024), 0, tgp_hs(stream), dy, ld, n, C, out, ldo);
    else
        hipLaunchKernelGGL(colsum_objects_kernel<64>, dim3(tgp_cdiv(C, 64), objects), dim3(1024), 0, tgp_hs(stream), dy, ld, n, C, out, ldo);
    return TGP_LAUNCH_RESULT();
}

// backward of out[o][c] = max over the object's rows of x: dx[r][c] = dpool[o][c] where r is the recorded winning row, 0 elsewhere.
// Dense write (every element exactly once: no atomics, no zero-fill pass), coalesced across channels.
__global__ __launch_bounds__(256) void colmax_bwd_kernel(const float *__restrict__ dpool, int ldp, const int *__restrict__ argrow, int lda,
                                                         int64_t rows, int rows_per_obj, int C, float *__restrict__ dx, int lddx)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * BW_CHUNK;
    const int64_t r1 = r0 + BW_CHUNK < rows ? r0 + BW_CHUNK : rows;
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const int64_t o = r / rows_per_obj;
        dx[r * lddx + c] = argrow[o * lda + c] == r ? dpool[o * ldp + c] : 0.f;
    }
}

extern "C" int tgp_colmax_bwd(const float *dpool, int ldp, const int *argrow, int lda, int objects, int rows_per_obj, int C, float *dx,
                              int lddx, tgp_stream_t stream)
{
    TGP_REQUIRE(dpool && argrow && dx && objects > 0 && rows_per_obj > 0 && C > 0 && ldp >= C && lda >= C && lddx >= C);
    const int64_t rows = (int64_t)objects * rows_per_obj;
    hipLaunchKernelGGL(colmax_bwd_kernel, dim3(tgp_cdiv(C, 64), tgp_cdiv(rows, (int64_t)BW_CHUNK)), dim3(256), 0, tgp_hs(stream), dpool,
                       ldp, argrow, lda, rows, rows_per_obj, C, dx, lddx);
    return TGP_LAUNCH_RESULT();
}

// dst (cols, rows) = src (rows, cols)^T   (weights, once per step, for da = dx W through the forward GEMM kernels)
__global__ void transpose_kernel(const float *__restrict__ src, int lds_, int rows, int cols, float *__restrict__ dst, int ldd)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = tile[threadIdx.x][j];
    }
}

extern "C" int tgp_transpose(const float *src, int ld_src, int rows, int cols, float *dst, int ld_dst, tgp_stream_t stream)
{
    TGP_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows);
    hipLaunchKernelGGL(transpose_kernel, dim3(tgp_cdiv(cols, 32), tgp_cdiv(rows, 32)), dim3(32, 8), 0, tgp_hs(stream), src, ld_src,
                       rows, cols, dst, ld_dst);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward GEMMs on the fp16 operand split.  A gradient tensor spans tens of binades below 1 -- outside fp16's range --
// but within ONE tensor the elements that matter lie within ~2^30 of the largest.  So: find max|dy| on the device, pick the
// power of two that puts it just under fp16's top, scale while transposing (exact), run the forward's split kernels
// (3 MFMA terms per product instead of 6 for bf16x3, or the fp32 MFMA of gemm_tn_kernel), unscale when the K-split
// partial sums are added up.  Elements below max * 2^-38 flush to zero, i.e. an error of 2^-38 relative to the largest
// element -- fp32 addition loses more.  Everything stays on the device: the step remains capturable in a graph.
// ---------------------------------------------------------------------------------------------------------------------
#define AM_THREADS 256
#define AM_BLOCKS 2048

__global__ __launch_bounds__(AM_THREADS) void absmax_partial_kernel(const float *__restrict__ x, int ld, int64_t rows, int cols,
                                                                    uint32_t *__restrict__ partial)
{
    __shared__ uint32_t red[AM_THREADS];
    const int64_t n = rows * cols;
    uint32_t m = 0;
    if (ld == cols && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // contiguous: 16-byte loads
        const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
        for (int64_t t = (int64_t)blockIdx.x * AM_THREADS + threadIdx.x; t < n / 4; t += (int64_t)gridDim.x * AM_THREADS) {
            const uint4 v = x4[t];
            const uint32_t a = (v.x & 0x7fffffffu) > (v.y & 0x7fffffffu) ? (v.x & 0x7fffffffu) : (v.y & 0x7fffffffu);
            const uint32_t b = (v.z & 0x7fffffffu) > (v.w & 0x7fffffffu) ? (v.z & 0x7fffffffu) : (v.w & 0x7fffffffu);
            const uint32_t u = a > b ? a : b;
            m = u > m ? u : m;
        }
    } else {
        for (int64_t t = (int64_t)blockIdx.x * AM_THREADS + threadIdx.x; t < n; t += (int64_t)gridDim.x * AM_THREADS) {
            const int64_t r = t / cols;
            const uint32_t u = __float_as_uint(x[r * ld + (t - r * cols)]) & 0x7fffffffu;      // |x| orders as an integer
            m = u > m ? u : m;
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = AM_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out = {s, 1 / s, max|x|}: s = 2^k with max|x| * s in [target / 2, target); s = 1 for an all-zero or non-finite tensor
__global__ __launch_bounds__(AM_THREADS) void absmax_final_kernel(const uint32_t *__restrict__ partial, int n, float target,
                                                                  float *__restrict__ out)
{
    __shared__ uint32_t red[AM_THREADS];
    uint32_t m = 0;
    for (int i = threadIdx.x; i < n; i += AM_THREADS) m = partial[i] > m ? partial[i] : m;
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = AM_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = __uint_as_float(red[0]);
        float s = 1.f;
        if (mx > 0.f && mx < INFINITY) {
            int e;
            frexpf(target / mx, &e);                           // target / mx = f * 2^e, f in [0.5, 1)
            e -= 1;                                            // 2^e <= target / mx
            e = e > 126 ? 126 : (e < -126 ? -126 : e);
            s = ldexpf(1.f, e);
        }
        out[0] = s, out[1] = 1.f / s, out[2] = mx;
    }
}

// words tgp_bn_bwd(absmax_bits) writes: one per workgroup of its apply pass
extern "C" int64_t tgp_bn_bwd_absmax_words(int64_t rows, int C) { return (int64_t)tgp_cdiv(C, 256) * tgp_cdiv(rows, (int64_t)BWV_CHUNK); }

// the scale from maxima already collected as bit patterns of |x| (tgp_bn_bwd's absmax_bits): {s, 1 / s, max}
extern "C" int tgp_absmax_scale_from_bits(const uint32_t *bits, int n, float target, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(bits && out && n > 0 && target > 0.f);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(AM_THREADS), 0, tgp_hs(stream), bits, n, target, out);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_absmax_scale(const float *x, int ld, int64_t rows, int cols, float target, uint32_t *workspace, float *out,
                                tgp_stream_t stream)
{
    TGP_REQUIRE(x && workspace && out && rows > 0 && cols > 0 && ld >= cols && target > 0.f);
    const int64_t n = rows * cols;
    const int blocks = (int)(n / AM_THREADS + 1 < AM_BLOCKS ? n / AM_THREADS + 1 : AM_BLOCKS);
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks), dim3(AM_THREADS), 0, tgp_hs(stream), x, ld, rows, cols, workspace);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(AM_THREADS), 0, tgp_hs(stream), workspace, blocks, target, out);
    return TGP_LAUNCH_RESULT();
}

// dst (cols, rows_pad) = (src (rows, cols) * *scale)^T, columns rows..rows_pad-1 zero.  SPLIT: dst is the fp16 hi / lo plane
// layout of tgp_split_f16 ([cols][rows_pad / 16][2][16]) instead of fp32.  64 x 64 tiles through LDS: rows of src are read
// as float4 along the columns; the transposed side is written in 16-byte pieces -- four consecutive source rows of one
// column as fp32, or eight of them as one plane's half K-tile (a K-tile's hi and lo halves are 64 contiguous bytes).
// rows_pad % 16 == 0 for the split form; fp32 form: rows_pad % 4 == 0.
#define TS_TILE 64
template <bool SPLIT>
__global__ __launch_bounds__(256) void transpose_scaled_kernel(const float *__restrict__ src, int lds_, int rows, int cols,
                                                               const float *__restrict__ scale, void *__restrict__ dst_, int rows_pad,
                                                               float *__restrict__ dst_f32 = nullptr)
{
    __shared__ float tile[TS_TILE][TS_TILE + 1];
    const int c0 = blockIdx.x * TS_TILE, r0 = blockIdx.y * TS_TILE;
    const float sc = scale ? scale[0] : 1.f;
    const bool vec = (lds_ & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (int t = threadIdx.x; t < TS_TILE * TS_TILE / 4; t += 256) {
        const int j = t / (TS_TILE / 4), q = t % (TS_TILE / 4);
        const int r = r0 + j, c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            if (vec && c + 3 < cols) {
                v = *reinterpret_cast<const float4 *>(src + (int64_t)r * lds_ + c);
            } else {
                if (c < cols) v.x = src[(int64_t)r * lds_ + c];
                if (c + 1 < cols) v.y = src[(int64_t)r * lds_ + c + 1];
                if (c + 2 < cols) v.z = src[(int64_t)r * lds_ + c + 2];
                if (c + 3 < cols) v.w = src[(int64_t)r * lds_ + c + 3];
            }
        }
        tile[j][4 * q] = v.x * sc, tile[j][4 * q + 1] = v.y * sc, tile[j][4 * q + 2] = v.z * sc, tile[j][4 * q + 3] = v.w * sc;
    }
    __syncthreads();
    if constexpr (SPLIT) {
        // item = (column, 8-row half of a 16-row K-tile): 64 columns x 8 halves per tile
        for (int t = threadIdx.x; t < TS_TILE * (TS_TILE / 8); t += 256) {
            const int cc = t / (TS_TILE / 8), hf = t % (TS_TILE / 8);
            const int c = c0 + cc, r = r0 + 8 * hf;
            if (c >= cols || r >= rows_pad) continue;
            alignas(16) uint16_t hi[8];
            alignas(16) uint16_t lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = tile[8 * hf + e][cc];
                const _Float16 hb = (_Float16)v;
                hi[e] = __builtin_bit_cast(uint16_t, hb);
                lo[e] = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)hb));
            }
            uint16_t *o = static_cast<uint16_t *>(dst_) + ((int64_t)c * (rows_pad / 16) + r / 16) * 32 + (r & 15);
            *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(hi);
            *reinterpret_cast<uint4 *>(o + 16) = *reinterpret_cast<const uint4 *>(lo);
            if (dst_f32) {       // the fp32 transpose beside the planes (tgp_transpose_both: one pass over a weight for the dx GEMM)
                float *f = dst_f32 + (int64_t)c * rows_pad + r;
                *reinterpret_cast<float4 *>(f) = make_float4(tile[8 * hf][cc], tile[8 * hf + 1][cc], tile[8 * hf + 2][cc], tile[8 * hf + 3][cc]);
                *reinterpret_cast<float4 *>(f + 4) = make_float4(tile[8 * hf + 4][cc], tile[8 * hf + 5][cc], tile[8 * hf + 6][cc], tile[8 * hf + 7][cc]);
            }
        }
    } else {
        for (int t = threadIdx.x; t < TS_TILE * (TS_TILE / 4); t += 256) {
            const int cc = t / (TS_TILE / 4), q = t % (TS_TILE / 4);
            const int c = c0 + cc, r = r0 + 4 * q;
            if (c >= cols || r >= rows_pad) continue;
            *reinterpret_cast<float4 *>(static_cast<float *>(dst_) + (int64_t)c * rows_pad + r) =
                make_float4(tile[4 * q][cc], tile[4 * q + 1][cc], tile[4 * q + 2][cc], tile[4 * q + 3][cc]);
        }
    }
}

extern "C" int tgp_transpose_scaled(const float *src, int ld_src, int rows, int cols, const float *scale, float *dst, int rows_pad,
                                    tgp_stream_t stream)
{
    TGP_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && rows_pad >= rows && (rows_pad & 3) == 0 &&
                (reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    hipLaunchKernelGGL((transpose_scaled_kernel<false>), dim3(tgp_cdiv(cols, TS_TILE), tgp_cdiv(rows_pad, TS_TILE)), dim3(256), 0,
                       tgp_hs(stream), src, ld_src, rows, cols, scale, dst, rows_pad);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_transpose_split_f16(const float *src, int ld_src, int rows, int cols, const float *scale, uint16_t *dst, int rows_pad,
                                       tgp_stream_t stream)
{
    TGP_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && rows_pad >= rows && (rows_pad & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    hipLaunchKernelGGL((transpose_scaled_kernel<true>), dim3(tgp_cdiv(cols, TS_TILE), tgp_cdiv(rows_pad, TS_TILE)), dim3(256), 0,
                       tgp_hs(stream), src, ld_src, rows, cols, scale, dst, rows_pad);
    return TGP_LAUNCH_RESULT();
}

// dst_f32 (cols, rows_pad) = src^T zero padded, dst_split = its fp16 hi / lo planes [cols][rows_pad / 16][2][16]: what the backward's
// dx GEMM takes as its weight (W^T and split_f16(W^T)), in one pass over W instead of a transpose, a padding copy and a split
extern "C" int tgp_transpose_both(const float *src, int ld_src, int rows, int cols, float *dst_f32, uint16_t *dst_split, int rows_pad,
                                  tgp_stream_t stream)
{
    TGP_REQUIRE(src && dst_f32 && dst_split && rows > 0 && cols > 0 && ld_src >= cols && rows_pad >= rows && (rows_pad & 15) == 0 &&
                ((reinterpret_cast<uintptr_t>(dst_f32) | reinterpret_cast<uintptr_t>(dst_split)) & 15) == 0);
    hipLaunchKernelGGL((transpose_scaled_kernel<true>), dim3(tgp_cdiv(cols, TS_TILE), tgp_cdiv(rows_pad, TS_TILE)), dim3(256), 0,
                       tgp_hs(stream), src, ld_src, rows, cols, (const float *)nullptr, dst_split, rows_pad, dst_f32);
    return TGP_LAUNCH_RESULT();
}

// out[i] (+)= *scale * sum_z parts[z * n + i], slabs added in order (deterministic)
__global__ void sum_slabs_kernel(const float *__restrict__ parts, int Z, int64_t n, const float *__restrict__ scale, float *__restrict__ out,
                                 int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = parts[i];
    for (int z = 1; z < Z; ++z) s += parts[(int64_t)z * n + i];
    s *= scale ? scale[0] : 1.f;
    out[i] = accumulate ? out[i] + s : s;
}

extern "C" int tgp_sum_slabs(const float *parts, int Z, int64_t n, const float *scale, float *out, int accumulate, tgp_stream_t stream)
{
    TGP_REQUIRE(parts && out && Z > 0 && n > 0);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(tgp_cdiv(n, 256)), dim3(256), 0, tgp_hs(stream), parts, Z, n, scale, out, accumulate);
    return TGP_LAUNCH_RESULT();
}
