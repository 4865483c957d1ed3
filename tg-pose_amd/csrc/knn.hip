// kNN graph construction for the TG-Pose forward path on gfx950.
//
// Reference semantics (network/fs_net_repo/gcn3d.py:14-35): distances are the fp32 expansion
// D_ij = fl(fl(-2*<a_i,a_j> + |a_j|^2) + |a_i|^2) computed with bmm / sum, then topk(k+1) and drop
// column 0.  The neighbour ORDER therefore depends on the exact rounding, which these kernels
// reproduce (DESIGN.md "kNN arithmetic"): <a,b> is an ascending-k FMA chain, |a|^2 follows ATen's
// cascade sum, ties are ordered by index.
//
// Selection: one 64-lane wavefront per query row.  Each lane keeps n/64 candidate distances in
// registers; k+1 rounds of (lane-local min, 6-step butterfly argmin over the wave, retire the
// winner) emit the neighbours in ascending (distance, index) order.  No N x N matrix exists for
// the 3-d case (the object's cloud sits in LDS as float4 {x,y,z,|p|^2}); the feature-space case
// reads rows of a distance matrix produced by the MFMA GEMM (gemm.hip, DIST epilogue).
#include <stdlib.h>

#include "tgp_common.h"

#define KNN_MAX_POINTS 2048
#define KNN_MAX_K 63

extern "C" int tgp_knn_max_points(void) { return KNN_MAX_POINTS; }
extern "C" int tgp_knn_max_k(void) { return KNN_MAX_K; }

// ------------------------------------------------------------------------------------------------
// ATen row_sum<float> over `size` elements spaced `stride` floats apart (SumKernel.cpp: 4
// interleaved accumulators, 16-step cascade levels).  Serial on purpose: it fixes the bits of
// the per-object mean, which fixes the centred cloud, which fixes the neighbour order.
__device__ float aten_row_sum_strided(const float *in, int stride, int size)
{
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f},
          a3[4] = {0.f, 0.f, 0.f, 0.f};
    const int size_ilp = size / 4;
    int i = 0;
    for (; i + 16 <= size_ilp;) {
        for (int j = 0; j < 16; ++j, ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a0[k] = a0[k] + in[(size_t)(i * 4 + k) * stride];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a1[k] = a1[k] + a0[k];
            a0[k] = 0.f;
        }
        if ((i & (15 << 4)) != 0) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a2[k] = a2[k] + a1[k];
            a1[k] = 0.f;
        }
        if ((i & (15 << 8)) != 0) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a3[k] = a3[k] + a2[k];
            a2[k] = 0.f;
        }
    }
    for (; i < size_ilp; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a0[k] = a0[k] + in[(size_t)(i * 4 + k) * stride];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a0[k] = a0[k] + a1[k];
        a0[k] = a0[k] + a2[k];
        a0[k] = a0[k] + a3[k];
    }
    for (int t = size_ilp * 4; t < size; ++t) a0[0] = a0[0] + in[(size_t)t * stride];
    a0[0] = a0[0] + a0[1];
    a0[0] = a0[0] + a0[2];
    a0[0] = a0[0] + a0[3];
    return 0.f + a0[0];
}

__global__ void center_mean_serial_kernel(const float *__restrict__ points, int B, int n, float *__restrict__ mean)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 3) return;
    const int b = t / 3, c = t - b * 3;
    const float s = aten_row_sum_strided(points + (size_t)b * n * 3 + c, 3, n);
    mean[t] = s / (float)n;
}

// The same sum, parallel: in ATen's cascade every run of 16 "rows" (4 interleaved elements each) is summed
// from zero before it is merged into the next level, so the runs are independent.  One workgroup per
// object: threads sum the runs, three threads then replay the merges in order.  Bit-identical to the
// serial routine above (tests compare both against torch on the CPU).
#define CM_MAX_RUNS 128
// out != NULL (round 4): the workgroup also writes its object's centred points -- tgp_center as one launch
// zero != NULL: the launch first clears zero_words 32-bit words there (the forward's arena of max keys, flags and magnitude words:
// tgp_center_zero) -- every later launch of the forward is ordered behind this one
__global__ __launch_bounds__(256) void center_mean_kernel(const float *__restrict__ points, int n, float *__restrict__ mean,
                                                          float *__restrict__ out, uint32_t *__restrict__ zero, int64_t zero_words)
{
    if (zero)
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < zero_words; t += (int64_t)gridDim.x * blockDim.x) zero[t] = 0u;
    __shared__ float part[3][4][CM_MAX_RUNS + 1];
    const int b = blockIdx.x;
    const float *in = points + (size_t)b * n * 3;
    const int size_ilp = n / 4, full = size_ilp / 16;
    for (int t = threadIdx.x; t < 12 * (full + 1); t += blockDim.x) {
        const int c = t % 3, k = (t / 3) & 3, j = t / 12;
        const int lo = j * 16, hi = (j < full) ? lo + 16 : size_ilp;
        float acc = 0.f;
        for (int i = lo; i < hi; ++i) acc = acc + in[(size_t)(i * 4 + k) * 3 + c];
        part[c][k][j] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        float fin[4];
        for (int k = 0; k < 4; ++k) {
            float a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int j = 0; j < full; ++j) {
                a1 = a1 + part[c][k][j];
                if (((j + 1) & 15) != 0) continue;
                a2 = a2 + a1;
                a1 = 0.f;
                if (((j + 1) & 255) != 0) continue;
                a3 = a3 + a2;
                a2 = 0.f;
            }
            float a0 = part[c][k][full];
            a0 = a0 + a1;
            a0 = a0 + a2;
            a0 = a0 + a3;
            fin[k] = a0;
        }
        for (int t = size_ilp * 4; t < n; ++t) fin[0] = fin[0] + in[(size_t)t * 3 + c];
        float s = fin[0] + fin[1];
        s = s + fin[2];
        s = s + fin[3];
        const float mu = (0.f + s) / (float)n;
        mean[b * 3 + c] = mu;
        part[c][0][0] = mu;
    }
    if (!out) return;
    __syncthreads();
    const float m3[3] = {part[0][0][0], part[1][0][0], part[2][0][0]};
    float *o = out + (size_t)b * n * 3;
    for (int t = threadIdx.x; t < n * 3; t += blockDim.x) o[t] = in[t] - m3[t % 3];
}

__global__ void center_sub_kernel(const float *__restrict__ points, const float *__restrict__ mean, int n,
                                  int64_t total, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t p = t / 3;
    const int c = (int)(t - p * 3);
    const int b = (int)(p / n);
    out[t] = points[t] - mean[b * 3 + c];
}

extern "C" int tgp_center_zero(const float *points, int B, int n, float *xyz_c, float *mean, void *zero, int64_t zero_words,
                               tgp_stream_t stream);
extern "C" int tgp_center(const float *points, int B, int n, float *xyz_c, float *mean, tgp_stream_t stream)
{
    return tgp_center_zero(points, B, n, xyz_c, mean, nullptr, 0, stream);
}

extern "C" int tgp_center_zero(const float *points, int B, int n, float *xyz_c, float *mean, void *zero, int64_t zero_words,
                               tgp_stream_t stream)
{
    TGP_REQUIRE(points && xyz_c && mean && B > 0 && n > 0 && zero_words >= 0 && (!zero_words || zero));
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(zero) & 3) == 0);
    if (n / 64 < CM_MAX_RUNS) {
        hipLaunchKernelGGL(center_mean_kernel, dim3(B), dim3(256), 0, tgp_hs(stream), points, n, mean, xyz_c,
                           zero_words ? reinterpret_cast<uint32_t *>(zero) : nullptr, zero_words);
        return TGP_LAUNCH_RESULT();
    }
    if (zero_words) {
        const hipError_t e = hipMemsetAsync(zero, 0, (size_t)zero_words * 4, tgp_hs(stream));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(center_mean_serial_kernel, dim3(tgp_cdiv(B * 3, 64)), dim3(64), 0, tgp_hs(stream), points, B, n, mean);
    const int64_t total = (int64_t)B * n * 3;
    hipLaunchKernelGGL(center_sub_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), points, mean, n,
                       total, xyz_c);
    return TGP_LAUNCH_RESULT();
}

// ------------------------------------------------------------------------------------------------
// wave-wide top-(k+1) selection over NT register slots per lane (slot t of lane l is candidate j = l + 64 t).
//
// NT <= 2: every round scans the slots.  NT > 2: each lane first builds a sorted cache of its 3 smallest
// (distance, slot) keys in one pass; a round is then a 6-step butterfly argmin over the cache heads plus a pop on
// the winning lane.  A lane rarely owns more than 3 of the k+1 nearest (0.5 % of lanes at k = 20, n = 1028); when one
// runs dry the whole wave rebuilds its caches from the keys greater than the last one each lane gave away.
// Wave-wide unsigned minimum on the VALU: four DPP steps (quad swaps, half-row and row mirrors) leave every 16-lane row
// uniform, the four row values are combined on the scalar unit.  No LDS crossbar (ds_bpermute) round trips: the selection
// below is a chain of dependent wave reductions, and each __shfl_xor in it cost a crossbar latency.
__device__ __forceinline__ uint32_t wave_umin(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); // row_mirror
    v = o < v ? o : v;
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

// (distance, index) argmin over the wave: the smallest order-preserving key, then the smallest index among its holders
__device__ __forceinline__ int wave_argmin(float best, int bj)
{
    const uint32_t key = tgp_float_key(best);
    const uint32_t mk = wave_umin(key);
    const unsigned long long holders = __ballot(key == mk);
    if (__popcll(holders) == 1)                                  // wave-uniform: the usual case, one lane holds the minimum
        return __builtin_amdgcn_readlane(bj, __ffsll((long long)holders) - 1);
    return (int)wave_umin(key == mk ? (uint32_t)bj : 0xffffffffu);
}

template <int NT>
__device__ __forceinline__ void wave_select(float (&d)[NT], int lane, int k, int32_t *__restrict__ out_row)
{
    int mine = 0; // lane r-1 keeps the neighbour of rank r
    if constexpr (NT <= 2) {
        for (int r = 0; r <= k; ++r) {
            float best = d[0];
            int bt = 0;
#pragma unroll
            for (int t = 1; t < NT; ++t) {
                const bool lt = d[t] < best;
                best = lt ? d[t] : best;
                bt = lt ? t : bt;
            }
            int bj = wave_argmin(best, lane + (bt << 6));
            const int owner = bj & 63, slot = bj >> 6;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t == slot && lane == owner) d[t] = INFINITY;
            if (r >= 1 && lane == r - 1) mine = bj;
        }
    } else {
        float c0, c1, c2;      // cache: c0 <= c1 <= c2, equal distances in slot order
        int s0, s1, s2;
        float lastd = -INFINITY; // the last key this lane gave away: (lastd, lasts)
        int lasts = -1;
        auto refill = [&]() {
            c0 = c1 = c2 = INFINITY;
            s0 = s1 = s2 = NT;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float v = d[t];
                const bool after = (v > lastd) || (v == lastd && t > lasts);
                const bool lt0 = after && v < c0, lt1 = after && v < c1, lt2 = after && v < c2;
                c2 = lt1 ? c1 : (lt2 ? v : c2);
                s2 = lt1 ? s1 : (lt2 ? t : s2);
                c1 = lt0 ? c0 : (lt1 ? v : c1);
                s1 = lt0 ? s0 : (lt1 ? t : s1);
                c0 = lt0 ? v : c0;
                s0 = lt0 ? t : s0;
            }
        };
        refill();
        for (int r = 0; r <= k; ++r) {
            // an empty cache (s0 == NT) carries distance +inf and never wins a finite round
            const int bj = wave_argmin(c0, lane + (s0 << 6));
            if (r >= 1 && lane == r - 1) mine = bj;
            bool dry = false;
            if (lane == (bj & 63)) { // pop
                lastd = c0, lasts = s0;
                c0 = c1, s0 = s1;
                c1 = c2, s1 = s2;
                c2 = INFINITY, s2 = NT;
                dry = (s0 == NT);
            }
            if (__any(dry)) refill();
        }
    }
    if (lane < k) out_row[lane] = mine;
}

// Wave-wide unsigned maximum (same DPP steps as wave_umin)
__device__ __forceinline__ uint32_t wave_umax(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
    v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);
    v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);
    v = o > v ? o : v;
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// Top-(k+1) of one row WITHOUT a serial chain of k+1 wave reductions (the fused feature-space kernel's consumers run one wave
// per SIMD: a chain of dependent cross-lane steps would leave the vector pipe idle most of the time).  key[t] is the
// order-preserving key of candidate lane + 64 t; the order is (key, index), as in wave_select.
//   1. every lane takes the minimum of its own keys; its rank among the 64 lane minima is a count of smaller ones (the minima
//      go through LDS, every lane reads all 64: no dependency between the compares);
//   2. tau = the largest lane minimum of rank <= k: at least k + 1 candidates are <= tau, so nothing above tau can be among the
//      k + 1 nearest; on average ~26 of the 1028 candidates survive;
//   3. the survivors are compacted into a per-wave list in LDS (ballot prefix sums), lane l takes survivor l and counts the
//      survivors that precede it in (key, index) order: that count is its rank; ranks 1 .. k are written out.
// Returns false (wave-uniform, nothing written) if more than 64 candidates survive; the caller then runs wave_select.
// (round 5) slot / nbr (may be NULL): the list entry this LANE wrote -- slot in [0, k), its neighbour's id -- or slot = -1: the caller that
// goes on to use the list (the neighbour directions) has it in registers and need not wait for its own stores and read them back.
// (round 5) The compiler turned the first form's booleans into 64-bit lane masks in scalar registers: every (key, index) comparison was
// three vector compares, two scalar mask operations and a select, every conditional list write a save / restore of the execution
// mask -- ~110 mask operations and ~120 hazard no-ops per row, each a round trip between the vector and the scalar unit on a wave that
// runs alone: ~10 k cycles per row (scripts/knn_pc_stamps.py), which -- not the matrix pipe -- bounded the fused kernels.  Now an entry is
// ONE 64-bit integer (key << 32 | index): a comparison is v_cmp_lt_u64 + add-with-carry; and every candidate is WRITTEN, a survivor to
// its place in the list, the others to a scratch entry of the lane's own (no execution-mask changes).  Same comparisons, same lists.
#define KNN_LIST 136                       // entries per wave: 64 survivors + 8 sentinels, then one scratch entry per lane
template <int NT>
__device__ __forceinline__ bool wave_select_ranked(const uint32_t (&key)[NT], int lane, int k, int32_t *__restrict__ out_row,
                                                   uint32_t *__restrict__ lmin, uint2 *__restrict__ list /* KNN_LIST entries, 16-byte aligned */,
                                                   int *slot = nullptr, int *nbr = nullptr, unsigned long long *ph = nullptr)
{
#define KNN_PH(I) do { if (ph) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[I] += t_ - ph[5]; ph[5] = t_; } } while (0)
    if (ph) ph[5] = __builtin_readcyclecounter();
    uint32_t mn = key[0];
#pragma unroll
    for (int t = 1; t < NT; ++t) mn = key[t] < mn ? key[t] : mn;
    lmin[lane] = mn;
    __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): the wave's own LDS writes have landed
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint4 v = reinterpret_cast<const uint4 *>(lmin)[j];     // broadcast reads
        rank += (v.x < mn) + (v.y < mn) + (v.z < mn) + (v.w < mn);
    }
    const uint32_t tau = wave_umax(rank <= k ? mn : 0u);
    KNN_PH(0);
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) cnt += key[t] <= tau;
    // exclusive prefix sum of cnt over the lanes, bit by bit: ballots and bit counts, no cross-lane data movement.  cnt <= NT
    // INCLUSIVE (a lane whose NT candidates all lie under tau: coincident points), so the loop covers the bits of NT itself --
    // knn_xyz_kernel<32> needs six; with five a full lane counted as empty and `total` wrapped below the 64-survivor test.
    constexpr int CNT_BITS = NT >= 32 ? 6 : NT >= 16 ? 5 : NT >= 8 ? 4 : NT >= 4 ? 3 : NT >= 2 ? 2 : 1;
    static_assert(NT < (1 << CNT_BITS) && NT <= 32, "cnt must fit the ballot prefix sum");
    int pos = 0, total = 0;
#pragma unroll
    for (int bit = 0; bit < CNT_BITS; ++bit) {
        const unsigned long long m = __ballot((cnt >> bit) & 1);
        pos += (int)(__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) << bit;
        total += __popcll(m) << bit;
    }
    KNN_PH(1);
    if (total > 64) return false;                                     // wave-uniform
    // an entry = (index, key) as it lies: read as one 64-bit integer it is key << 32 | index, the (key, index) order
    unsigned long long *list64 = reinterpret_cast<unsigned long long *>(list);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool s = key[t] <= tau;
        list64[s ? pos : 72 + lane] = ((unsigned long long)key[t] << 32) | (uint32_t)(lane + (t << 6));
        pos += s;
    }
    if (lane < 8) list64[total + lane] = ~0ull;                       // sentinels: the rank loop reads whole groups of 8
    __builtin_amdgcn_s_waitcnt(0xc07f);
    KNN_PH(2);
    const unsigned long long mine = list64[lane < total ? lane : 0];
    int r = 0;
    for (int j = 0; j < total; j += 8) {                              // wave-uniform trip count; broadcast reads, issued together
        ulonglong2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = reinterpret_cast<const ulonglong2 *>(list64 + j)[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) r += (e[u].x < mine) + (e[u].y < mine);
    }
    const bool has = lane < total && r >= 1 && r <= k;
    if (has) out_row[r - 1] = (int32_t)(uint32_t)mine;
    if (slot) *slot = has ? r - 1 : -1, *nbr = (int)(uint32_t)mine;
    KNN_PH(3);
#undef KNN_PH
    return true;
}

// The serial form on keys, compact (a runtime loop of k + 1 rounds, each a lane-local scan for the smallest key after the last
// one emitted plus two wave reductions): the fall-back of wave_select_ranked, rare, so size matters more than speed.
template <int NT>
__device__ __forceinline__ void wave_select_serial_keys(const uint32_t (&key)[NT], int lane, int k, int32_t *__restrict__ out_row,
                                                        int *slot = nullptr, int *nbr = nullptr)
{
    uint32_t lastk = 0;
    int lastj = -1, mine = 0;
    for (int r = 0; r <= k; ++r) {
        uint32_t bk = 0xffffffffu;
        int bj = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + (t << 6);
            const bool after = key[t] > lastk || (key[t] == lastk && j > lastj);
            const bool better = after && (key[t] < bk || (key[t] == bk && j < bj));
            bk = better ? key[t] : bk;
            bj = better ? j : bj;
        }
        lastk = wave_umin(bk);
        lastj = (int)wave_umin(bk == lastk ? (uint32_t)bj : 0x7fffffffu);
        if (r >= 1 && lane == r - 1) mine = lastj;
    }
    if (lane < k) out_row[lane] = mine;
    if (slot) *slot = lane < k ? lane : -1, *nbr = mine;
}

template <int NT>
__global__ __launch_bounds__(256) void knn_xyz_kernel(const float *__restrict__ xyz, int B, int n, int k,
                                                      int32_t *__restrict__ idx, int tiles_per_obj, int rpb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);
    uint2 *s_list = reinterpret_cast<uint2 *>(smem + (size_t)n * sizeof(float4));          // [4 waves][KNN_LIST]
    uint32_t *s_lmin = reinterpret_cast<uint32_t *>(s_list + 4 * KNN_LIST);                   // [4 waves][64]
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const float *xb = xyz + (size_t)b * n * 3;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float x = xb[j * 3 + 0], y = xb[j * 3 + 1], z = xb[j * 3 + 2];
        float q = x * x;          // torch.sum(v**2, dim=2) over 3 elements: ((0+x^2)+y^2)+z^2
        q = q + y * y;
        q = q + z * z;
        pts[j] = make_float4(x, y, z, q);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rpb; r += 4) {
        const int i = tile * rpb + r;
        if (i >= n) break;
        const float4 pi = pts[i];
        uint32_t key[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + (t << 6);
            float dv = INFINITY;
            if (j < n) {
                const float4 pj = pts[j];
                float inner = pi.x * pj.x;         // bmm(v, v^T): ascending-k FMA chain from 0
                inner = fmaf(pi.y, pj.y, inner);
                inner = fmaf(pi.z, pj.z, inner);
                const float t1 = inner * -2.0f;
                const float t2 = t1 + pj.w;        // + quadratic.unsqueeze(1)  (column term)
                dv = t2 + pi.w;                    // + quadratic.unsqueeze(2)  (row term)
            }
            key[t] = tgp_float_key(dv);
        }
        // rank-counting selection (no chain of k + 1 wave reductions); clouds with many coincident points can leave more than 64
        // candidates under its bound: those rows take the serial form
        int32_t *out = idx + ((size_t)b * n + i) * k;
        if (!wave_select_ranked<NT>(key, lane, k, out, s_lmin + wave * 64, s_list + wave * KNN_LIST))
            wave_select_serial_keys<NT>(key, lane, k, out);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void knn_matrix_kernel(const float *__restrict__ D, int B, int n, int k,
                                                         int32_t *__restrict__ idx, int tiles_per_obj, int rpb)
{
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rpb; r += 4) {
        const int i = tile * rpb + r;
        if (i >= n) break;
        const float *row = D + ((size_t)b * n + i) * n;
        float d[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + (t << 6);
            d[t] = (j < n) ? row[j] : INFINITY;
        }
        wave_select<NT>(d, lane, k, idx + ((size_t)b * n + i) * k);
    }
}

// Query rows per workgroup: one wave handles rpb/4 rows back to back.  The selection is a chain of
// dependent cross-lane steps, so it needs many waves per SIMD to hide latency: aim at >= 2048 workgroups.
static int knn_rows_per_block(int B, int n)
{
    int rpb = 32;
    while (rpb > 4 && (int64_t)B * tgp_cdiv(n, rpb) < 2048) rpb >>= 1;
    return rpb;
}

static int knn_check(int B, int n, int k)
{
    if (B <= 0 || n <= 0 || k <= 0) return TGP_EINVAL;
    if (n > KNN_MAX_POINTS || k > KNN_MAX_K || k + 1 > n) return TGP_EUNSUPPORTED;
    return 0;
}

extern "C" int tgp_knn_xyz(const float *xyz, int B, int n, int k, int32_t *idx, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && idx);
    const int chk = knn_check(B, n, k);
    if (chk) return chk;
    const int rpb = knn_rows_per_block(B, n);
    const int tiles = tgp_cdiv(n, rpb);
    const dim3 grid(tgp_xcd_grid(B, tiles)), block(256);
    const size_t lds = (size_t)n * sizeof(float4) + 4 * (KNN_LIST * sizeof(uint2) + 64 * sizeof(uint32_t));
    const int nt = tgp_cdiv(n, 64);
#define LAUNCH_XYZ(NT) \
    hipLaunchKernelGGL(knn_xyz_kernel<NT>, grid, block, lds, tgp_hs(stream), xyz, B, n, k, idx, tiles, rpb)
    if (nt <= 1) LAUNCH_XYZ(1);
    else if (nt <= 2) LAUNCH_XYZ(2);
    else if (nt <= 5) LAUNCH_XYZ(5);
    else if (nt <= 8) LAUNCH_XYZ(8);
    else if (nt <= 17) LAUNCH_XYZ(17);
    else LAUNCH_XYZ(32);
#undef LAUNCH_XYZ
    return TGP_LAUNCH_RESULT();
}

// |x_r|^2 in ATen's vectorised cascade order (8-wide vectors, 4 interleaved accumulators,
// lanes summed last).  8 threads cooperate on a row; d % 32 == 0 and d / 32 < 16 (no cascade level).
__global__ __launch_bounds__(256) void sqnorm_aten_kernel(const float *__restrict__ x, int ld, int64_t rows, int d,
                                                          float *__restrict__ q)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l8 = threadIdx.x & 7;
    const bool live = row < rows;
    const float *xr = x + (live ? row : 0) * (int64_t)ld;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int size_ilp = d >> 5;
    for (int i = 0; i < size_ilp; ++i) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float v = xr[((i << 2) + kk) * 8 + l8];
            acc[kk] = acc[kk] + v * v;
        }
    }
    float p = acc[0] + acc[1];
    p = p + acc[2];
    p = p + acc[3];
    float fin = 0.f;
    const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin = fin + __shfl(p, base + l, 64);
    if (live && l8 == 0) q[row] = 0.f + fin;
}

// ------------------------------------------------------------------------------------------------
// Feature-space kNN WITHOUT the (B, n, n) matrix (gcn3d.py:14-23 with d = 128 / 256).  One 512-thread workgroup per 32-row
// block of an object's distance matrix, two phases:
//   1. all eight waves compute the block's 32 x n distances on the fp32 matrix cores -- the block's own 32 feature rows stay
//      in registers as the A operand (lane (r, h) holds x[i0 + r][2 s + h] for every MFMA step s), the object's other rows
//      stream past as the B operand straight from the XCD's L2 (an object's features are 0.5 MB; all its row blocks run on one
//      XCD), prefetched one chunk ahead -- in the same ascending-k FMA chain and the same three roundings as the stored-matrix
//      form (bit-identical distances), and leave them in LDS (32 x n floats, 135 KB at n = 1028);
//   2. every wave takes four rows out of LDS and selects their k + 1 nearest by (distance, index) with wave_select_ranked.
// The distances never leave the CU: HBM sees the features once (16.8 MB at B = 32, n = 1028, d = 128) and the index lists
// (2.6 MB) instead of 2 x 135 MB of matrix.
// Tail rows.  The LDS block admits one workgroup per CU, so the launch runs in rounds of 256 row blocks, and the network's
// clouds are 32 m + 4 (1028) or 32 m + 1 (257) rows: a 33rd / 9th block per object with 4 / 1 live rows cost a whole extra
// round (1056 blocks = 4.1 rounds, 288 = 1.1).  When the tail is that short (n_extra <= 8 rows) those rows ride along instead:
// block rb < n_extra also takes row 32 * nrb + rb, its distances computed on the vector pipe (the same ascending-k FMA chain,
// one column per thread and pass) into a 33rd LDS row, selected by wave 7 after its four rows.
#ifdef TGP_DEV   // development build: rows whose rank-counting selection met more than 64 survivors and took the serial form
__device__ unsigned long long tgp_knn_serial_rows = 0ull;
extern "C" int tgp_debug_knn_serial_rows(unsigned long long *out, int reset)
{
    unsigned long long z = 0ull;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_knn_serial_rows), sizeof(z));
    if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(tgp_knn_serial_rows), &z, sizeof(z));
    return (int)e;
}
#define KNN_COUNT_SERIAL() do { if (lane == 0) atomicAdd(&tgp_knn_serial_rows, 1ull); } while (0)
#else
#define KNN_COUNT_SERIAL() do { } while (0)
#endif
typedef float knn_f32x16 __attribute__((ext_vector_type(16)));

#define KF_ROWS 32
#define KF_MAX_LDW 1152      // 33 x 1152 x 4 B = 148.5 KiB of LDS (+ 6.5 KiB of selection scratch)

// xt: the object's features transposed, (B, DIM, ldw) with ldw = 32 * ncb columns (zero beyond n): lane (r, h) of a 32-wide
// block then reads xt[k = 2 s + h][32 cb + r] -- 32 consecutive floats per half wave, a fully used 128-byte line -- where a
// row-major operand would make every lane walk its own row (32 lines touched per load instruction for 16 useful bytes each;
// measured: the vector-memory pipe, not the matrix cores, then sets the pace: 43 instead of 18 us per row block).
template <int DIM, int NT, int CH>   // CH: MFMA steps (k pairs) per prefetched chunk
__global__ __launch_bounds__(512, 2) void knn_feat_fused_kernel(const float *__restrict__ xt, const float *__restrict__ q, int B, int n, int k,
                                                                int32_t *__restrict__ idx, int nrb, int ncb, int ldw, int n_extra,
                                                                unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) float dblk[];      // [32 (+ 1 with tail rows)][ldw]
    __shared__ uint32_t s_lmin[8][64];
    __shared__ __attribute__((aligned(16))) uint2 s_list[8][KNN_LIST];
    constexpr int STEPS = DIM / 2;                                    // MFMA steps per column block (k pairs)
    constexpr int NCHUNK = STEPS / CH;
    constexpr int RING = 4;                                           // B-operand chunks in flight: three ahead of the MFMAs
    static_assert(NCHUNK % RING == 0, "the chunk ring's phase must repeat per column block");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    int b, rb;
    if (!tgp_xcd_object_tile(blockIdx.x, B, nrb, b, rb)) return;
    const int i0 = rb * KF_ROWS;
    const bool has_extra = rb < n_extra;                              // workgroup-uniform
    const int ix = nrb * KF_ROWS + rb;                                // the tail row this block takes along
    const float *xb = xt + (size_t)b * DIM * ldw + (size_t)h * ldw + r;      // + 2 s ldw + 32 cb: the lane's element of step s, block cb
    const float *qb = q + (size_t)b * n;
    // development builds: 12 stamps per workgroup -- start, wave 0's phase 1 end, after the barrier, end; then each wave's phase-1 end
    unsigned long long *st = (stamps && threadIdx.x == 0) ? stamps + 12 * (size_t)blockIdx.x : nullptr;
    if (st) st[0] = __builtin_amdgcn_s_memrealtime();
    {
        // ---- phase 1: distances of rows [i0, i0 + 32) against the column blocks cb = wave, wave + 8, ...
        float a[STEPS];
#pragma unroll
        for (int s2 = 0; s2 < STEPS; ++s2) a[s2] = xb[(size_t)2 * s2 * ldw + i0];
        float qrow[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) qrow[e] = qb[min(i0 + (e & 3) + 8 * (e >> 2) + 4 * h, n - 1)];
        // a chunk of CH MFMA steps is CH x 64 cycles of matrix work (0.27 us at CH = 8) against ~1 us of loaded L2 latency: with one
        // chunk of lookahead the matrix cores waited for operands 40 % of this phase; three chunks ahead they do not
        float buf[RING][CH];
        auto fetch = [&](int cb, int c, float (&dst)[CH]) {
            const float *br = xb + (size_t)2 * c * CH * ldw + cb * 32;
#pragma unroll
            for (int t = 0; t < CH; ++t) dst[t] = br[(size_t)2 * t * ldw];
        };
        if (wave < ncb) {
#pragma unroll
            for (int c = 0; c < RING - 1; ++c) fetch(wave, c, buf[c]);
        }
        for (int cb = wave; cb < ncb; cb += 8) {
            knn_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {                        // compile-time chunk index: a[] and the ring stay in registers
                constexpr int AHEAD = RING - 1;
                if (c + AHEAD < NCHUNK) fetch(cb, c + AHEAD, buf[(c + AHEAD) % RING]);
                else if (cb + 8 < ncb) fetch(cb + 8, c + AHEAD - NCHUNK, buf[(c + AHEAD) % RING]);   // NCHUNK % RING == 0: same slot
#pragma unroll
                for (int t = 0; t < CH; ++t)                          // ascending k: step s = c CH + t takes k = 2 s + h
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c * CH + t], buf[c % RING][t], acc, 0, 0, 0);
            }
            const int col = cb * 32 + r;
            const float qc = qb[min(col, n - 1)];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float t1 = acc[e] * -2.0f;                      // inner * (-2)
                const float t2 = t1 + qc;                             // + quadratic.unsqueeze(1)
                dblk[row * ldw + col] = col < n ? t2 + qrow[e] : INFINITY;           // + quadratic.unsqueeze(2)
            }
        }
    }
    if (has_extra && wave != 0) {
        // the tail row against every column: inner = x[0] y[0], then fmaf in ascending k -- what the MFMA chain computes.  Done by
        // waves 1-7, which have one column block less than wave 0 (33 = 8 x 4 + 1): three columns per thread, their chains
        // interleaved so that 48 loads are in flight, in the shadow of wave 0's fifth block
        const float *xo = xt + (size_t)b * DIM * ldw;
        const float qx = qb[ix];
        const int t0 = (wave - 1) * 64 + lane;
        for (int cbase = t0; cbase < ldw; cbase += 3 * 448) {
            const int c0 = cbase, c1 = cbase + 448, c2 = cbase + 896;
            const int l0 = c0, l1 = min(c1, ldw - 1), l2 = min(c2, ldw - 1);
            const float x0 = xo[ix];
            float in0 = x0 * xo[l0], in1 = x0 * xo[l1], in2 = x0 * xo[l2];
#pragma unroll 16
            for (int kk = 1; kk < DIM; ++kk) {
                const float xv = xo[(size_t)kk * ldw + ix];
                in0 = fmaf(xv, xo[(size_t)kk * ldw + l0], in0);
                in1 = fmaf(xv, xo[(size_t)kk * ldw + l1], in1);
                in2 = fmaf(xv, xo[(size_t)kk * ldw + l2], in2);
            }
            auto put = [&](int col, float inner) {
                if (col < ldw) {
                    const float t1 = inner * -2.0f;
                    const float t2 = t1 + qb[min(col, n - 1)];
                    dblk[KF_ROWS * ldw + col] = col < n ? t2 + qx : INFINITY;
                }
            };
            put(c0, in0), put(c1, in1), put(c2, in2);
        }
    }
    if (st) st[1] = __builtin_amdgcn_s_memrealtime();
    if (stamps && lane == 0) stamps[12 * (size_t)blockIdx.x + 4 + wave] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (st) st[2] = __builtin_amdgcn_s_memrealtime();
    // ---- phase 2: four rows per wave
#pragma unroll 1
    for (int rr = 0; rr < 5; ++rr) {
        if (rr == 4 && !(has_extra && wave == 7)) break;              // wave-uniform: the tail row is wave 7's fifth
        const int lrow = rr == 4 ? KF_ROWS : wave * 4 + rr, i = rr == 4 ? ix : i0 + lrow;
        if (i >= n) break;                                            // wave-uniform
        uint32_t key[NT];
        const float *row = dblk + lrow * ldw;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + (t << 6);
            key[t] = tgp_float_key(j < n ? row[j] : INFINITY);
        }
        int32_t *out = idx + ((size_t)b * n + i) * k;
        if (!wave_select_ranked<NT>(key, lane, k, out, s_lmin[wave], s_list[wave]))
            wave_select_serial_keys<NT>(key, lane, k, out);           // more than 64 candidates under the bound (rare)
    }
    if (st) st[3] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------------
// (round 4) The same on 16-row blocks: 256 threads, v_mfma_f32_16x16x4_f32, 16 x ldw distances in LDS (<= 74 KB) -- TWO workgroups per
// CU.  The 32-row form holds the CU alone (135 KB of LDS) and runs its two phases one after the other: 28 us of matrix work during
// which the vector pipe idles, 11 us of selection during which the matrix cores idle (profiles/r02_f_knn_feat_fused_*.txt).  Two
// independent half-size workgroups drift apart by themselves, one selecting while the other multiplies; the producer / consumer split
// INSIDE a workgroup (round 2) failed because one MFMA wave per SIMD could not feed the pipe -- here every wave keeps four
// independent accumulator chains (four interleaved column sets), enough to fill it alone.
// Arithmetic: lane (c, g) of the 16x16x4 instruction supplies k = 4 s + g of row / column c in step s, and the matrix core adds the
// four products to the accumulator in ascending k with one rounding each -- the same chain as two steps of the 32x32x2 form, so the
// distances carry the same bits (tests/test_gpu_parity.py::test_knn_feat_bit_exact_vs_oracle runs both forms).
// Operands: A = the block's 16 rows, one float per lane and step (registers); B = four column sets at once -- lane (c, g) loads the
// float4 xt[k][64 grp + 4 c ..+3] (a wave-instruction reads four 256-byte runs), component j being column 64 grp + 4 c + j of set j.
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));

#define KF16_ROWS 16
// Tail rows: the network's clouds are 16 m + 4 (1028) or 16 m + 1 (257) rows, and a last block with 4 / 1 live rows per object is a
// whole extra round of workgroups (2080 = 4.06 rounds of 512 slots; 544 = 1.06).  As in the 32-row form, a short tail (n_extra <= 8
// rows) rides along: block rb < n_extra also takes row 16 nrb + rb, its distances computed on the vector pipe (the same ascending-k
// FMA chain) by waves 1-3 -- which have one column group less than wave 0 -- into a 17th LDS row that wave 3 selects after its four.
// (Measured and dropped: a start skew -- the workgroups of the launch's second half-round, i.e. the second slot of every CU, pausing for
// about one selection phase once, so that the two workgroups of a CU would not run their phases in step: 18.92 k against 18.97 k
// objects/s without it, two runs each on one box; the workgroups drift apart by themselves within a round.)
// (round 4) The unit directions to a row's neighbours (gcn3d.py:48-58 get_neighbor_direction_norm), for the LDS-staged graph
// convolution that walks this list next (gconv.hip, nbr_dirs_kernel: the same arithmetic) -- written by the wave that has just
// selected the row, instead of by a launch of its own.  The wave reads its own list back: its stores are acknowledged (vmcnt 0)
// and the loads go past the L1 (sc1).
__device__ __forceinline__ void knn_emit_dirs(const float *__restrict__ xyz, float4 *__restrict__ dirs, const int slot, const int nb,
                                              const int64_t rowi, const int b, const int n, const int k)
{
    // (round 5) from the selecting lane's registers.  Round 4 read the list back -- `s_waitcnt vmcnt(0)` for the stores' acknowledgements,
    // then sc1 loads, then the two coordinate loads: four dependent memory round trips per row on a wave that selects four rows per
    // block, ~5 us each -- which, not the matrix pipe, was what the kernel waited for (its selection phase took ~30 us per block).
    if (slot < 0) return;
    const float *pc = xyz + rowi * 3;
    const float *pn = xyz + ((int64_t)b * n + nb) * 3;
    const float dx = pn[0] - pc[0], dy = pn[1] - pc[1], dz = pn[2] - pc[2];
    const float nrm = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
    dirs[rowi * k + slot] = make_float4(dx / nrm, dy / nrm, dz / nrm, 0.f);
}

template <int DIM, int NT, int CH>
__global__ __launch_bounds__(256, 2) void knn_feat_fused16_kernel(const float *__restrict__ xt, const float *__restrict__ q, int B, int n, int k,
                                                                  int32_t *__restrict__ idx, int nrb, int ldw, int n_extra,
                                                                  const float *__restrict__ xyz, float4 *__restrict__ dirs)
{
    extern __shared__ __attribute__((aligned(16))) float dblk16[];    // [16 (+ 1 with a tail row)][ldw]
    __shared__ uint32_t s_lmin[4][64];
    __shared__ __attribute__((aligned(16))) uint2 s_list[4][KNN_LIST];
    constexpr int STEPS = DIM / 4;                                    // MFMA steps per column group (k quads)
    constexpr int NCHUNK = STEPS / CH;
    constexpr int RING = 4;
    static_assert(NCHUNK % RING == 0, "the chunk ring's phase must repeat per column group");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    int b, rb;
    if (!tgp_xcd_object_tile(blockIdx.x, B, nrb, b, rb)) return;
    const int i0 = rb * KF16_ROWS;
    const float *xo = xt + (size_t)b * DIM * ldw + (size_t)g * ldw;   // + 4 s ldw: the lane's k of step s
    const float *qb = q + (size_t)b * n;
    const int ngrp = (ldw + 63) >> 6;                                  // column groups of 64 (the last may hold 32: ldw % 32 == 0)
    const bool has_extra = rb < n_extra;                              // workgroup-uniform
    const int ix = nrb * KF16_ROWS + rb;                              // the tail row this block takes along
    {
        float a[STEPS];
#pragma unroll
        for (int s2 = 0; s2 < STEPS; ++s2) a[s2] = xo[(size_t)4 * s2 * ldw + i0 + c];
        float qrow[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) qrow[e] = qb[min(i0 + 4 * g + e, n - 1)];
        float4 buf[RING][CH];
        auto fetch = [&](int grp, int cc, float4 (&dst)[CH]) {
            // (a group past ldw -- the half group of an odd number of 32-column blocks -- re-reads the previous columns; masked below)
            const int col = min(64 * grp + 4 * c, ldw - 4);
            const float *br = xo + (size_t)4 * cc * CH * ldw + col;
#pragma unroll
            for (int t = 0; t < CH; ++t) dst[t] = *reinterpret_cast<const float4 *>(br + (size_t)4 * t * ldw);
        };
        if (wave < ngrp) {
#pragma unroll
            for (int cc = 0; cc < RING - 1; ++cc) fetch(wave, cc, buf[cc]);
        }
        for (int grp = wave; grp < ngrp; grp += 4) {
            knn_f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = knn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < NCHUNK; ++cc) {
                constexpr int AHEAD = RING - 1;
                if (cc + AHEAD < NCHUNK) fetch(grp, cc + AHEAD, buf[(cc + AHEAD) % RING]);
                else if (grp + 4 < ngrp) fetch(grp + 4, cc + AHEAD - NCHUNK, buf[(cc + AHEAD) % RING]);
#pragma unroll
                for (int t = 0; t < CH; ++t) {                        // ascending k: step s = cc CH + t takes k = 4 s + g
                    const float av = a[cc * CH + t];
                    const float4 bv = buf[cc % RING][t];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.w, acc[3], 0, 0, 0);
                }
            }
            // accumulator element e of lane (c, g) in set j: row 4 g + e, column 64 grp + 4 c + j
            const int col0 = 64 * grp + 4 * c;
            if (col0 < ldw) {
                float qc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) qc[j] = qb[min(col0 + j, n - 1)];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float4 o;
                    float *op = reinterpret_cast<float *>(&o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t1 = acc[j][e] * -2.0f;           // inner * (-2)
                        const float t2 = t1 + qc[j];                  // + quadratic.unsqueeze(1)
                        op[j] = col0 + j < n ? t2 + qrow[e] : INFINITY;   // + quadratic.unsqueeze(2)
                    }
                    *reinterpret_cast<float4 *>(dblk16 + (4 * g + e) * ldw + col0) = o;
                }
            }
        }
    }
    if (has_extra && wave != 0) {
        // the tail row against every column: inner = x[0] y[0], then fmaf in ascending k -- what the MFMA chain computes.  Waves 1-3:
        // three columns per thread and pass, their chains interleaved (48 loads in flight)
        const float *xa = xt + (size_t)b * DIM * ldw;
        const float qx = qb[ix];
        const int t0 = (wave - 1) * 64 + lane;
        for (int cbase = t0; cbase < ldw; cbase += 3 * 192) {
            const int c0 = cbase, c1 = cbase + 192, c2 = cbase + 384;
            const int l0 = c0, l1 = min(c1, ldw - 1), l2 = min(c2, ldw - 1);
            const float x0 = xa[ix];
            float in0 = x0 * xa[l0], in1 = x0 * xa[l1], in2 = x0 * xa[l2];
#pragma unroll 16
            for (int kk = 1; kk < DIM; ++kk) {
                const float xv = xa[(size_t)kk * ldw + ix];
                in0 = fmaf(xv, xa[(size_t)kk * ldw + l0], in0);
                in1 = fmaf(xv, xa[(size_t)kk * ldw + l1], in1);
                in2 = fmaf(xv, xa[(size_t)kk * ldw + l2], in2);
            }
            auto put = [&](int col, float inner) {
                if (col < ldw) {
                    const float t1 = inner * -2.0f;
                    const float t2 = t1 + qb[min(col, n - 1)];
                    dblk16[KF16_ROWS * ldw + col] = col < n ? t2 + qx : INFINITY;
                }
            };
            put(c0, in0), put(c1, in1), put(c2, in2);
        }
    }
    __syncthreads();
    // ---- phase 2: four rows per wave (wave 3: the tail row as its fifth)
#pragma unroll 1
    for (int rr = 0; rr < 5; ++rr) {
        if (rr == 4 && !(has_extra && wave == 3)) break;              // wave-uniform
        const int lrow = rr == 4 ? KF16_ROWS : wave * 4 + rr, i = rr == 4 ? ix : i0 + lrow;
        if (i >= n) break;                                            // wave-uniform
        uint32_t key[NT];
        const float *row = dblk16 + lrow * ldw;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + (t << 6);
            key[t] = tgp_float_key(j < n ? row[j] : INFINITY);
        }
        int32_t *out = idx + ((size_t)b * n + i) * k;
        int slot, nb;
        if (!wave_select_ranked<NT>(key, lane, k, out, s_lmin[wave], s_list[wave], &slot, &nb)) {
            KNN_COUNT_SERIAL();
            wave_select_serial_keys<NT>(key, lane, k, out, &slot, &nb);
        }
        if (dirs) knn_emit_dirs(xyz, dirs, slot, nb, (int64_t)b * n + i, b, n, k);       // workgroup-uniform
    }
}

// (round 5) The same two phases as PRODUCER and CONSUMER waves of one workgroup: 512 threads, waves 0-3 compute the distances of row block
// j + 1 into one half of a double-buffered LDS image (2 x 17 rows x ldw: 144 KB at n = 1028) while waves 4-7 select block j from the
// other half; one barrier per block; a workgroup walks `bpw` consecutive row blocks of one object (B x 8 workgroups for the
// benchmark's B = 32, n = 1028: one per CU, ONE round instead of four).  Wave w and wave w + 4 share a SIMD, so every SIMD holds one
// wave on the matrix pipe and one on the vector / LDS pipes at all times -- in the 16-row form two workgroups per CU were meant to
// alternate like that but nothing kept them out of phase (SQ counters, round 4: matrix pipe busy 34 %).  Per element the same arithmetic
// in the same order as knn_feat_fused16_kernel (and the same selection): the same lists, bit for bit.
// MEASURED, NOT THE DEFAULT (scripts/knn_time.py, scripts/knn_pc_stamps.py, B = 32, n = 1028, d = 128): 255 us against 165-180 us for the
// 16-row form.  Stamps per wave: producers 26-32 k cycles per block inside their loop, consumers 36-43 k (9-10 k per row, with a selection of
// ~600 instructions), producers parked at the barrier for 40 % of the kernel.  The premise was wrong: the distance products are exact fp32
// MFMAs, which run at the fp32 VECTOR rate on the lanes the selection's vector instructions need, so a selecting wave beside a multiplying
// wave gets a fraction of the SIMD's issue slots (s_setprio 2 for the selectors: -4 %) -- the two phases do not overlap on one SIMD, they add
// (~17 k cycles of MFMA + ~16 k of selection per block and SIMD: the 16-row form's 163 us) -- and here only four waves select instead of
// eight.  What it did show: the selection, not the matrix pipe, is the longer half; its rewrite on 64-bit compares (wave_select_ranked)
// took 7 % off every form.
template <int DIM, int NT, int CH>
__global__ __launch_bounds__(512, 2) void knn_feat_pc_kernel(const float *__restrict__ xt, const float *__restrict__ q, int B, int n, int k,
                                                             int32_t *__restrict__ idx, int nrb, int ldw, int n_extra,
                                                             const float *__restrict__ xyz, float4 *__restrict__ dirs, int bpw, int chunks,
                                                             unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) float dpc[];       // [2][16 + 1][ldw]
    unsigned long long t_body = 0, t_bar = 0, t_keys = 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    __shared__ uint32_t s_lmin[4][64];
    __shared__ __attribute__((aligned(16))) uint2 s_list[4][KNN_LIST];
    constexpr int STEPS = DIM / 4;
    constexpr int NCHUNK = STEPS / CH;
    constexpr int RING = 4;
    static_assert(NCHUNK % RING == 0, "the chunk ring's phase must repeat per column group");
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3, consumer = wave8 >> 2;
    const int c = lane & 15, g = lane >> 4;
    int b, chunk;
    if (!tgp_xcd_object_tile(blockIdx.x, B, chunks, b, chunk)) return;
    const int rb0 = chunk * bpw, nb = min(nrb, rb0 + bpw) - rb0;
    // the selecting waves are the second-dispatched half of the workgroup: at equal priority they lose the arbitration for the SIMD's issue
    // slots against their producer partner on every instruction (micro-architecture guide, "Two waves per SIMD", item 4), and they are the
    // longer of the two chains
    if (consumer) __builtin_amdgcn_s_setprio(2);
    const size_t bufsz = (size_t)(KF16_ROWS + 1) * ldw;
    const float *qb = q + (size_t)b * n;
    const int ngrp = (ldw + 63) >> 6;
    const float *xo = xt + (size_t)b * DIM * ldw + (size_t)g * ldw;
#pragma unroll 1
    for (int it = 0; it <= nb; ++it) {
        const unsigned long long t0 = stamps ? __builtin_readcyclecounter() : 0ull;
        if (!consumer && it < nb) {
            // ---- phase 1 of block rb0 + it (knn_feat_fused16_kernel's, on the same four-wave split)
            const int rb = rb0 + it;
            const int i0 = rb * KF16_ROWS;
            float *dblk16 = dpc + (size_t)(it & 1) * bufsz;
            const bool has_extra = rb < n_extra;
            const int ix = nrb * KF16_ROWS + rb;
            {
                float a[STEPS];
#pragma unroll
                for (int s2 = 0; s2 < STEPS; ++s2) a[s2] = xo[(size_t)4 * s2 * ldw + i0 + c];
                float qrow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) qrow[e] = qb[min(i0 + 4 * g + e, n - 1)];
                float4 buf[RING][CH];
                auto fetch = [&](int grp, int cc, float4 (&dst)[CH]) {
                    const int col = min(64 * grp + 4 * c, ldw - 4);
                    const float *br = xo + (size_t)4 * cc * CH * ldw + col;
#pragma unroll
                    for (int t = 0; t < CH; ++t) dst[t] = *reinterpret_cast<const float4 *>(br + (size_t)4 * t * ldw);
                };
                if (wave < ngrp) {
#pragma unroll
                    for (int cc = 0; cc < RING - 1; ++cc) fetch(wave, cc, buf[cc]);
                }
                for (int grp = wave; grp < ngrp; grp += 4) {
                    knn_f32x4 acc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = knn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cc = 0; cc < NCHUNK; ++cc) {
                        constexpr int AHEAD = RING - 1;
                        if (cc + AHEAD < NCHUNK) fetch(grp, cc + AHEAD, buf[(cc + AHEAD) % RING]);
                        else if (grp + 4 < ngrp) fetch(grp + 4, cc + AHEAD - NCHUNK, buf[(cc + AHEAD) % RING]);
#pragma unroll
                        for (int t = 0; t < CH; ++t) {                // ascending k: step s = cc CH + t takes k = 4 s + g
                            const float av = a[cc * CH + t];
                            const float4 bv = buf[cc % RING][t];
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.z, acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.w, acc[3], 0, 0, 0);
                        }
                    }
                    const int col0 = 64 * grp + 4 * c;
                    if (col0 < ldw) {
                        float qc[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) qc[j] = qb[min(col0 + j, n - 1)];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float4 o;
                            float *op = reinterpret_cast<float *>(&o);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float t1 = acc[j][e] * -2.0f;           // inner * (-2)
                                const float t2 = t1 + qc[j];                  // + quadratic.unsqueeze(1)
                                op[j] = col0 + j < n ? t2 + qrow[e] : INFINITY;   // + quadratic.unsqueeze(2)
                            }
                            *reinterpret_cast<float4 *>(dblk16 + (4 * g + e) * ldw + col0) = o;
                        }
                    }
                }
            }
            if (has_extra && wave != 0) {
                const float *xa = xt + (size_t)b * DIM * ldw;
                const float qx = qb[ix];
                const int t0 = (wave - 1) * 64 + lane;
                for (int cbase = t0; cbase < ldw; cbase += 3 * 192) {
                    const int c0 = cbase, c1 = cbase + 192, c2 = cbase + 384;
                    const int l0 = c0, l1 = min(c1, ldw - 1), l2 = min(c2, ldw - 1);
                    const float x0 = xa[ix];
                    float in0 = x0 * xa[l0], in1 = x0 * xa[l1], in2 = x0 * xa[l2];
#pragma unroll 16
                    for (int kk = 1; kk < DIM; ++kk) {
                        const float xv = xa[(size_t)kk * ldw + ix];
                        in0 = fmaf(xv, xa[(size_t)kk * ldw + l0], in0);
                        in1 = fmaf(xv, xa[(size_t)kk * ldw + l1], in1);
                        in2 = fmaf(xv, xa[(size_t)kk * ldw + l2], in2);
                    }
                    auto put = [&](int col, float inner) {
                        if (col < ldw) {
                            const float t1 = inner * -2.0f;
                            const float t2 = t1 + qb[min(col, n - 1)];
                            dblk16[KF16_ROWS * ldw + col] = col < n ? t2 + qx : INFINITY;
                        }
                    };
                    put(c0, in0), put(c1, in1), put(c2, in2);
                }
            }
        } else if (consumer && it >= 1) {
            // ---- phase 2 of block rb0 + it - 1: four rows per wave (wave 3: the tail row as its fifth)
            const int rb = rb0 + it - 1;
            const int i0 = rb * KF16_ROWS;
            const float *dblk16 = dpc + (size_t)((it - 1) & 1) * bufsz;
            const bool has_extra = rb < n_extra;
            const int ix = nrb * KF16_ROWS + rb;
#pragma unroll 1
            for (int rr = 0; rr < 5; ++rr) {
                if (rr == 4 && !(has_extra && wave == 3)) break;
                const int lrow = rr == 4 ? KF16_ROWS : wave * 4 + rr, i = rr == 4 ? ix : i0 + lrow;
                if (i >= n) break;
                uint32_t key[NT];
                const float *row = dblk16 + lrow * ldw;
                const unsigned long long tk0 = stamps ? __builtin_readcyclecounter() : 0ull;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = lane + (t << 6);
                    key[t] = tgp_float_key(j < n ? row[j] : INFINITY);
                }
                int32_t *out = idx + ((size_t)b * n + i) * k;
                int slot, nb;
                if (stamps) t_keys += __builtin_readcyclecounter() - tk0;
                if (!wave_select_ranked<NT>(key, lane, k, out, s_lmin[wave], s_list[wave], &slot, &nb, stamps ? ph : nullptr)) {
                    KNN_COUNT_SERIAL();
                    wave_select_serial_keys<NT>(key, lane, k, out, &slot, &nb);
                }
                if (dirs) knn_emit_dirs(xyz, dirs, slot, nb, (int64_t)b * n + i, b, n, k);
            }
        }
        const unsigned long long t1 = stamps ? __builtin_readcyclecounter() : 0ull;
        __syncthreads();
        if (stamps) t_body += t1 - t0, t_bar += __builtin_readcyclecounter() - t1;
    }
    if (stamps && lane == 0) {
        stamps[(size_t)blockIdx.x * 16 + wave8 * 2] = t_body, stamps[(size_t)blockIdx.x * 16 + wave8 * 2 + 1] = t_bar;
        if (wave8 == 5) {          // one consumer wave's phases: keys, lane minima + bound, count + prefix, compaction, ranks
            unsigned long long *o = stamps + (size_t)gridDim.x * 16 + (size_t)blockIdx.x * 8;
            o[0] = t_keys, o[1] = ph[0], o[2] = ph[1], o[3] = ph[2], o[4] = ph[3];
        }
    }
}

#ifdef TGP_DEV
static unsigned long long *tgp_knn_stamps = nullptr;
extern "C" void tgp_debug_set_knn_stamps(unsigned long long *buf) { tgp_knn_stamps = buf; }
static int tgp_knn_pc_mode = 0;            // 1: the library picks the producer / consumer form where a workgroup gets >= 4 row blocks (A/B runs)
extern "C" void tgp_debug_set_knn_pc(int v) { tgp_knn_pc_mode = v; }
#else
static constexpr unsigned long long *tgp_knn_stamps = nullptr;
static constexpr int tgp_knn_pc_mode = 0;  // measured SLOWER (below): form 3 stays a test / measurement handle
#endif

// implemented in gemm.hip: D[b,i,j] = fl(fl(-2*<x_i,x_j> + q_j) + q_i), natural-k MFMA chain
int tgp_launch_dist_gemm(const float *x, int ld, const float *q, int B, int n, int d, float *D, hipStream_t stream);

// xt[b][c][j] = x[b][j][c] for j < n, 0 for n <= j < ldt (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void knn_transpose_kernel(const float *__restrict__ x, int ld, int n, int d, float *__restrict__ xt, int ldt)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int rr = ty; rr < 32; rr += 8) tile[rr][tx] = (j0 + rr < n) ? x[((size_t)b * n + j0 + rr) * ld + c0 + tx] : 0.f;
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) xt[((size_t)b * d + c0 + cc) * ldt + j0 + tx] = tile[tx][cc];
}

// Both pre-passes of the fused kernel in ONE launch (round 3; they were two: 4 feature-space graphs x 2 launches per forward): a
// workgroup takes 32 rows of an object and walks their d / 32 column tiles; each tile goes through LDS once and leaves as a
// transposed block of xt AND as the next four terms of the rows' squared norms, accumulated in sqnorm_aten_kernel's order (thread
// (row, l8) holds ATen's four interleaved accumulators over the columns 8 kk + l8 of every 32-column group, groups ascending;
// then acc0 + acc1 + acc2 + acc3, then the eight lanes in order): bit-identical norms.
__global__ __launch_bounds__(256) void knn_prep_kernel(const float *__restrict__ x, int ld, int n, int d, float *__restrict__ xt, int ldt,
                                                       float *__restrict__ q)
{
    __shared__ float tile[4][32][33];          // four 32-column groups per pass (d = 128: one pass)
    const int b = blockIdx.y, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d; c0 += 128) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            for (int rr = ty; rr < 32; rr += 8)
                tile[g][rr][tx] = (j0 + rr < n && c0 + g * 32 < d) ? x[((size_t)b * n + j0 + rr) * ld + c0 + g * 32 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (c0 + g * 32 >= d) break;
            for (int cc = ty; cc < 32; cc += 8) xt[((size_t)b * d + c0 + g * 32 + cc) * ldt + j0 + tx] = tile[g][tx][cc];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {       // groups ascending, then ATen's four interleaved accumulators
                const float v = tile[g][r][kk * 8 + l8];
                acc[kk] = acc[kk] + v * v;
            }
        }
        __syncthreads();
    }
    float p = acc[0] + acc[1];
    p = p + acc[2];
    p = p + acc[3];
    float fin = 0.f;
    const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin = fin + __shfl(p, base + l, 64);
    if (l8 == 0 && j0 + r < n) q[(size_t)b * n + j0 + r] = 0.f + fin;
}

// The same with 16-byte global accesses (round 4): a quarter of the load / store instructions -- thread (row r, l8) loads the
// float4 at columns 4 (l8 + 8 i) of its row (eight lanes = 128 contiguous bytes), the transposed block leaves as float4 along the
// points.  The LDS tile and the norm's arithmetic are those of the kernel above (same values, same order: bit-identical norms).
// Needs x 16-byte addressable (base and ld): the launcher falls back to the scalar kernel otherwise.
__global__ __launch_bounds__(256) void knn_prep_v4_kernel(const float *__restrict__ x, int ld, int n, int d, float *__restrict__ xt, int ldt,
                                                          float *__restrict__ q)
{
    __shared__ float tile[4][32][33];
    const int b = blockIdx.y, j0 = blockIdx.x * 32;
    const int r = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d; c0 += 128) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 + r < n && c0 + g * 32 < d) v = *reinterpret_cast<const float4 *>(x + ((size_t)b * n + j0 + r) * ld + c0 + g * 32 + 4 * l8);
            tile[g][r][4 * l8 + 0] = v.x, tile[g][r][4 * l8 + 1] = v.y, tile[g][r][4 * l8 + 2] = v.z, tile[g][r][4 * l8 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (c0 + g * 32 >= d) break;
            // transposed block: thread (column r of the group, points 4 l8 .. 4 l8 + 3)
            const float4 t4 = make_float4(tile[g][4 * l8 + 0][r], tile[g][4 * l8 + 1][r], tile[g][4 * l8 + 2][r], tile[g][4 * l8 + 3][r]);
            *reinterpret_cast<float4 *>(xt + ((size_t)b * d + c0 + g * 32 + r) * ldt + j0 + 4 * l8) = t4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {       // groups ascending, then ATen's four interleaved accumulators
                const float v = tile[g][r][kk * 8 + l8];
                acc[kk] = acc[kk] + v * v;
            }
        }
        __syncthreads();
    }
    float p = acc[0] + acc[1];
    p = p + acc[2];
    p = p + acc[3];
    float fin = 0.f;
    const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin = fin + __shfl(p, base + l, 64);
    if (l8 == 0 && j0 + r < n) q[(size_t)b * n + j0 + r] = 0.f + fin;
}

// does the fused (matrix-free) kernel serve this shape?
static bool knn_feat_fused_ok(int n, int d)
{
    return (d == 128 || d == 256) && tgp_cdiv(n, 32) * 32 <= KF_MAX_LDW;
}

extern "C" int64_t tgp_knn_feat_workspace_bytes(int B, int n, int d)
{
    if (B <= 0 || n <= 0) return 0;
    // the squared norms always; beside them the transposed features (fused kernel) or the (B, n, n) distance matrix (the shapes the
    // fused kernel does not serve)
    const int64_t extra = knn_feat_fused_ok(n, d) ? (int64_t)B * d * (tgp_cdiv(n, 32) * 32) : (int64_t)B * n * n;
    return (extra + (int64_t)B * n) * (int64_t)sizeof(float);
}

// form: 0 = the library's choice, 1 = 32-row blocks (one workgroup per CU), 2 = 16-row blocks (two per CU)
template <int DIM, int NT, int CH>
static int launch_knn_fused(const float *xt, const float *q, int B, int n, int k, int32_t *idx, hipStream_t stream, int form,
                            const float *xyz = nullptr, float4 *dirs = nullptr)
{
    const int ncb = tgp_cdiv(n, 32), ldw = ncb * 32;
    if (form != 1) {
        const int tail16 = n % KF16_ROWS, n_extra16 = (tail16 > 0 && tail16 <= 8 && tail16 <= n / KF16_ROWS) ? tail16 : 0;
        const int nrb16 = n_extra16 ? n / KF16_ROWS : tgp_cdiv(n, KF16_ROWS);
        const size_t lds16 = (size_t)(KF16_ROWS + (n_extra16 ? 1 : 0)) * ldw * sizeof(float);
        // (round 5) producer / consumer waves over a double-buffered image: form 3, and the library's choice when a workgroup gets at
        // least four row blocks of an object to walk (one workgroup per CU: B x chunks of them) and the image fits
        {
            static int cus = 0;
            if (!cus) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                    cus = 256;
            }
            const size_t ldspc = (size_t)2 * (KF16_ROWS + 1) * ldw * sizeof(float);
            const int want = B >= cus ? 1 : cus / B;                   // workgroups per object for about one per CU
            const int bpw = tgp_cdiv(nrb16, want < nrb16 ? want : nrb16), chunks = tgp_cdiv(nrb16, bpw);
            if (ldspc <= 152 * 1024 && (form == 3 || (form == 0 && bpw >= 4 && tgp_knn_pc_mode))) {
                static TgpLdsAttr attrpc;
                if (const int e = tgp_lds_attr(attrpc, reinterpret_cast<const void *>(knn_feat_pc_kernel<DIM, NT, CH / 2>), 152 * 1024)) return e;
                hipLaunchKernelGGL((knn_feat_pc_kernel<DIM, NT, CH / 2>), dim3(tgp_xcd_grid(B, chunks)), dim3(512), ldspc, stream, xt, q, B, n, k,
                                   idx, nrb16, ldw, n_extra16, xyz, dirs, bpw, chunks, tgp_knn_stamps);
                return TGP_LAUNCH_RESULT();
            }
        }
        static TgpLdsAttr attr16;
        if (const int e = tgp_lds_attr(attr16, reinterpret_cast<const void *>(knn_feat_fused16_kernel<DIM, NT, CH / 2>),
                                       (KF16_ROWS + 1) * KF_MAX_LDW * (int)sizeof(float))) return e;
        // (four steps per prefetched chunk, three chunks ahead; eight measured the same: 18.68 vs 18.67 k objects/s)
        hipLaunchKernelGGL((knn_feat_fused16_kernel<DIM, NT, CH / 2>), dim3(tgp_xcd_grid(B, nrb16)), dim3(256), lds16, stream, xt, q, B, n, k,
                           idx, nrb16, ldw, n_extra16, xyz, dirs);
        return TGP_LAUNCH_RESULT();
    }
    if (dirs) return TGP_EUNSUPPORTED;
    // a tail of at most 8 rows (and fewer than there are full blocks) rides along with the first blocks instead of forming its own
    const int tail = n % KF_ROWS, n_extra = (tail > 0 && tail <= 8 && tail <= n / KF_ROWS) ? tail : 0;
    const int nrb = n_extra ? n / KF_ROWS : tgp_cdiv(n, KF_ROWS);
    const size_t lds = (size_t)(KF_ROWS + (n_extra ? 1 : 0)) * ldw * sizeof(float);
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(knn_feat_fused_kernel<DIM, NT, CH>),
                                   (KF_ROWS + 1) * KF_MAX_LDW * (int)sizeof(float))) return e;
    hipLaunchKernelGGL((knn_feat_fused_kernel<DIM, NT, CH>), dim3(tgp_xcd_grid(B, nrb)), dim3(512), lds, stream, xt, q, B, n, k, idx, nrb, ncb,
                       ldw, n_extra, tgp_knn_stamps);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_knn_feat_form(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                                 int64_t workspace_bytes, int form, tgp_stream_t stream);
extern "C" int tgp_knn_feat(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                            int64_t workspace_bytes, tgp_stream_t stream)
{
    return tgp_knn_feat_form(feat, ld, B, n, d, k, idx, workspace, workspace_bytes, 0, stream);
}

static int knn_feat_go(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace, int64_t workspace_bytes,
                       int form, const float *xyz, float4 *dirs, tgp_stream_t stream);

extern "C" int tgp_knn_feat_form(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                                 int64_t workspace_bytes, int form, tgp_stream_t stream)
{
    return knn_feat_go(feat, ld, B, n, d, k, idx, workspace, workspace_bytes, form, nullptr, nullptr, stream);
}

// tgp_knn_feat that also leaves the unit directions to the selected neighbours in the points' xyz (B, n, 3): dirs (B, n, k) float4
// (x, y, z, 0), what tgp_gconv_hs_fwd_dirs takes.  Only the shapes the fused 16-row kernel serves: TGP_EUNSUPPORTED otherwise
// (nothing launched).
extern "C" int tgp_knn_feat_dirs(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                                 int64_t workspace_bytes, const float *xyz, float *dirs, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && dirs && (reinterpret_cast<uintptr_t>(dirs) & 15) == 0);
    if (!knn_feat_fused_ok(n, d)) return TGP_EUNSUPPORTED;
    return knn_feat_go(feat, ld, B, n, d, k, idx, workspace, workspace_bytes, 0, xyz, reinterpret_cast<float4 *>(dirs), stream);
}

static int knn_feat_go(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace, int64_t workspace_bytes,
                       int form, const float *xyz, float4 *dirs, tgp_stream_t stream)
{
    TGP_REQUIRE(feat && idx && workspace && form >= 0 && form <= 3);
    const int chk = knn_check(B, n, k);
    if (chk) return chk;
    if (d <= 0 || (d & 31) || (d >> 5) >= 16) return TGP_EUNSUPPORTED;
    TGP_REQUIRE(ld >= d && (ld & 3) == 0);
    TGP_REQUIRE(workspace_bytes >= tgp_knn_feat_workspace_bytes(B, n, d));
    const bool fused = knn_feat_fused_ok(n, d);
    const int ldt = tgp_cdiv(n, 32) * 32;
    float *D = reinterpret_cast<float *>(workspace);   // the transposed features or the (B,n,n) distance matrix, then (B,n) squared norms
    float *q = D + (fused ? (size_t)B * d * ldt : (size_t)B * n * n);
    const int64_t rows = (int64_t)B * n;
    const int nt = tgp_cdiv(n, 64);
    if (fused) {
        if ((reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(D) & 15) == 0)
            hipLaunchKernelGGL(knn_prep_v4_kernel, dim3(ldt / 32, B), dim3(256), 0, tgp_hs(stream), feat, ld, n, d, D, ldt, q);
        else
            hipLaunchKernelGGL(knn_prep_kernel, dim3(ldt / 32, B), dim3(256), 0, tgp_hs(stream), feat, ld, n, d, D, ldt, q);
#define LAUNCH_FUSED(NT) \
    (d == 128 ? launch_knn_fused<128, NT, 8>(D, q, B, n, k, idx, tgp_hs(stream), form, xyz, dirs)                     \
              : launch_knn_fused<256, NT, 8>(D, q, B, n, k, idx, tgp_hs(stream), form, xyz, dirs))
        if (nt <= 1) return LAUNCH_FUSED(1);
        if (nt <= 2) return LAUNCH_FUSED(2);
        if (nt <= 5) return LAUNCH_FUSED(5);
        if (nt <= 8) return LAUNCH_FUSED(8);
        if (nt <= 17) return LAUNCH_FUSED(17);
        return LAUNCH_FUSED(19);
#undef LAUNCH_FUSED
    }
    hipLaunchKernelGGL(sqnorm_aten_kernel, dim3(tgp_cdiv(rows * 8, 256)), dim3(256), 0, tgp_hs(stream), feat, ld, rows,
                       d, q);
    int rc = tgp_launch_dist_gemm(feat, ld, q, B, n, d, D, tgp_hs(stream));
    if (rc) return rc;
    const int rpb = knn_rows_per_block(B, n);
    const int tiles = tgp_cdiv(n, rpb);
    const dim3 grid(tgp_xcd_grid(B, tiles)), block(256);
#define LAUNCH_MAT(NT) hipLaunchKernelGGL(knn_matrix_kernel<NT>, grid, block, 0, tgp_hs(stream), D, B, n, k, idx, tiles, rpb)
    if (nt <= 1) LAUNCH_MAT(1);
    else if (nt <= 2) LAUNCH_MAT(2);
    else if (nt <= 5) LAUNCH_MAT(5);
    else if (nt <= 8) LAUNCH_MAT(8);
    else if (nt <= 17) LAUNCH_MAT(17);
    else LAUNCH_MAT(32);
#undef LAUNCH_MAT
    return TGP_LAUNCH_RESULT();
}

// ------------------------------------------------------------------------------------------------
// get_nearest_index: one thread per target point, source cloud staged through LDS in chunks.
#define NN1_CHUNK 1024
// one source cloud against the workgroup's 256 targets: returns the lane's nearest source point (first index on ties)
__device__ __forceinline__ int nn1_scan(const float *__restrict__ src, const int m, const float tx, const float ty, const float tz,
                                        const float qt, float4 *s)
{
    float best = 0.f;
    int besti = 0;
    for (int j0 = 0; j0 < m; j0 += NN1_CHUNK) {
        const int cnt = min(NN1_CHUNK, m - j0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
            const float *p = src + (size_t)(j0 + j) * 3;
            const float x = p[0], y = p[1], z = p[2];
            float q = x * x;
            q = q + y * y;
            q = q + z * z;
            s[j] = make_float4(x, y, z, q);
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 p = s[j];
            float inner = tx * p.x;
            inner = fmaf(ty, p.y, inner);
            inner = fmaf(tz, p.z, inner);
            const float sum = p.w + qt;               // s_norm_2.unsqueeze(1) + t_norm_2.unsqueeze(2)
            const float dv = sum - 2.0f * inner;      // - 2 * inner
            if ((j0 + j) == 0 || dv < best) {
                best = dv;
                besti = j0 + j;
            }
        }
    }
    return besti;
}

// src2 != NULL (round 4, tgp_nn1_pair): the same targets against a second cloud in the same launch (the two up-sampling look-ups
// of Face_Enc.forward, FaceRecon.py:71-77)
__global__ __launch_bounds__(256) void nn1_kernel(const float *__restrict__ tgt, const float *__restrict__ src, int B,
                                                  int n, int m, int32_t *__restrict__ idx, int tiles_per_obj,
                                                  const float *__restrict__ src2, int m2, int32_t *__restrict__ idx2,
                                                  const float *__restrict__ obj_id, int n_cls, float *__restrict__ feat, int ld,
                                                  int col0)
{
    __shared__ float4 s[NN1_CHUNK];
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int i = tile * 256 + threadIdx.x;
    const bool live = i < n;
    float tx = 0.f, ty = 0.f, tz = 0.f, qt = 0.f;
    if (live) {
        const float *t = tgt + ((size_t)b * n + i) * 3;
        tx = t[0], ty = t[1], tz = t[2];
        qt = tx * tx;
        qt = qt + ty * ty;
        qt = qt + tz * tz;
    }
    if (feat && live) {
        // the point's row of the concat buffer behind the feature maps: one-hot category | x y z | zero padding (tgp_fill_tail;
        // FaceRecon.py:54,78-79) -- this kernel already has one thread per point and its coordinates
        const int cls = (int)obj_id[b];
        float *f = feat + ((size_t)b * n + i) * ld + col0;
        for (int c = 0; c < n_cls; ++c) f[c] = (c == cls) ? 1.f : 0.f;
        f[n_cls + 0] = tx, f[n_cls + 1] = ty, f[n_cls + 2] = tz;
        for (int c = n_cls + 3; c < ld - col0; ++c) f[c] = 0.f;
    }
    const int b1 = nn1_scan(src + (size_t)b * m * 3, m, tx, ty, tz, qt, s);
    if (live) idx[(size_t)b * n + i] = b1;
    if (!src2) return;
    const int b2 = nn1_scan(src2 + (size_t)b * m2 * 3, m2, tx, ty, tz, qt, s);
    if (live) idx2[(size_t)b * n + i] = b2;
}

extern "C" int tgp_nn1(const float *target, const float *source, int B, int n, int m, int32_t *idx, tgp_stream_t stream)
{
    TGP_REQUIRE(target && source && idx && B > 0 && n > 0 && m > 0);
    const int tiles = tgp_cdiv(n, 256);
    hipLaunchKernelGGL(nn1_kernel, dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, tgp_hs(stream), target, source, B, n, m,
                       idx, tiles, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, 0);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_nn1_pair_tail(const float *target, const float *source1, const float *source2, int B, int n, int m1, int m2,
                                 int32_t *idx1, int32_t *idx2, const float *obj_id, int n_cls, float *feat, int ld, int col0,
                                 tgp_stream_t stream);
extern "C" int tgp_nn1_pair(const float *target, const float *source1, const float *source2, int B, int n, int m1, int m2,
                            int32_t *idx1, int32_t *idx2, tgp_stream_t stream)
{
    return tgp_nn1_pair_tail(target, source1, source2, B, n, m1, m2, idx1, idx2, nullptr, 0, nullptr, 0, 0, stream);
}

// tgp_nn1_pair that also writes the targets' tail columns of the concat buffer (tgp_fill_tail's: target = the centred cloud)
extern "C" int tgp_nn1_pair_tail(const float *target, const float *source1, const float *source2, int B, int n, int m1, int m2,
                                 int32_t *idx1, int32_t *idx2, const float *obj_id, int n_cls, float *feat, int ld, int col0,
                                 tgp_stream_t stream)
{
    TGP_REQUIRE(target && source1 && source2 && idx1 && idx2 && B > 0 && n > 0 && m1 > 0 && m2 > 0);
    TGP_REQUIRE(!feat || (obj_id && n_cls > 0 && col0 >= 0 && ld >= col0 + n_cls + 3));
    const int tiles = tgp_cdiv(n, 256);
    hipLaunchKernelGGL(nn1_kernel, dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, tgp_hs(stream), target, source1, B, n, m1,
                       idx1, tiles, source2, m2, idx2, obj_id, n_cls, feat, ld, col0);
    return TGP_LAUNCH_RESULT();
}
