// Backward of "fetch the row of my nearest coarse point" (the factored wide layers' gathered residuals, csrc/gemm.hip gemm_epilogue_lds;
// FaceRecon.py:70-75 nearest up-sampling folded into the layers that read the concat buffer): d P[r] = sum over the points i whose
// nearest coarse point is r of d Y[i].  A scatter-add with float atomics would issue one atomic per element of d Y (M x 4608 of them
// per step) in an order that changes from run to run; here the map point -> parent is inverted once per forward into child lists
// (counting sort, children in point order) and each coarse row then SUMS its children's rows: coalesced 16-byte reads, every row
// of d Y read once, fixed order, no atomics.
#include "tgp_common.h"

#define CL_MAX_PARENTS 4096

#define CL_SPLIT 16  // workgroups per object (each repeats the counting sort's histogram and ranks a sixteenth of the points)

// ptr[b * R + r] = first slot of parent r's children in idx (global), idx[...] = global rows (b * n + i): a counting sort of the
// object's points by parent.  Slot of point i = (points with a smaller parent) + (earlier points with the same parent); the second
// term is counted by scanning the object's parent ids in LDS (every lane of a wave reads the same 16 bytes: a broadcast).
__global__ __launch_bounds__(256) void child_lists_kernel(const int32_t *__restrict__ near, int B, int n, int R, int gbase,
                                                          int32_t *__restrict__ ptr, int32_t *__restrict__ idx)
{
    __shared__ int s_cnt[CL_MAX_PARENTS];
    __shared__ int s_part[256];
    extern __shared__ int4 s_near4[];                     // the object's parent ids, padded to a multiple of 4 with -1
    int *s_near = reinterpret_cast<int *>(s_near4);
    const int b = blockIdx.x;
    const int32_t *nb = near + (size_t)b * n;
    for (int r = threadIdx.x; r < R; r += 256) s_cnt[r] = 0;
    __syncthreads();
    const int n4 = (n + 3) & ~3;
    for (int i = threadIdx.x; i < n4; i += 256) {
        int p = -1;
        if (i < n) {
            p = nb[i] - b * gbase;                        // global ids carry the object's offset b * R
            p = p < 0 ? 0 : (p >= R ? R - 1 : p);         // (a parent id outside [0, R) would be the caller's bug: clamped, never a fault)
            atomicAdd(&s_cnt[p], 1);                      // integer counts: order-free
        }
        s_near[i] = p;
    }
    __syncthreads();
    // exclusive scan of the counts: each thread owns `per` consecutive parents
    const int per = (R + 255) / 256;
    const int r0 = threadIdx.x * per;
    int local = 0;
    for (int r = r0; r < r0 + per && r < R; ++r) local += s_cnt[r];
    s_part[threadIdx.x] = local;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - local;
    for (int r = r0; r < r0 + per && r < R; ++r) {
        const int c = s_cnt[r];
        s_cnt[r] = run;
        run += c;
    }
    __syncthreads();
    if (blockIdx.y == 0) {
        for (int r = threadIdx.x; r < R; r += 256) ptr[(size_t)b * R + r] = b * n + s_cnt[r];
        if (b == B - 1 && threadIdx.x == 0) ptr[(size_t)B * R] = B * n;
    }
    const int chunk = (n + CL_SPLIT - 1) / CL_SPLIT;
    const int q0 = blockIdx.y * chunk, q1 = q0 + chunk < n ? q0 + chunk : n;
    for (int base = q0; base < q1; base += 256) {
        const int i = base + threadIdx.x;
        const int p = i < q1 ? s_near[i] : -2;
        const int top = base + 256 < q1 ? base + 256 : q1;      // uniform bound: the last point this pass ranks
        int rank = 0;
        for (int j4 = 0; j4 * 4 < top; ++j4) {
            const int4 v = s_near4[j4];
            const int j = j4 * 4;
            rank += (v.x == p && j < i) + (v.y == p && j + 1 < i) + (v.z == p && j + 2 < i) + (v.w == p && j + 3 < i);
        }
        if (i < q1) idx[(size_t)b * n + s_cnt[p] + rank] = b * n + i;    // children in point order: deterministic sums downstream
    }
}

extern "C" int tgp_child_lists(const int32_t *near, int B, int n, int R, int global_ids, int32_t *ptr, int32_t *idx, tgp_stream_t stream)
{
    TGP_REQUIRE(near && ptr && idx && B > 0 && n > 0 && R > 0);
    if (R > CL_MAX_PARENTS || n > 8192 || (int64_t)B * n >= 0x7fffffff) return TGP_EUNSUPPORTED;
    hipLaunchKernelGGL(child_lists_kernel, dim3(B, CL_SPLIT), dim3(256), (size_t)((n + 3) & ~3) * sizeof(int), tgp_hs(stream), near, B, n, R,
                       global_ids ? R : 0, ptr, idx);
    return TGP_LAUNCH_RESULT();
}

// out[r][c] = sum over k in [ptr[r], ptr[r + 1]) of g[idx[k]][c], children in list order; a workgroup per (parent, 256 x V columns).
// V = 4: 16-byte accesses; V = 2: 8-byte accesses, for rows that are only 8-byte aligned (a column slice of the 1286-wide gradient of
// the concat buffer: row stride 5144 bytes)
template <int V>
__global__ __launch_bounds__(256) void segsum_rows_kernel(const float *__restrict__ g, int ldg, int C, const int32_t *__restrict__ ptr,
                                                          const int32_t *__restrict__ idx, float *__restrict__ out, int ldo)
{
    const int r = blockIdx.x;
    const int c = blockIdx.y * 256 * V + threadIdx.x * V;
    if (c >= C) return;
    const int k0 = ptr[r], k1 = ptr[r + 1];
    float acc[V];
#pragma unroll
    for (int u = 0; u < V; ++u) acc[u] = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float *src = g + (int64_t)idx[k] * ldg + c;
        if (V == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src);
            acc[0] += v.x, acc[1] += v.y, acc[2] += v.z, acc[3] += v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(src);
            acc[0] += v.x, acc[1] += v.y;
        }
    }
    float *dst = out + (int64_t)r * ldo + c;
    if (V == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else *reinterpret_cast<float2 *>(dst) = make_float2(acc[0], acc[1]);
}

extern "C" int tgp_segsum_rows(const float *g, int ldg, int C, const int32_t *ptr, const int32_t *idx, int R, float *out, int ldo,
                               tgp_stream_t stream)
{
    TGP_REQUIRE(g && ptr && idx && out && C > 0 && R > 0 && ldg >= C && ldo >= C);
    const uintptr_t al = reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out);
    if ((C & 3) == 0 && (ldg & 3) == 0 && (ldo & 3) == 0 && (al & 15) == 0)
        hipLaunchKernelGGL(segsum_rows_kernel<4>, dim3(R, tgp_cdiv(C, 1024)), dim3(256), 0, tgp_hs(stream), g, ldg, C, ptr, idx, out, ldo);
    else {
        TGP_REQUIRE((C & 1) == 0 && (ldg & 1) == 0 && (ldo & 1) == 0 && (al & 7) == 0);
        hipLaunchKernelGGL(segsum_rows_kernel<2>, dim3(R, tgp_cdiv(C, 512)), dim3(256), 0, tgp_hs(stream), g, ldg, C, ptr, idx, out, ldo);
    }
    return TGP_LAUNCH_RESULT();
}
