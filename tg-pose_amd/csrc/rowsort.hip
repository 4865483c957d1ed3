// Row order of the factored wide layers (engine.encoder_forward(factored=True)): an object's fine rows sorted by their
// nearest coarse points, key = near2 * n1 + near1, ties in point order -- what
//     order = torch.argsort(near2.long() * N1 + near1.long(), dim=1, stable=True)
// plus two gathers, two offset additions and the dtype conversions did in a dozen ATen launches (radix sort 28 us + ~12 x 5 us
// per forward).  One workgroup per object: the composite key (key << 11 | point) is unique, so a bitonic network in LDS is a
// stable sort; nothing here touches HBM beyond the 8 KB in and 24 KB out per object.
#include "tgp_common.h"

#define SORT_THREADS 1024
#define SORT_MAX 2048

__global__ __launch_bounds__(SORT_THREADS) void sort_by_parent_kernel(const int32_t *__restrict__ near1, const int32_t *__restrict__ near2,
                                                                      int n, int n1, int n2, int P, int32_t *__restrict__ order,
                                                                      int64_t *__restrict__ order64, int32_t *__restrict__ near1_out,
                                                                      int32_t *__restrict__ near2_out)
{
    __shared__ uint32_t keys[SORT_MAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < P; t += SORT_THREADS) {
        uint32_t k = 0xffffffffu;                                   // padding sorts to the end
        if (t < n) k = ((uint32_t)(near2[(size_t)b * n + t] * n1 + near1[(size_t)b * n + t]) << 11) | (uint32_t)t;
        keys[t] = k;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += SORT_THREADS) {
                // pair (lo, lo + j): lo skips the bit j
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo + j;
                const bool up = (lo & k) == 0;
                const uint32_t a = keys[lo], c = keys[hi];
                if ((a > c) == up) keys[lo] = c, keys[hi] = a;
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < n; t += SORT_THREADS) {
        const uint32_t e = keys[t];
        const int idx = (int)(e & 2047u), key = (int)(e >> 11);
        const size_t o = (size_t)b * n + t;
        order[o] = idx;
        order64[o] = idx;
        near1_out[o] = key % n1 + b * n1;
        near2_out[o] = key / n1 + b * n2;
    }
}

extern "C" int tgp_sort_by_parent(const int32_t *near1, const int32_t *near2, int B, int n, int n1, int n2, int32_t *order, int64_t *order64,
                                  int32_t *near1_out, int32_t *near2_out, tgp_stream_t stream)
{
    TGP_REQUIRE(near1 && near2 && order && order64 && near1_out && near2_out && B > 0 && n > 0 && n1 > 0 && n2 > 0);
    if (n > SORT_MAX || (int64_t)n1 * n2 > (1 << 21)) return TGP_EUNSUPPORTED;        // 11 bits of point, 21 bits of key
    int P = 2;
    while (P < n) P <<= 1;
    hipLaunchKernelGGL(sort_by_parent_kernel, dim3(B), dim3(SORT_THREADS), 0, tgp_hs(stream), near1, near2, n, n1, n2, P, order, order64,
                       near1_out, near2_out);
    return TGP_LAUNCH_RESULT();
}
