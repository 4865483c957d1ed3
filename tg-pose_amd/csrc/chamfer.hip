// Chamfer distance (losses/chamfer3D: chamfer3D.cu NmDistanceKernel / NmDistanceGradKernel) on gfx950.
//
// Forward: tiled all-pairs search, both directions in ONE launch.  A workgroup owns 256 query
// points of one cloud of one batch element; the other cloud is streamed through LDS in 1024-point
// tiles as float4 (ds_read_b128 broadcast: all lanes read the same candidate), each lane keeps its
// running (best, index).  The squared distance is (dx*dx + dy*dy) + dz*dz with separate fp32
// roundings and a strict '<' so the lowest index wins ties -- exactly the CPU path
// (tools/pyTorchChamferDistance/chamfer_distance.cpp:59-87), hence bit-identical results.
//
// Backward: deterministic replacement for the reference's atomicAdd scatter (chamfer3D.cu:166-171):
// a thread owns one point's gradient and scans the other cloud's index list for hits, adding
// terms in the serial order of the reference's CPU loop (chamfer_distance.cpp:140-175).
#include "tgp_common.h"

#define CH_TILE 1024

__global__ __launch_bounds__(256) void chamfer_fwd_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                          int B, int n, int m, float *__restrict__ dist1,
                                                          float *__restrict__ dist2, int32_t *__restrict__ idx1,
                                                          int32_t *__restrict__ idx2, int tiles1, int tiles_per_obj)
{
    __shared__ float4 cand[CH_TILE];
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    // tiles [0, tiles1): queries from xyz1 against xyz2; the rest: queries from xyz2 against xyz1
    const bool fwd = tile < tiles1;
    const float *qs = fwd ? xyz1 + (size_t)b * n * 3 : xyz2 + (size_t)b * m * 3;
    const float *cs = fwd ? xyz2 + (size_t)b * m * 3 : xyz1 + (size_t)b * n * 3;
    const int nq = fwd ? n : m, nc = fwd ? m : n;
    const int i = (fwd ? tile : tile - tiles1) * 256 + threadIdx.x;
    const bool live = i < nq;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (live) x1 = qs[i * 3 + 0], y1 = qs[i * 3 + 1], z1 = qs[i * 3 + 2];
    float best = 0.f;
    int besti = 0;
    for (int k0 = 0; k0 < nc; k0 += CH_TILE) {
        const int cnt = min(CH_TILE, nc - k0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += blockDim.x)
            cand[j] = make_float4(cs[(k0 + j) * 3 + 0], cs[(k0 + j) * 3 + 1], cs[(k0 + j) * 3 + 2], 0.f);
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 c = cand[j];
            const float dx = c.x - x1, dy = c.y - y1, dz = c.z - z1;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if ((k0 + j) == 0 || d < best) {
                best = d;
                besti = k0 + j;
            }
        }
    }
    if (live) {
        if (fwd) {
            dist1[(size_t)b * n + i] = best;
            idx1[(size_t)b * n + i] = besti;
        } else {
            dist2[(size_t)b * m + i] = best;
            idx2[(size_t)b * m + i] = besti;
        }
    }
}

extern "C" int tgp_chamfer_fwd(const float *xyz1, const float *xyz2, int B, int n, int m, float *dist1, float *dist2,
                               int32_t *idx1, int32_t *idx2, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2 && B > 0 && n > 0 && m > 0);
    const int t1 = tgp_cdiv(n, 256), t2 = tgp_cdiv(m, 256);
    hipLaunchKernelGGL(chamfer_fwd_kernel, dim3(tgp_xcd_grid(B, t1 + t2)), dim3(256), 0, tgp_hs(stream), xyz1, xyz2, B, n,
                       m, dist1, dist2, idx1, idx2, t1, t1 + t2);
    return TGP_LAUNCH_RESULT();
}

// One thread per point of either cloud.
//   point j of xyz1: g1[j] += own(j)            then, for j' ascending with idx2[j'] == j:  g1[j] -= term2(j')
//   point j of xyz2: for j' ascending with idx1[j'] == j: g2[j] -= term1(j')   then  g2[j] += own(j)
// which is the order in which the reference's serial loop touches each element.
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                          int B, int n, int m, const float *__restrict__ gd1,
                                                          const float *__restrict__ gd2, const int32_t *__restrict__ idx1,
                                                          const int32_t *__restrict__ idx2, float *__restrict__ g1,
                                                          float *__restrict__ g2, int tiles1, int tiles_per_obj)
{
    __shared__ int32_t hit[CH_TILE];
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const bool first = tile < tiles1; // this thread owns a point of xyz1
    const float *own = first ? xyz1 + (size_t)b * n * 3 : xyz2 + (size_t)b * m * 3;
    const float *oth = first ? xyz2 + (size_t)b * m * 3 : xyz1 + (size_t)b * n * 3;
    const float *gown = first ? gd1 + (size_t)b * n : gd2 + (size_t)b * m;
    const float *goth = first ? gd2 + (size_t)b * m : gd1 + (size_t)b * n;
    const int32_t *iown = first ? idx1 + (size_t)b * n : idx2 + (size_t)b * m;
    const int32_t *ioth = first ? idx2 + (size_t)b * m : idx1 + (size_t)b * n;
    float *gout = first ? g1 + (size_t)b * n * 3 : g2 + (size_t)b * m * 3;
    const int no = first ? n : m, nx = first ? m : n;
    const int j = (first ? tile : tile - tiles1) * 256 + threadIdx.x;
    const bool live = j < no;

    float ax = 0.f, ay = 0.f, az = 0.f, px = 0.f, py = 0.f, pz = 0.f;
    float ox = 0.f, oy = 0.f, oz = 0.f; // own term: g * (p - nearest)
    if (live) {
        ax = gout[j * 3 + 0], ay = gout[j * 3 + 1], az = gout[j * 3 + 2];
        px = own[j * 3 + 0], py = own[j * 3 + 1], pz = own[j * 3 + 2];
        const int j2 = iown[j];
        const float g = gown[j] * 2.0f;
        ox = g * (px - oth[j2 * 3 + 0]);
        oy = g * (py - oth[j2 * 3 + 1]);
        oz = g * (pz - oth[j2 * 3 + 2]);
        if (first) ax = ax + ox, ay = ay + oy, az = az + oz;
    }
    for (int k0 = 0; k0 < nx; k0 += CH_TILE) {
        const int cnt = min(CH_TILE, nx - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) hit[t] = ioth[k0 + t];
        __syncthreads();
        if (live) {
            for (int t = 0; t < cnt; ++t) {
                if (hit[t] == j) {
                    const int jp = k0 + t;
                    const float g = goth[jp] * 2.0f;
                    ax = ax - g * (oth[jp * 3 + 0] - px);
                    ay = ay - g * (oth[jp * 3 + 1] - py);
                    az = az - g * (oth[jp * 3 + 2] - pz);
                }
            }
        }
    }
    if (live) {
        if (!first) ax = ax + ox, ay = ay + oy, az = az + oz;
        gout[j * 3 + 0] = ax, gout[j * 3 + 1] = ay, gout[j * 3 + 2] = az;
    }
}

extern "C" int tgp_chamfer_bwd(const float *xyz1, const float *xyz2, int B, int n, int m, const float *graddist1,
                               const float *graddist2, const int32_t *idx1, const int32_t *idx2, float *gradxyz1,
                               float *gradxyz2, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz1 && xyz2 && graddist1 && graddist2 && idx1 && idx2 && gradxyz1 && gradxyz2 && B > 0 && n > 0 && m > 0);
    const int t1 = tgp_cdiv(n, 256), t2 = tgp_cdiv(m, 256);
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(tgp_xcd_grid(B, t1 + t2)), dim3(256), 0, tgp_hs(stream), xyz1, xyz2, B, n,
                       m, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, t1, t1 + t2);
    return TGP_LAUNCH_RESULT();
}
