// Pairwise metrics of the NOCS pose evaluation (evaluation/eval_utils_v1.py:829-963), all (prediction, ground truth)
// pairs of a whole result set in one launch, in double precision like the numpy original.  One thread per pair.
#include "tgp_common.h"

struct M34 {
    double m[4][4];
};

__device__ __forceinline__ M34 load_rt(const double *p)
{
    M34 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.m[i][j] = p[i * 4 + j];
    return r;
}

// get_3d_bbox + transform_coordinates_3d (:966-1012) + amax / amin as the reference takes them (:842-845): the transformed
// corners form a [3, 8] array and the reduction runs over axis 0, so the "extent" is, per CORNER, the largest and the
// smallest of its three coordinates -- eight (lo, hi) pairs, corner c of one box meeting corner c of the other.  Corner
// order is get_3d_bbox's: x sign from bit 1, y sign from bit 2, z sign from bit 0 of c.
__device__ __forceinline__ void box_extent(const M34 &rt, const double *s, double (&lo)[8], double (&hi)[8])
{
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double x = (c & 2) ? -s[0] / 2 : s[0] / 2, y = (c & 4) ? -s[1] / 2 : s[1] / 2, z = (c & 1) ? -s[2] / 2 : s[2] / 2;
        double v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) v[a] = ((rt.m[a][0] * x + rt.m[a][1] * y) + rt.m[a][2] * z) + rt.m[a][3];
        const double p0 = v[0] / v[3], p1 = v[1] / v[3], p2 = v[2] / v[3];
        lo[c] = fmin(fmin(p0, p1), p2), hi[c] = fmax(fmax(p0, p1), p2);
    }
}

__device__ __forceinline__ double box_iou(const double (&lo1)[8], const double (&hi1)[8], const double (&lo2)[8], const double (&hi2)[8])
{
    double inter = 1.0, v1 = 1.0, v2 = 1.0, dmin = INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const double d = fmin(hi1[a], hi2[a]) - fmax(lo1[a], lo2[a]);
        dmin = fmin(dmin, d);
        inter *= d;
        v1 *= hi1[a] - lo1[a], v2 *= hi2[a] - lo2[a];
    }
    if (dmin < 0) inter = 0;
    return inter / ((v1 + v2) - inter);
}

// compute_3d_iou_new (:829-887): symmetric != 0 -> the best IoU over 20 rotations of box 1 about its y axis
__global__ void iou3d_pairs_kernel(const double *__restrict__ RT1, const double *__restrict__ RT2, const double *__restrict__ S1,
                                   const double *__restrict__ S2, const int *__restrict__ symmetric, int P, double *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P) return;
    const M34 r1 = load_rt(RT1 + (size_t)t * 16), r2 = load_rt(RT2 + (size_t)t * 16);
    const double *s1 = S1 + (size_t)t * 3, *s2 = S2 + (size_t)t * 3;
    double lo2[8], hi2[8], lo1[8], hi1[8];
    box_extent(r2, s2, lo2, hi2);
    if (!symmetric[t]) {
        box_extent(r1, s1, lo1, hi1);
        out[t] = box_iou(lo1, hi1, lo2, hi2);
        return;
    }
    double best = 0.0;
    for (int i = 0; i < 20; ++i) {
        const double th = 2.0 * 3.141592653589793 * i / 20.0;
        const double c = cos(th), s = sin(th);
        M34 r = r1;        // RT_1 @ [[c,0,s],[0,1,0],[-s,0,c]]: column 0 = c col0 - s col2, column 2 = s col0 + c col2
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            r.m[a][0] = r1.m[a][0] * c + r1.m[a][2] * -s;
            r.m[a][2] = r1.m[a][0] * s + r1.m[a][2] * c;
        }
        box_extent(r, s1, lo1, hi1);
        best = fmax(best, box_iou(lo1, hi1, lo2, hi2));
    }
    out[t] = best;
}

extern "C" int tgp_iou3d_pairs(const double *RT1, const double *RT2, const double *scales1, const double *scales2, const int *symmetric,
                               int P, double *iou, tgp_stream_t stream)
{
    TGP_REQUIRE(RT1 && RT2 && scales1 && scales2 && symmetric && iou && P > 0);
    hipLaunchKernelGGL(iou3d_pairs_kernel, dim3(tgp_cdiv(P, 128)), dim3(128), 0, tgp_hs(stream), RT1, RT2, scales1, scales2, symmetric, P, iou);
    return TGP_LAUNCH_RESULT();
}

// compute_RT_degree_cm_symmetry (:890-963).  mode 0: angle of R1 R2^T; 1: angle between the y axes (bottle / can / bowl, mug
// with a hidden handle); 2: the smaller of the angles of R1 R2^T and R1 diag(-1,1,-1) R2^T (phone / eggbox / glue).
// out[t] = {degrees, centimetres}.  acos of an argument just outside [-1, 1] is NaN, as numpy's.
__global__ void rt_error_pairs_kernel(const double *__restrict__ RT1, const double *__restrict__ RT2, const int *__restrict__ mode, int P,
                                      double *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P) return;
    const M34 a = load_rt(RT1 + (size_t)t * 16), b = load_rt(RT2 + (size_t)t * 16);
    auto det3 = [](const M34 &r) {
        return r.m[0][0] * (r.m[1][1] * r.m[2][2] - r.m[1][2] * r.m[2][1]) - r.m[0][1] * (r.m[1][0] * r.m[2][2] - r.m[1][2] * r.m[2][0]) +
               r.m[0][2] * (r.m[1][0] * r.m[2][1] - r.m[1][1] * r.m[2][0]);
    };
    const double ka = cbrt(det3(a)), kb = cbrt(det3(b));
    double R1[3][3], R2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R1[i][j] = a.m[i][j] / ka, R2[i][j] = b.m[i][j] / kb;
    double theta;
    const int m = mode[t];
    if (m == 1) {
        double dot = 0, n1 = 0, n2 = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) dot += R1[i][1] * R2[i][1], n1 += R1[i][1] * R1[i][1], n2 += R2[i][1] * R2[i][1];
        theta = acos(dot / (sqrt(n1) * sqrt(n2)));
    } else {
        double tr = 0, tr_rot = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double pr = R1[i][j] * R2[i][j];
                tr += pr;
                tr_rot += (j == 1) ? pr : -pr;
            }
        theta = acos((tr - 1) / 2);
        if (m == 2) theta = fmin(theta, acos((tr_rot - 1) / 2));
    }
    double d2 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) d2 += (a.m[i][3] - b.m[i][3]) * (a.m[i][3] - b.m[i][3]);
    out[(size_t)t * 2] = theta * (180.0 / 3.141592653589793);
    out[(size_t)t * 2 + 1] = sqrt(d2) * 100;
}

extern "C" int tgp_rt_error_pairs(const double *RT1, const double *RT2, const int *mode, int P, double *out, tgp_stream_t stream)
{
    TGP_REQUIRE(RT1 && RT2 && mode && out && P > 0);
    hipLaunchKernelGGL(rt_error_pairs_kernel, dim3(tgp_cdiv(P, 128)), dim3(128), 0, tgp_hs(stream), RT1, RT2, mode, P, out);
    return TGP_LAUNCH_RESULT();
}
