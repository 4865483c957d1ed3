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
#include <stdlib.h>

#include "tgp_common.h"

#include "gemm_epi.h"

#define GEMM_LDPAD 4

// One BM x BN output tile at (m0, n0) of batch z, computed by 64*NWM*NWN threads (NWM x NWN waves, each
// owning a (BM/NWM) x (BN/NWN) sub-tile as 32x32 MFMA tiles).  BK-wide K-tiles; DBUF selects two LDS buffers
// and one barrier per K-tile, otherwise one buffer and two barriers.
// Measured on MI355X (scripts/gemm_variants.py, M=32768 N=4096 K=1280, row stride 1292): what matters is waves
// per SIMD -- 256-thread workgroups (2-3 waves/SIMD) reach 58-90 TF whatever the tile, 1024-thread workgroups
// (4 waves/SIMD) 124 TF on 128x128 and 131-137 TF on 256x256 tiles; BK=16 beats BK=32 by 25 %.
template <int BM, int BN, int NWM, int NWN, int BK, bool DBUF, bool NATURAL_K, bool DIST>
__device__ __forceinline__ void gemm_tile(const GemmParams &p, const int m0, const int n0, const int z, float *smem)
{
    constexpr int THREADS = 64 * NWM * NWN;
    constexpr int LD = BK + GEMM_LDPAD;
    constexpr int WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int F4 = BK / 4;                      // float4 per staged row
    constexpr int RPP = THREADS / F4;               // rows per staging pass
    constexpr int PA = (BM + RPP - 1) / RPP, PW = (BN + RPP - 1) / RPP;
    constexpr int BUF = (BM + BN) * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    const float *A = p.A + (int64_t)z * p.sA;
    const float *W = p.W + (int64_t)z * p.sW;

    const int kq = tid % F4, r0 = tid / F4;
    float4 ra[PA], rw[PW];

    // Guards without branches: every lane loads from a clamped (always valid) address, then selects.
    auto load_tile = [&](int kt) {
        const int kcol = kt * BK + kq * 4;
        const bool kok = kcol < p.K;
        const int kc = kok ? kcol : 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (RPP * i + r0 < BM) {
                const int row = m0 + r0 + RPP * i;
                const bool ok = kok && row < p.M;
                const float4 v = *reinterpret_cast<const float4 *>(A + (int64_t)(row < p.M ? row : p.M - 1) * p.lda + kc);
                ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            if (RPP * i + r0 < BN) {
                const int row = n0 + r0 + RPP * i;
                const bool ok = kok && row < p.N;
                const float4 v = *reinterpret_cast<const float4 *>(W + (int64_t)(row < p.N ? row : p.N - 1) * p.ldw + kc);
                rw[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // NATURAL_K reads single floats (row r, column 2s + h) -- with rows LD = BK + 4 floats apart, rows r and r + 8 share a
    // bank: a 4-way conflict on every operand read.  The two low column bits are therefore XOR-ed with (row >> 3) & 3: the
    // four rows that collide land in the four banks of one aligned quad (a float4 store just permutes its components).
    auto swz = [&](const float4 v, const int row) -> float4 {
        if constexpr (!NATURAL_K) return v;
        const int g = (row >> 3) & 3;
        const float e[4] = {v.x, v.y, v.z, v.w};
        return make_float4(e[0 ^ g], e[1 ^ g], e[2 ^ g], e[3 ^ g]);
    };
    auto store_tile = [&](int buf) {
        float *as = smem + buf * BUF;
        float *ws = as + BM * LD;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (RPP * i + r0 < BM) *reinterpret_cast<float4 *>(as + (r0 + RPP * i) * LD + kq * 4) = swz(ra[i], r0 + RPP * i);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            if (RPP * i + r0 < BN) *reinterpret_cast<float4 *>(ws + (r0 + RPP * i) * LD + kq * 4) = swz(rw[i], r0 + RPP * i);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int numK = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < numK; ++kt) {
        const bool more = (kt + 1) < numK;
        if (more) load_tile(kt + 1);
        const int cur = DBUF ? (kt & 1) : 0;
        const float *as = smem + cur * BUF + (wm * WTM + r) * LD;
        const float *ws = smem + cur * BUF + BM * LD + (wn * WTN + r) * LD;
        if constexpr (NATURAL_K) {
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = as[i * 32 * LD + ((2 * s + h) ^ ((r >> 3) & 3))];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = ws[j * 32 * LD + ((2 * s + h) ^ ((r >> 3) & 3))];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LD + kk * 8 + h * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(ws + j * 32 * LD + kk * 8 + h * 4);
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
        if (DBUF) {
            if (more) store_tile((kt + 1) & 1);
            __syncthreads();
        } else {
            __syncthreads();
            if (more) store_tile(0);
            __syncthreads();
        }
    }

    if constexpr (DIST) {
        gemm_epilogue<TM, TN, WTM, WTN, true>(p, acc, m0, n0, z, wm, wn, r, h);
    } else {
        if (p.gres1 || p.gres2) gemm_epilogue_gather<TM, TN, WTM, WTN>(p, acc, m0, n0, z, wm, wn, r, h);
    else if (!(p.rowbias || p.cm) || p.rows_per_obj >= WTM) gemm_epilogue_fast<TM, TN, WTM, WTN>(p, acc, m0, n0, z, wm, wn, r, h);
        else gemm_epilogue<TM, TN, WTM, WTN, false>(p, acc, m0, n0, z, wm, wn, r, h);
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32-accurate GEMM on the bf16 matrix cores: x = hi + mid + lo with three bf16 terms (3 x 8 significand bits =
// fp32's 24), a*b ~ hh + hm + mh + hl + lh + mm (the dropped ml, lm, ll terms are <= 2^-23 |ab|), every term
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six MFMAs of 32 cycles replace sixteen fp32 MFMAs of 64
// cycles per 32x32x16 block: 2.67x fewer matrix-core cycles at fp32-level accuracy (measured error vs fp64
// equals the fp32 MFMA kernel's: tests/test_gpu_parity.py::test_gemm_split_bf16_accuracy).
// W is split once at weight-pack time (tgp_split_bf16); A (activations, fp32 in HBM) is split while it is
// staged into LDS.  1024 threads, 4x4 waves, wave tile 64x64 (BIG 256x256) or 32x32 (MID 128x128), BK = 16.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void split3(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo)
{
    bf16x4 h, m, l;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (__bf16)x[i];
        const float r1 = x[i] - (float)h[i];
        m[i] = (__bf16)r1;
        const float r2 = r1 - (float)m[i];
        l[i] = (__bf16)r2;
    }
    hi = __builtin_bit_cast(uint2, h);
    mid = __builtin_bit_cast(uint2, m);
    lo = __builtin_bit_cast(uint2, l);
}

// W planes are stored interleaved per K-tile: Wsplit[row][k / 16][plane][16] bf16, so the 3 x 16 values one
// K-tile needs from a row are 96 contiguous bytes = six 16-byte chunks (chunk c: plane c / 2, k half c % 2).
//
// F16 = true is the two-term fp16 variant: x = hi + lo with 2 x 11 significand bits, a*b ~ hh + hl + lh (the dropped ll
// term is <= 2^-22 |ab|; the operand itself is represented to ~2^-23), three v_mfma_f32_32x32x16_f16 per block instead
// of six: half the matrix-core cycles again.  Its price is fp16's range: |a|, |w| must stay below 65504 (an
// overflowing operand becomes inf and the output NaN -- loud, not silent) and terms below 6e-8 vanish (an absolute
// error far under fp32's rounding of any O(1) dot product).  W planes: Wsplit[row][k / 16][2][16].
// (f16x8 / f16x4 / f32x4 and split2 live in gemm_epi.h: the epilogue writes fp16 planes too)

//
// KG > 1 is the quarter tile of the launch's last round: the workgroup's waves form KG groups that each take every KG-th
// K-tile of the same BM x BN output (intra-workgroup split-K), so a 128 x 128 tile keeps 64 x 64 wave tiles -- 12 MFMAs
// per wave between barriers like the big tile -- and needs a quarter of the K-steps; the groups' accumulators are combined
// through LDS in a fixed order before the epilogue.  LDS rows are "virtual rows" v = group * BM + row.
template <int BM, int BN, int NWM, int NWN, int PD, bool F16, bool SKEW, int KG = 1, int STAGES = (PD > 0 ? 2 : 1), bool PLANES = false>
__device__ __forceinline__ void gemm_split_tile(const GemmParams &p, const int m0, const int n0, const int z, char *smem)
{
    constexpr int WPG = NWM * NWN;            // waves per K-group
    constexpr int THREADS = 64 * WPG * KG;
    constexpr int BK = 16;
    constexpr int NP = F16 ? 2 : 3;           // operand planes
    constexpr int ROWB = BK * 2 + 16;         // LDS row of one plane: 16 bf16 + 16 B pad (conflict-free ds_read_b128)
    constexpr int WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int VA = BM * KG, VW = BN * KG; // virtual rows staged per step
    constexpr int RPP = THREADS / 4;          // A: 4 threads per row, one float4 each
    constexpr int PA = (VA + RPP - 1) / RPP;
    constexpr int NCH = 2 * NP;               // 16-byte chunks per row and K-tile
    constexpr int WCH = VW * NCH;             // W: 16-byte chunks per step
    constexpr int PW = (WCH + THREADS - 1) / THREADS;
    // plane strides carry 32 extra bytes: the hi / mid / lo planes then start 8 banks apart, which removes the 3-way
    // conflicts of the staging writes (one row's six W chunks, or one A quad's three terms, hit distinct banks)
    constexpr int PLANE_A = VA * ROWB + 32, PLANE_W = VW * ROWB + 32;
    constexpr int BUFB = NP * (PLANE_A + PLANE_W);
    static_assert(KG == 1 || STAGES == 1, "the split-K tile uses one LDS stage");
    char *lds_a = smem;
    char *lds_w = smem + NP * PLANE_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kg = wave / WPG;
    const int wm = (wave % WPG) / NWN, wn = (wave % WPG) % NWN;
    const int r = lane & 31, h = lane >> 5;
    const float *A = p.A + (int64_t)z * p.sA;
    const uint16_t *WS = p.Wsplit + (int64_t)z * p.sWS;
    const int64_t wrow = NP * (int64_t)p.ldws; // elements per row of Wsplit

    // PD = 0: one LDS stage.  PD >= 1: two LDS stages, global loads PD K-tiles ahead of the MFMAs through a ring of PD
    // register sets (8 VGPRs each on the big tile): one K-tile of MFMAs is ~0.6 us, shorter than a loaded memory
    // system's latency, so a single tile of lookahead leaves the matrix cores waiting.
    constexpr bool DBUF = STAGES == 2;        // LDS stages; with one stage a step is compute | barrier | store | barrier
    constexpr int NS = PD > 0 ? PD : 1;
    const int kq = tid & 3, r0 = tid >> 2;
    float4 ra_[NS][PA];
    uint4 rw_[NS][PW];

    // Operand loads are buffer loads against per-tile resource descriptors whose extent ends at the operand's last
    // valid row: rows past M (N) read as zero in hardware, so the loads carry no predicates, sit in straight-line code
    // and stay in flight across K-tiles (with per-lane guards the compiler wraps each load in a branch and has to drain
    // vmcnt to zero around it, which serialises the prefetch).  The K tail of A is masked when the tile is stored.
    const int64_t a_rows = p.M - m0, w_rows = p.N - n0;
    const int64_t a_bytes = a_rows * p.lda * 4 - (p.lda - p.K) * 4, w_bytes = w_rows * wrow * 2 - (p.ksplit ? z * p.sWS * 2 : 0);
    const float asc = p.a_scale ? p.a_scale[0] : 1.f;
    // fp16 range guard (see after the K loop): armed in the production form of the fp16 tile
#ifdef TGP_NO_RANGE_GUARD                     // measurement builds only (scripts/ab_bench.py: what does the guard cost?)
    constexpr bool RANGE_GUARD = false;
#else
    constexpr bool RANGE_GUARD = F16 && KG == 1 && STAGES == 2 && !SKEW;
#endif
    float amax = 0.f;                          // the largest |a * asc| this thread has split
    __shared__ int s_range_flag;
    // (round 3) the small side of fp16's range: the lo plane of x is ~2^-11 x, and fp16 resolves nothing finer than 6e-8 (its
    // subnormal step), so an operand is reproduced to an ABSOLUTE 3e-8 at best: 3e-6 relative for x ~ 1e-2, percent level for
    // x ~ 1e-6 (measured: 3e-5 of the output scale at activations of 1e-3).  The tile's largest magnitude is collected with one
    // LDS atomic per thread (positive floats order like their bit patterns); a tile that stays under 2^-4 recomputes in exact
    // fp32 like an overflowing one.  A tile holding ordinary magnitudes beside tiny ones keeps the split: its absolute error
    // (3e-8 per operand) is invisible at the scale its large entries give the output.  Not armed when the caller scales the
    // operand itself (a_scale: the backward lifts gradients to 2^14, and their quiet tiles must not pay the exact path).
    __shared__ unsigned s_amax_bits;
    if (RANGE_GUARD && threadIdx.x == 0) s_range_flag = 0, s_amax_bits = 0u;      // ordered before every use by the prologue's barrier
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A + (int64_t)m0 * p.lda), 0, (int)(a_bytes > 0x7fffffff ? 0x7fffffff : a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(WS + (int64_t)n0 * wrow), 0, (int)(w_bytes > 0x7fffffff ? 0x7fffffff : w_bytes), 0x00020000);
    int voff_a[PA], voff_w[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int v = r0 + RPP * i;            // virtual row: K-group v / BM, tile row v % BM
        voff_a[i] = ((v % BM) * p.lda + (v / BM) * BK + kq * 4) * 4;
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int g = tid + THREADS * i;
        const int v = g / NCH;
        voff_w[i] = (int)((v % BN) * wrow * 2) + (v / BN) * (32 * NP) + (g % NCH) * 16;
    }
    // kt counts steps: step kt covers the K-tiles kt * KG .. kt * KG + KG - 1 (one per K-group)
    auto load_tile = [&](int kt, float4 (&ra)[PA], uint4 (&rw)[PW]) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (RPP * i + r0 < VA)
                ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[i], kt * (KG * BK * 4), 0));
#pragma unroll
        for (int i = 0; i < PW; ++i)
            if (tid + THREADS * i < WCH)
                rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w[i], kt * (KG * 32 * NP), 0));
    };
    auto store_tile = [&](int buf, const int kt, const float4 (&ra_in)[PA], const uint4 (&rw)[PW]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (RPP * i + r0 < VA) {
                // K % 4 == 0: a quad is wholly inside or outside
                const bool kok = (kt * KG + (r0 + RPP * i) / BM) * BK + kq * 4 < p.K;
                char *dst = lds_a + buf * BUFB + (r0 + RPP * i) * ROWB + kq * 8;
                float4 ra[PA];   // masked with AND, not a select: a select on a pending load is compiled into a branch
                const uint32_t km = kok ? 0xffffffffu : 0u;
                const uint4 rbits = __builtin_bit_cast(uint4, ra_in[i]);
                ra[i] = __builtin_bit_cast(float4, make_uint4(rbits.x & km, rbits.y & km, rbits.z & km, rbits.w & km));
                if constexpr (F16) {
                    ra[i].x *= asc, ra[i].y *= asc, ra[i].z *= asc, ra[i].w *= asc;       // exact: asc is 1 or a power of two
                    if constexpr (RANGE_GUARD)
                        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(ra[i].x), fabsf(ra[i].y))), fmaxf(fabsf(ra[i].z), fabsf(ra[i].w)));
                    uint2 q0, q1;
                    split2(ra[i], q0, q1);
                    *reinterpret_cast<uint2 *>(dst) = q0;
                    *reinterpret_cast<uint2 *>(dst + PLANE_A) = q1;
                } else {
                    uint2 q0, q1, q2;
                    split3(ra[i], q0, q1, q2);
                    *reinterpret_cast<uint2 *>(dst) = q0;
                    *reinterpret_cast<uint2 *>(dst + PLANE_A) = q1;
                    *reinterpret_cast<uint2 *>(dst + 2 * PLANE_A) = q2;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int g = tid + THREADS * i;
            if (g < WCH) {
                const int row = g / NCH, c = g % NCH;
                *reinterpret_cast<uint4 *>(lds_w + buf * BUFB + (c >> 1) * PLANE_W + row * ROWB + (c & 1) * 16) = rw[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int numK = ((p.K + BK - 1) / BK + KG - 1) / KG;     // steps
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();
    load_tile(0, ra_[0], rw_[0]);
    store_tile(0, 0, ra_[0], rw_[0]);
    if (RANGE_GUARD && numK == 1) {
        if (amax >= 65504.f) s_range_flag = 1;
        atomicMax(&s_amax_bits, __float_as_uint(amax));
    }
#pragma unroll
    for (int s = 1; s < NS; ++s) load_tile(s, ra_[s], rw_[s]);
    __syncthreads();
    if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    // step kt: tile kt is in LDS stage kt & 1; registers slot (kt + 1) % NS hold tile kt + 1 (loaded NS steps ago);
    // the load of tile kt + NS goes into slot kt % NS, whose tile kt was written to LDS during step kt - 1.
    // SKEW (fp16, PD >= 2) = interleaved schedule: tile kt + 1 is already in registers when step kt starts, so its
    // conversion and LDS store need not sit in one block between two groups of MFMAs, where all four waves of a SIMD
    // reach it together and leave the matrix core idle.  The step is one basic block (loads and stores unconditional:
    // past the last K-tile they move zeros) and sched_group_barrier asks for 1 MFMA : 3 VALU throughout, so every
    // wave's conversion runs in the shadow of its own MFMAs.
    static_assert(!SKEW || (PD >= 2 && F16), "the interleaved schedule needs two tiles of prefetch");
    const int role = 1;
    auto step = [&](const int kt, float4 (&ra_new)[PA], uint4 (&rw_new)[PW], float4 (&ra_next)[PA], uint4 (&rw_next)[PW]) {
        const bool more = (kt + 1) < numK;
        // issued unconditionally (past the last K-tile the descriptor's bounds make it a load of zeros that is never
        // stored): a conditional issue would force the compiler to drain vmcnt to zero before the next tile's store
        if (DBUF || NS > 1 || more) load_tile(kt + NS, ra_new, rw_new);
        const int cur = DBUF ? (kt & 1) : 0;
        const char *as = lds_a + cur * BUFB + (kg * BM + wm * WTM + r) * ROWB + h * 16;
        const char *ws = lds_w + cur * BUFB + (kg * BN + wn * WTN + r) * ROWB + h * 16;
        uint4 a[TM][NP];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) a[i][q] = *reinterpret_cast<const uint4 *>(as + i * 32 * ROWB + q * PLANE_A);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            uint4 b[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const uint4 *>(ws + j * 32 * ROWB + q * PLANE_W);
            // terms smallest first; consecutive MFMAs alternate between the accumulators of this column
#define SPLIT_TERM(QA, QB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                       \
    {                                                                                                                    \
        if constexpr (F16)                                                                                               \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][QA]),                      \
                                                               __builtin_bit_cast(f16x8, b[QB]), acc[i][j], 0, 0, 0);    \
        else                                                                                                             \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][QA]),                    \
                                                                __builtin_bit_cast(bf16x8, b[QB]), acc[i][j], 0, 0, 0);  \
    }
            if constexpr (F16) {
                SPLIT_TERM(0, 1)
                SPLIT_TERM(1, 0)
                SPLIT_TERM(0, 0)
            } else {
                SPLIT_TERM(0, 2)
                SPLIT_TERM(2, 0)
                SPLIT_TERM(1, 1)
                SPLIT_TERM(0, 1)
                SPLIT_TERM(1, 0)
                SPLIT_TERM(0, 0)
            }
#undef SPLIT_TERM
            // With two LDS stages the next tile can be split and written while this wave still has MFMAs to issue:
            // done after the first column block, the VALU / LDS-write work overlaps the other waves' MFMAs instead of
            // sitting between the last MFMA and the barrier.
            if (!SKEW && DBUF && j == (TN > 1 ? TN / 2 - 1 : 0) && more) {
                store_tile((kt + 1) & 1, kt + 1, ra_next, rw_next);
                // the last K-tile has just been staged: this thread's amax is final, and the barriers that end this step and the
                // next one order the flag before anybody reads it
                if (RANGE_GUARD && kt + 2 >= numK) {
                    if (amax >= 65504.f) s_range_flag = 1;
                    atomicMax(&s_amax_bits, __float_as_uint(amax));
                }
            }
        }
        if constexpr (SKEW) {
            store_tile((kt + 1) & 1, kt + 1, ra_next, rw_next);
            // masks: 0x2 VALU, 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
            __builtin_amdgcn_sched_group_barrier(0x20, PA + PW, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM * NP + NP, 0);
#pragma unroll
            for (int g = 0; g < TM * TN * 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                if (g == 1) __builtin_amdgcn_sched_group_barrier(0x100, (TN - 1) * NP, 0);
                if (g == TM * TN * 3 - 3) __builtin_amdgcn_sched_group_barrier(0x200, 3 * PA, 0);
                if (g == TM * TN * 3 - 2) __builtin_amdgcn_sched_group_barrier(0x200, PW, 0);
            }
        }
        if (DBUF) {
            __syncthreads();
        } else {
            __syncthreads();
            if (more) {
                if constexpr (NS == 1) store_tile(0, kt + 1, ra_new, rw_new);
                else store_tile(0, kt + 1, ra_next, rw_next);
            }
            __syncthreads();
        }
    };
    if constexpr (NS == 1) {
        for (int kt = 0; kt < numK; ++kt) step(kt, ra_[0], rw_[0], ra_[0], rw_[0]);
    } else {
        // unrolled by NS so that the ring slots are compile-time register sets
        for (int kt0 = 0; kt0 < numK; kt0 += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (kt0 + s < numK) step(kt0 + s, ra_[s], rw_[s], ra_[(s + 1) % NS], rw_[(s + 1) % NS]);
        }
    }
    // fp16 range guard.  The two-term fp16 split needs |a| < 65504 (an operand beyond it splits into inf and the sums become NaN).
    // Every thread has tracked the largest magnitude it split (one register, two v_max3 per staged quad) and, after staging its last
    // K-tile, raised a flag in LDS if that magnitude left fp16's range (before the loop's last barrier, so no extra one is needed
    // here).  A tile that met such an operand throws its sums away and
    // recomputes them in exact fp32: v_mfma_f32_32x32x2_f32 fed straight from global memory (A and the fp32 weight the caller
    // always passes beside the split one), same accumulator layout, same epilogue.  Slower by an order of magnitude, but only for
    // the tiles that need it; no host involvement, nothing read back, the launch stays capturable.  Every tile of this network's
    // forward at sane weights takes the fast path.  (Weights are range-checked when they are packed: ops.split_w.  Not armed for
    // the K-split form, whose fp32 W pointer is a placeholder and whose A operand the caller has already scaled.)
    bool exact_fallback = false;
    if constexpr (RANGE_GUARD) {
        // 0x3d800000 = 2^-4; an all-zero tile (bits 0) needs nothing
        const bool tiny_tile = !p.a_scale && s_amax_bits != 0u && s_amax_bits < 0x3d800000u;
        if ((s_range_flag != 0 || tiny_tile) && !p.ksplit) {                      // workgroup-uniform: written before the loop's last barrier
            exact_fallback = true;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            // lane (r, h) supplies k = 8 t + 4 h + s to MFMA sub-step s of the 8-wide group t, for both operands; rows / columns past
            // the end are clamped to the last valid one (their results are never stored)
            const float *W32 = p.W + (int64_t)z * p.sW;
            const float *ag[TM], *wg[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + wm * WTM + i * 32 + r;
                ag[i] = A + (int64_t)(row < p.M ? row : p.M - 1) * p.lda + 4 * h;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WTN + j * 32 + r;
                wg[j] = W32 + (int64_t)(col < p.N ? col : p.N - 1) * p.ldw + 4 * h;
            }
#pragma unroll 1
            for (int k0 = 0; k0 < p.K; k0 += 8) {
                const bool ok = k0 + 4 * h < p.K;                                 // K % 4 == 0: the quad is wholly inside or outside
                float4 av[TM], wv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = ok ? *reinterpret_cast<const float4 *>(ag[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < TN; ++j) wv[j] = ok ? *reinterpret_cast<const float4 *>(wg[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, wv[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, wv[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, wv[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, wv[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    if constexpr (KG > 1) {
        // combine the K-groups' accumulators: (g0 + g1) + (g2 + g3), one group's 64 registers x 64 lanes through LDS at a
        // time (lane-major rows: conflict free); the loop's last barrier has already retired every operand read
        static_assert(KG == 4 || KG == 2, "pairwise combination");
        float *slot = reinterpret_cast<float *>(smem) + (wave % WPG) * (TM * TN * 16 * 64) + lane;
        auto give = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) slot[((i * TN + j) * 16 + e) * 64] = acc[i][j][e];
        };
        auto take = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += slot[((i * TN + j) * 16 + e) * 64];
        };
        auto hand = [&](int from, int to) {
            if (kg == from) give();
            __syncthreads();
            if (kg == to) take();
            __syncthreads();
        };
        hand(1, 0);
        if (KG == 4) hand(3, 2), hand(2, 0);
        if (kg != 0) return;
    }
    if (p.c_scale && !exact_fallback) {           // (the exact recomputation used the unscaled operands)
        const float cs = p.c_scale[0];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= cs;
    }
    // PLANES: a kernel instance of its own (the small-tile kernel only) whose epilogue also writes the result as fp16 planes; in
    // the same instance the extra state cost the 128-register tiles 60-70 spilled registers
    if (KG == 1 && p.vec_epi && (!(p.rowbias || p.cm) || p.rows_per_obj >= WTM))
        gemm_epilogue_lds<TM, TN, WTM, WTN, PLANES>(p, acc, m0, n0, z, wm, wn, r, h, reinterpret_cast<float *>(smem) + wave * 1024);
    else if (!(p.rowbias || p.cm) || p.rows_per_obj >= WTM) gemm_epilogue_fast<TM, TN, WTM, WTN>(p, acc, m0, n0, z, wm, wn, r, h);
    else gemm_epilogue<TM, TN, WTM, WTN, false>(p, acc, m0, n0, z, wm, wn, r, h);
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long *o = p.stamps + 5 * (size_t)blockIdx.x;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = st0, o[1] = st1, o[2] = st2, o[3] = __builtin_amdgcn_s_memrealtime(), o[4] = xcc;
    }
}

template <int PD, bool F16, bool SKEW>
__global__ __launch_bounds__(1024) void gemm_split_kernel(GemmParams p)
{
    // two stages of the 256 x 256 tile = one stage of the split-K quarter tile (4 x (128 + 128) virtual rows)
    __shared__ __attribute__((aligned(16))) char smem[2 * (F16 ? 2 : 3) * ((256 + 256) * 48 + 64)];
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    int seg = 0;
    while (seg < p.nseg - 1 && (int)blockIdx.x >= p.seg_end[seg]) ++seg;
    int L = p.seg_base[seg] + (int)blockIdx.x - (seg ? p.seg_end[seg - 1] : 0);
    if (!((p.seg_small >> seg) & 1)) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile<256, 256, 4, 4, PD, F16, SKEW>(p, (L / p.tiles_n_big) * 256, (L % p.tiles_n_big) * 256, z, smem);
    } else {
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        const int sm0 = p.mt_big * 256 + (L / p.tiles_n_small) * 128, sn0 = (L % p.tiles_n_small) * 128;
        // (an intra-workgroup split-K form of this tile -- template parameter KG -- was measured: 73 instead of 80 us for
        // the wide layer's last round, but it sums K in another order than the big tiles, so an object's result would
        // depend on whether its rows fall into the last 128 of the batch; not used)
        gemm_split_tile<128, 128, 4, 4, PD, F16, false>(p, sm0, sn0, z, smem);
    }
}

// 256 x 256 fp16-split tile with 32-wide K steps: half the barriers of the BK = 16 form (24 MFMAs per wave between
// barriers).  LDS rows are 64 bytes (32 fp16) without padding -- two stages of 2 planes x 512 rows = 128 KB -- and the four
// 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3, which makes the ds_read_b128 fragment reads (16 lanes per
// LDS cycle: rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of a 32-row block) hit 64 distinct banks.  One register slot of
// prefetch (a step is ~2 us of matrix work, enough to cover the loads issued one step ahead).
template <int BM, int BN, int NWM, int NWN>
__device__ __forceinline__ void gemm_split_tile32(const GemmParams &p, const int m0, const int n0, const int z, char *smem)
{
    constexpr int THREADS = 64 * NWM * NWN;
    constexpr int BK = 32, ROWB = 64, NP = 2;
    constexpr int WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int RPP = THREADS / 8;                 // A: 8 threads per row, one float4 each
    constexpr int PA = BM / RPP;
    constexpr int WCH = BN * 8;                      // W: 16-byte chunks per step (2 K-tiles x 2 planes x 2 halves)
    constexpr int PW = WCH / THREADS;
    constexpr int PLANE_A = BM * ROWB, PLANE_W = BN * ROWB;
    constexpr int BUFB = NP * (PLANE_A + PLANE_W);
    static_assert(BM % RPP == 0 && WCH % THREADS == 0, "tile / thread mapping");
    char *lds_a = smem;
    char *lds_w = smem + NP * PLANE_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    const float *A = p.A + (int64_t)z * p.sA;
    const uint16_t *WS = p.Wsplit + (int64_t)z * p.sWS;
    const int64_t wrow = NP * (int64_t)p.ldws;

    const int kq = tid & 7, r0 = tid >> 3;
    float4 ra[PA];
    uint4 rw[PW];
    const int64_t a_rows = p.M - m0, w_rows = p.N - n0;
    const int64_t a_bytes = a_rows * p.lda * 4 - (p.lda - p.K) * 4, w_bytes = w_rows * wrow * 2;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A + (int64_t)m0 * p.lda), 0, (int)(a_bytes > 0x7fffffff ? 0x7fffffff : a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(WS + (int64_t)n0 * wrow), 0, (int)(w_bytes > 0x7fffffff ? 0x7fffffff : w_bytes), 0x00020000);
    int voff_a[PA], voff_w[PW], lds_wo[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) voff_a[i] = ((r0 + RPP * i) * p.lda + kq * 4) * 4;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int g = tid + THREADS * i;
        const int row = g >> 3, cc = g & 7;           // cc: K-tile t = cc >> 2, plane = (cc >> 1) & 1, half = cc & 1
        voff_w[i] = (int)(row * wrow * 2) + (cc >> 2) * 64 + ((cc >> 1) & 1) * 32 + (cc & 1) * 16;
        const int chunk = ((cc >> 2) * 2 + (cc & 1)) ^ ((row >> 2) & 3);
        lds_wo[i] = ((cc >> 1) & 1) * PLANE_W + row * ROWB + chunk * 16;
    }
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[i], kt * (BK * 4), 0));
#pragma unroll
        for (int i = 0; i < PW; ++i)
            rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w[i], kt * 128, 0));
    };
    auto store_tile = [&](int buf, const int kt) {
        const uint32_t km = (kt * BK + kq * 4 < p.K) ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = r0 + RPP * i;
            const uint4 rbits = __builtin_bit_cast(uint4, ra[i]);
            const float4 v = __builtin_bit_cast(float4, make_uint4(rbits.x & km, rbits.y & km, rbits.z & km, rbits.w & km));
            uint2 q0, q1;
            split2(v, q0, q1);
            char *dst = lds_a + buf * BUFB + row * ROWB + (((kq >> 1) ^ ((row >> 2) & 3)) * 16) + (kq & 1) * 8;
            *reinterpret_cast<uint2 *>(dst) = q0;
            *reinterpret_cast<uint2 *>(dst + PLANE_A) = q1;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) *reinterpret_cast<uint4 *>(lds_w + buf * BUFB + lds_wo[i]) = rw[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int numK = (p.K + BK - 1) / BK;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    const int swz = (r >> 2) & 3;
    for (int kt = 0; kt < numK; ++kt) {
        load_tile(kt + 1);        // unconditional: past K the descriptors return zeros / the A mask clears the quad
        const int cur = kt & 1;
        const char *as = lds_a + cur * BUFB + (wm * WTM + r) * ROWB;
        const char *ws = lds_w + cur * BUFB + (wn * WTN + r) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int co = ((2 * kk + h) ^ swz) * 16;
            uint4 a[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) a[i][q] = *reinterpret_cast<const uint4 *>(as + i * 32 * ROWB + q * PLANE_A + co);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                uint4 b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const uint4 *>(ws + j * 32 * ROWB + q * PLANE_W + co);
#define T32(QA, QB)                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                    \
        __builtin_bit_cast(f16x8, a[i][QA]), __builtin_bit_cast(f16x8, b[QB]), acc[i][j], 0, 0, 0);
                T32(0, 1)
                T32(1, 0)
                T32(0, 0)
#undef T32
            }
            if (kk == 0) store_tile(cur ^ 1, kt + 1);     // between the two halves: under the other waves' MFMAs
        }
        __syncthreads();
    }
    if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    if (!(p.rowbias || p.cm) || p.rows_per_obj >= WTM) gemm_epilogue_fast<TM, TN, WTM, WTN>(p, acc, m0, n0, z, wm, wn, r, h);
    else gemm_epilogue<TM, TN, WTM, WTN, false>(p, acc, m0, n0, z, wm, wn, r, h);
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long *o = p.stamps + 5 * (size_t)blockIdx.x;
        o[0] = st0, o[1] = st1, o[2] = st2, o[3] = __builtin_amdgcn_s_memrealtime(), o[4] = 0;
    }
}

__global__ __launch_bounds__(1024) void gemm_split32_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * (256 + 256) * 64];
    int seg = 0;
    while (seg < p.nseg - 1 && (int)blockIdx.x >= p.seg_end[seg]) ++seg;
    int L = p.seg_base[seg] + (int)blockIdx.x - (seg ? p.seg_end[seg - 1] : 0);
    if (!((p.seg_small >> seg) & 1)) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile32<256, 256, 4, 4>(p, (L / p.tiles_n_big) * 256, (L % p.tiles_n_big) * 256, z, smem);
    } else {
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile<128, 128, 4, 4, 2, true, false>(p, p.mt_big * 256 + (L / p.tiles_n_small) * 128, (L % p.tiles_n_small) * 128, z, smem);
    }
}

// (Round 3, measured and dropped: the 256 x 256 tile on 512 threads with 128 x 64 wave tiles -- a third less LDS read traffic per
// MFMA, which is the K loop's co-bottleneck: this kernel and the transposed-read TN kernel both level off near 300 TF, where LDS
// reads + writes take about as long as the MFMAs.  At two waves per SIMD a wave has 256 registers; 128 accumulators + fragments +
// the prefetch ring + the epilogue's state spill 221 of them: 30 TF.)
// The same tiles as two independent 512-thread workgroups per CU (256 x 128 outputs each, 4 x 2 waves of 64 x 64; tail:
// 128 x 128).  Two barrier domains per CU de-phase on their own: while one workgroup waits at its barrier or runs its
// epilogue the other keeps the matrix cores busy.  The price is 1.5x the operand traffic per output (A rows are shared
// by half as many columns).
template <int PD, bool F16>
__global__ __launch_bounds__(512, 4) void gemm_split512_kernel(GemmParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[(PD > 0 ? 2 : 1) * (F16 ? 2 : 3) * ((256 + 128) * 48 + 64)];
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    int seg = 0;
    while (seg < p.nseg - 1 && (int)blockIdx.x >= p.seg_end[seg]) ++seg;
    int L = p.seg_base[seg] + (int)blockIdx.x - (seg ? p.seg_end[seg - 1] : 0);
    if (!((p.seg_small >> seg) & 1)) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile<256, 128, 4, 2, PD, F16, false>(p, (L / p.tiles_n_big) * 256, (L % p.tiles_n_big) * 128, z, smem);
    } else {
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile<128, 128, 4, 2, PD, F16, false>(p, p.mt_big * 256 + (L / p.tiles_n_small) * 128, (L % p.tiles_n_small) * 128,
                                                        z, smem);
    }
}

// Few-tile launches (round 3): the HS layers' last GEMM (M x C x C with C = 128 .. 512), the decoder's 256 -> 128 layer.  On the
// 256 x 128 tiles above they are 32 .. 129 workgroups for 512 resident slots, each a prologue + 8-32 K-steps + epilogue in
// sequence: 25-40 us per launch at 3-5 % of the matrix roof (profiles/r02_g: six such launches = 180 us for 7.5 GFLOP).  Here the
// same tile template runs 64 x 128 outputs on 256 threads (2 x 2 waves of 32 x 64), four workgroups per CU (37 KB of LDS, <= 128
// registers): four times as many, four times shorter workgroups, and BN = 128 keeps the whole output row of the C = 128 layers
// in one workgroup, so the large operand (the activations) is still read once.  Same K order per output element (16-wide steps,
// three terms): results are bit-identical to the larger tiles'.
// PLOOP (round 4): the form every PREDICATED launch takes (tgp_gemm_args.pred: the fp16-range repairs, which normally return at
// once).  A repair used to be launched on the tile shape its size called for -- up to 1560 workgroups of 1024 threads and 98 KB of
// LDS each, every one of which has to become resident before it can read the flag and leave: 27 us per empty launch with four
// batches in flight, four to eight such launches per forward.  Here at most 256 workgroups of this small tile walk the launch's
// tiles (same K order per output element, so the same bits as any other tile shape); how fast a repair runs does not matter.
// (Effect on the default line, A/B of two builds on one box, four runs each: 19.57 vs 19.43 k objects/s median, ranges overlapping --
// the stalls were stream latency that the other batches in flight already covered.)
template <int PD, bool F16, bool PLANES = false, bool PLOOP = false>
__global__ __launch_bounds__(256, 4) void gemm_split256_kernel(GemmParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[(PD > 0 ? 2 : 1) * (F16 ? 2 : 3) * ((64 + 128) * 48 + 64)];
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    const int per_batch = p.tiles_m_small * p.tiles_n_small;
    if constexpr (PLOOP) {
        const int total = per_batch * p.batch;
        for (int L0 = (int)blockIdx.x; L0 < total; L0 += (int)gridDim.x) {
            const int z = L0 / per_batch, L = L0 - z * per_batch;
            gemm_split_tile<64, 128, 2, 2, PD, F16, false, 1, (PD > 0 ? 2 : 1), PLANES>(p, (L / p.tiles_n_small) * 64, (L % p.tiles_n_small) * 128, z, smem);
            __syncthreads();                 // the tile's epilogue has turned its blocks through the stages the next prologue fills
        }
    } else {
        int L = (int)blockIdx.x;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_split_tile<64, 128, 2, 2, PD, F16, false, 1, (PD > 0 ? 2 : 1), PLANES>(p, (L / p.tiles_n_small) * 64, (L % p.tiles_n_small) * 128, z, smem);
    }
}

// W (rows, K) fp32 row stride ld -> out[rows][ldo / 16][3][16] bf16 (hi, mid, lo per K-tile), zero padded to ldo
__global__ void split_bf16_kernel(const float *__restrict__ W, int rows, int K, int ld, uint16_t *__restrict__ out, int ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * ldo) return;
    const int rr = (int)(t / ldo), c = (int)(t - (int64_t)rr * ldo);
    const float x = c < K ? W[(int64_t)rr * ld + c] : 0.f;
    const __bf16 hb = (__bf16)x;
    const float r1 = x - (float)hb;
    const __bf16 mb = (__bf16)r1;
    const __bf16 lb = (__bf16)(r1 - (float)mb);
    uint16_t *o = out + ((int64_t)rr * (ldo / 16) + c / 16) * 48 + (c & 15);
    o[0] = __builtin_bit_cast(uint16_t, hb);
    o[16] = __builtin_bit_cast(uint16_t, mb);
    o[32] = __builtin_bit_cast(uint16_t, lb);
}

extern "C" int tgp_split_bf16(const float *W, int rows, int K, int ld, uint16_t *out, int ldo, tgp_stream_t stream)
{
    TGP_REQUIRE(W && out && rows > 0 && K > 0 && ld >= K && ldo >= K && (ldo & 15) == 0);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(tgp_cdiv((int64_t)rows * ldo, 256)), dim3(256), 0, tgp_hs(stream), W, rows, K,
                       ld, out, ldo);
    return TGP_LAUNCH_RESULT();
}

// W (rows, K) fp32 -> out[rows][ldo / 16][2][16] fp16 (hi, lo per K-tile), zero padded to ldo
__global__ void split_f16_kernel(const float *__restrict__ W, int rows, int K, int ld, uint16_t *__restrict__ out, int ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * ldo) return;
    const int rr = (int)(t / ldo), c = (int)(t - (int64_t)rr * ldo);
    const float x = c < K ? W[(int64_t)rr * ld + c] : 0.f;
    const _Float16 hb = (_Float16)x;
    const _Float16 lb = (_Float16)(x - (float)hb);
    uint16_t *o = out + ((int64_t)rr * (ldo / 16) + c / 16) * 32 + (c & 15);
    o[0] = __builtin_bit_cast(uint16_t, hb);
    o[16] = __builtin_bit_cast(uint16_t, lb);
}

extern "C" int tgp_split_f16(const float *W, int rows, int K, int ld, uint16_t *out, int ldo, tgp_stream_t stream)
{
    TGP_REQUIRE(W && out && rows > 0 && K > 0 && ld >= K && ldo >= K && (ldo & 15) == 0);
    hipLaunchKernelGGL(split_f16_kernel, dim3(tgp_cdiv((int64_t)rows * ldo, 256)), dim3(256), 0, tgp_hs(stream), W, rows, K,
                       ld, out, ldo);
    return TGP_LAUNCH_RESULT();
}

// Main kernel, 1024 threads (16 waves = 4 per SIMD, one workgroup per CU): blocks [0, tiles_big) take
// 256x256 tiles (N fastest, then M, then batch); the remaining blocks cover the leftover M rows with 128x128
// tiles.  M = B*1028 rarely divides into whole rounds of 256 resident workgroups (e.g. 129 x 16 = 2064 tiles
// = 8.06 rounds -> 9); handing the last M-tile rows to quarter-size tiles, dispatched last, fills the final
// round instead.
#define GEMM_BIG 256
#define GEMM_MID 128
__global__ __launch_bounds__(1024) void gemm_main_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    __shared__ __attribute__((aligned(16))) float smem[(GEMM_BIG + GEMM_BIG) * (16 + GEMM_LDPAD)];
    int L = blockIdx.x;
    if (L < p.tiles_big) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<GEMM_BIG, GEMM_BIG, 4, 4, 16, false, false, false>(p, (L / p.tiles_n_big) * GEMM_BIG,
                                                                     (L % p.tiles_n_big) * GEMM_BIG, z, smem);
    } else {
        L -= p.tiles_big;
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<GEMM_MID, GEMM_MID, 4, 4, 16, false, false, false>(p, p.mt_big * GEMM_BIG + (L / p.tiles_n_small) * GEMM_MID,
                                                                     (L % p.tiles_n_small) * GEMM_MID, z, smem);
    }
}

// 256-thread schedule (two 73.7 KB workgroups per CU): 128x128 tiles + a tail of 64x64 tiles.  Better than the
// 1024-thread schedule when a launch has only one to three rounds of tiles (finer quantisation, and the second
// resident workgroup overlaps another one's prologue/epilogue).
__global__ __launch_bounds__(256) void gemm_main256_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int L = blockIdx.x;
    if (L < p.tiles_big) {
        const int per_batch = p.mt_big * p.tiles_n_big;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<128, 128, 2, 2, 32, true, false, false>(p, (L / p.tiles_n_big) * 128, (L % p.tiles_n_big) * 128, z, smem);
    } else {
        L -= p.tiles_big;
        const int per_batch = p.tiles_m_small * p.tiles_n_small;
        const int z = L / per_batch;
        L -= z * per_batch;
        gemm_tile<64, 64, 2, 2, 32, true, false, false>(p, p.mt_big * 128 + (L / p.tiles_n_small) * 64, (L % p.tiles_n_small) * 64, z, smem);
    }
}

// Feature-space distance matrix for larger clouds: 128x128 tiles on 1024 threads (4 waves per SIMD), ascending-k
// fp32 MFMA chain (bit-identical to the CPU sgemm the reference's neighbour order depends on).
__global__ __launch_bounds__(1024) void gemm_dist_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    __shared__ __attribute__((aligned(16))) float smem[(128 + 128) * (16 + GEMM_LDPAD)];
    gemm_tile<128, 128, 4, 4, 16, false, true, true>(p, blockIdx.y * 128, blockIdx.x * 128, blockIdx.z, smem);
}

template <bool NAT, bool DIST>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_tile<64, 64, 2, 2, 32, true, NAT, DIST>(p, blockIdx.y * 64, blockIdx.x * 64, blockIdx.z, smem);
}

// ---------------------------------------------------------------------------------------------------
// M <= 32 rows (per-object vectors: the PH predictor's linears, the heads after their max-pool): weight streaming.
// A workgroup owns 32 output columns and all M rows; its 8 waves split K (wave w takes the 16-wide K-chunks
// w, w + 8, ...), each accumulating the 32 x 32 block as four v_mfma_f32_16x16x4_f32 tiles, so every W element is read
// once with 16-byte lane loads and the (<= 32, K) activation is re-read once per 32 columns (from L2).  Lane
// (c = lane % 16, q = lane / 16) loads the float4 at k = chunk + 4q of row / column c (and c + 16); MFMA sub-step s
// takes component s of every lane, i.e. k-slot q of the instruction is k = chunk + 4q + s for both operands.  The 8
// partial blocks are combined through LDS in fixed order (deterministic).
// Narrow outputs (few column blocks) use 16 columns per workgroup and 16 waves instead, to put more CUs on the stream.
#define SKINNY_UNROLL 4
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int NT, int SKINNY_WAVES>   // NT 16-column tiles per workgroup
__global__ __launch_bounds__(64 * SKINNY_WAVES) void skinny_gemm_kernel(GemmParams p)
{
    if (p.pred && *p.pred == 0) return;      // a repair launch whose condition did not arise (workgroup-uniform)
    constexpr int SKINNY_COLS = 16 * NT;
    __shared__ float red[SKINNY_WAVES][32][SKINNY_COLS + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int nb = blockIdx.x * SKINNY_COLS;
    const int z = blockIdx.y;
    const float *A = p.A + (int64_t)z * p.sA;
    const float *W = p.W + (int64_t)z * p.sW;
    // rows / columns past the end are clamped: they only feed outputs that are never stored
    const float *arow[2], *wrow[NT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = c16 + 16 * t;
        arow[t] = A + (int64_t)(row < p.M ? row : p.M - 1) * p.lda;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = nb + c16 + 16 * t;
        wrow[t] = W + (int64_t)(col < p.N ? col : p.N - 1) * p.ldw;
    }
    f32x4v acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

    const int chunks = (p.K + 15) / 16;
    for (int c0 = wave; c0 < chunks; c0 += SKINNY_WAVES * SKINNY_UNROLL) {
        float4 a[SKINNY_UNROLL][2], w[SKINNY_UNROLL][NT];
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
            const int k = (c0 + u * SKINNY_WAVES) * 16;
            const bool ok = k + 4 * q < p.K;          // K % 4 == 0; also false for chunks past the end
            // a lane whose quad lies past K reads the row's first quad instead (and discards it).  It used to read at 4 q past the
            // row start whatever K was: with K = 4 (the heads' conv4 backward, dy (32, 4)) up to 48 bytes beyond the operand's last
            // row -- a memory fault once such a tensor ended at the end of a mapped segment (round 4, the trainer step at B = 32).
            const int kc = ok ? k + 4 * q : 0;
            int ka = kc;
            if (p.a_wrap) ka = ka % p.a_wrap;                // cat((max, max), 1) without the copy (workgroup-uniform branch)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 av = *reinterpret_cast<const float4 *>(arow[t] + ka);
                if (p.a_keys)                                // max keys as the colmax epilogues left them: decoded here
                    av = make_float4(tgp_key_float(__float_as_uint(av.x)), tgp_key_float(__float_as_uint(av.y)),
                                     tgp_key_float(__float_as_uint(av.z)), tgp_key_float(__float_as_uint(av.w)));
                a[u][t] = ok ? av : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 wv = *reinterpret_cast<const float4 *>(wrow[t] + kc);
                w[u][t] = ok ? wv : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].x, w[u][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].y, w[u][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].z, w[u][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].w, w[u][j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    // accumulator element e of lane (c16, q) is C[row = 16 i + 4 q + e][col = 16 j + c16]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][16 * i + 4 * q + e][16 * j + c16] = acc[i][j][e];
    __syncthreads();
    for (int o = threadIdx.x; o < 32 * SKINNY_COLS; o += 64 * SKINNY_WAVES) {
        const int m = o / SKINNY_COLS, c = o % SKINNY_COLS;
        const int col = nb + c;
        if (m < p.M && col < p.N) {
            const int64_t vo = (int64_t)z * p.sV;
            float v = red[0][m][c];
#pragma unroll
            for (int wv = 1; wv < SKINNY_WAVES; ++wv) v += red[wv][m][c];
            v += (p.bias ? p.bias[vo + col] : 0.f);
            if (p.res1) v += p.res1[(int64_t)m * p.ldr1 + col];
            if (p.scale) v = v * p.scale[vo + col] + (p.shift ? p.shift[vo + col] : 0.f);
            if (p.act == 1) v = v > 0.f ? v : v * p.slope;
            p.C[(int64_t)z * p.sC + (int64_t)m * p.ldc + col] = v;
            if (p.Csig) p.Csig[(int64_t)m * p.ldc + col] = 1.0f / (1.0f + expf(-v));
        }
    }
}

static int resident_slots(void)
{
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = cus; // one 1024-thread workgroup per CU
    }
    return slots;
}

template <bool NAT, bool DIST>
static int launch_small(const GemmParams &p, hipStream_t stream)
{
    const size_t lds = (size_t)2 * (64 + 64) * (32 + GEMM_LDPAD) * sizeof(float);
    hipLaunchKernelGGL((gemm_small_kernel<NAT, DIST>), dim3(tgp_cdiv(p.N, 64), tgp_cdiv(p.M, 64), p.batch), dim3(256), lds,
                       stream, p);
    return TGP_LAUNCH_RESULT();
}

// Split the M-tile rows between big tiles and a tail of half-size tiles so that the last round is filled.
// Returns the estimated duration in units of one big tile; fills the schedule fields of p.
static double plan_tiles(GemmParams &p, int big, int64_t S, double tail_cost, int tail_per_idle, int big_n = 0, int small_n = 0)
{
    const int small = big / 2;
    if (!big_n) big_n = big;
    if (!small_n) small_n = small;
    const int rows_big = tgp_cdiv(p.M, big), tiles_nb = tgp_cdiv(p.N, big_n), tiles_ns = tgp_cdiv(p.N, small_n);
    auto estimate = [&](int mt) {
        const int64_t tb = (int64_t)mt * tiles_nb * p.batch;
        const int left = p.M - mt * big;
        const int64_t ts = left > 0 ? (int64_t)tgp_cdiv(left, small) * tiles_ns * p.batch : 0;
        const int64_t idle = (S - tb % S) % S;                 // slots the last big round leaves idle
        const int64_t ts_after = ts > idle * tail_per_idle ? ts - idle * tail_per_idle : 0;
        return (double)((tb + S - 1) / S) + tail_cost * (double)((ts_after + S - 1) / S);
    };
    int best_mt = rows_big; // (a) every row on big tiles
    double best = estimate(rows_big);
    const int64_t per_row = (int64_t)tiles_nb * p.batch;
    const int mt_b = (int)((((int64_t)rows_big * per_row) / S) * S / per_row); // (b) whole rows up to the last full round
    if (mt_b < rows_big && estimate(mt_b) < best) best = estimate(mt_b), best_mt = mt_b;
    if (estimate(0) < best) best = estimate(0), best_mt = 0;                   // (c) everything on half-size tiles
    p.mt_big = best_mt;
    p.tiles_n_big = tiles_nb;
    p.tiles_big = best_mt * tiles_nb * p.batch;
    const int rows_left = p.M - best_mt * big;
    p.tiles_m_small = rows_left > 0 ? tgp_cdiv(rows_left, small) : 0;
    p.tiles_n_small = tiles_ns;
    return best;
}

// Kernel-variant switches exist in development builds only (-DTGP_DEV: `python -m tgpose_amd.build --dev` writes
// libtgpose_hip_dev.so for scripts/*_ab.py); the product library has no process-global mutable state -- the variant is a
// compile-time constant there and the setters are not exported.
// low 3 bits: -1/7 = production choice (fp16: two LDS stages + two K-tiles of register prefetch; bf16: one K-tile, its
// registers allow no more), 0 single LDS stage, 1 / 2 double buffer with that many K-tiles of prefetch, 4 = 2 + skewed waves
// (fp16 only); bit 3: staggered block order
#ifdef TGP_DEV
int tgp_split_variant = 7;
unsigned long long *tgp_split_stamps = nullptr;
extern "C" void tgp_debug_set_split_variant(int v) { tgp_split_variant = v; }
extern "C" void tgp_debug_set_split_stamps(unsigned long long *buf) { tgp_split_stamps = buf; }
static int dev_env(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
#else
static constexpr int tgp_split_variant = 7;
static constexpr unsigned long long *tgp_split_stamps = nullptr;
static constexpr int dev_env(const char *) { return 0; }
#endif

// Block order of the split kernel.  Default: all big tiles, then the half-size tail tiles.  With many rounds of equal
// tiles every CU reaches its epilogue at the same moment and the chip alternates between "all CUs compute, HBM idle"
// and "all CUs store 256 KB each, matrix cores idle".  The staggered order starts a quarter of the CUs on a big tile and
// gives the others one, two or three quarter-size tiles first, which leaves four populations a quarter of a tile apart
// for the rest of the launch: store bursts of a quarter of the CUs, four times as often.
static void order_tiles(GemmParams &p, int64_t S, bool stagger)
{
    const int tiles_small = p.tiles_m_small * p.tiles_n_small * p.batch;
    const int q = (int)(S / 4);
    int nb = 0, ns = 0, n = 0, end = 0;
    auto seg = [&](int count, bool small) {
        if (count <= 0) return;
        p.seg_base[n] = small ? ns : nb;
        (small ? ns : nb) += count;
        end += count;
        p.seg_end[n] = end;
        p.seg_small |= (small ? 1 : 0) << n;
        ++n;
    };
    p.seg_small = 0;
    if (stagger && q > 0 && (S & 3) == 0 && p.tiles_big >= 4 * q && tiles_small >= 6 * q) {
        seg(q, false), seg(3 * q, true), seg(q, false), seg(2 * q, true), seg(q, false), seg(q, true);
    }
    seg(p.tiles_big - nb, false);
    seg(tiles_small - ns, true);
    p.nseg = n;
}

static bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// LDS-staged vector epilogue: every operand it touches must be addressable in aligned 16-byte pieces
static bool vec_epilogue_ok(const GemmParams &p)
{
    return !p.scalar_epi && (p.N & 3) == 0 && (p.c_col0 & 3) == 0 && (p.cm_cols & 3) == 0 && (p.sV & 3) == 0 &&
           (p.sC & 3) == 0 && (!p.C || ((p.ldc & 3) == 0 && al16(p.C))) && al16(p.bias) && al16(p.scale) && al16(p.shift) &&
           al16(p.slope_vec) && (!p.rowbias || ((p.ldrb & 3) == 0 && al16(p.rowbias))) &&
           (!p.res1 || ((p.ldr1 & 3) == 0 && al16(p.res1))) && (!p.res2 || ((p.ldr2 & 3) == 0 && al16(p.res2))) &&
           (!p.gres1 || ((p.ldg1 & 3) == 0 && al16(p.gres1))) && (!p.gres2 || ((p.ldg2 & 3) == 0 && al16(p.gres2)));
}

int tgp_launch_gemm_pp(GemmParams &p, int config, hipStream_t stream);      // gemm_pp.hip

static int launch_split(GemmParams &p, hipStream_t stream)
{
    p.vec_epi = !p.scalar_epi && (p.N & 3) == 0 && (p.c_col0 & 3) == 0 && (p.cm_cols & 3) == 0 && (p.sV & 3) == 0 &&
                (p.sC & 3) == 0 && (!p.C || ((p.ldc & 3) == 0 && al16(p.C))) && al16(p.bias) && al16(p.scale) && al16(p.shift) &&
                al16(p.slope_vec) && (!p.rowbias || ((p.ldrb & 3) == 0 && al16(p.rowbias))) &&
                (!p.res1 || ((p.ldr1 & 3) == 0 && al16(p.res1))) && (!p.res2 || ((p.ldr2 & 3) == 0 && al16(p.res2))) &&
                (!p.gres1 || ((p.ldg1 & 3) == 0 && al16(p.gres1))) && (!p.gres2 || ((p.ldg2 & 3) == 0 && al16(p.gres2)));
    // result planes (fp32 operand): the small-tile kernel's planes-writing instance, whatever the launch's size
    TGP_REQUIRE(!p.Cp || (p.vec_epi && p.split_f16 && !p.ksplit && !p.gres1 && !p.gres2 && (!(p.rowbias || p.cm) || p.rows_per_obj >= 32)));
    // gathered residuals exist in the LDS-staged epilogue only (and in the fp32 kernels' register form): refuse the rest
    TGP_REQUIRE(!(p.gres1 || p.gres2) || (p.vec_epi && !(tgp_split_variant != 7 && (tgp_split_variant & 64))));
    // few 256 x 256 tiles (N <= 256, or less than one round of them) leave CUs idle or half empty: such launches run as
    // 256 x 128 tiles on two 512-thread workgroups per CU.  Measured over the forward's 15 tile-kernel launches, each timed
    // alone: 139 us average against 154 us with square tiles only (the wide layer alone would lose: 1.04 -> 1.32 ms)
    const int64_t sq_tiles = (int64_t)tgp_cdiv(p.M, GEMM_BIG) * tgp_cdiv(p.N, GEMM_BIG) * p.batch;
    // (round 2, re-measured on the forward's shapes with the current epilogues, a since-deleted A/B script: between one and 1.5 rounds of
    // square tiles the 1024-thread form is the faster one -- M = 32896, N = 512, K = 512: 85 vs 97 us; M = 8224, N = 2304,
    // K = 256: 54 vs 58 us -- below one round the two-workgroups-per-CU form wins by 25-30 %)
    const bool narrow = p.N <= 256 || sq_tiles < (int64_t)resident_slots();
    const bool force512 = tgp_split_variant != 7 && (tgp_split_variant & 16), forbid512 = tgp_split_variant != 7 && (tgp_split_variant & 32);
    // (per launch the narrow layers are 25 % faster this way; in the whole forward, where they overlap other branches,
    // the routing measured +-0: 8769 / 8730 vs 8782 / 8881 objects/s)
    // development A/B (TGP_GATHER_512=1): epilogue-bound launches -- gathered residuals, short K -- on two workgroups per CU,
    // so that one workgroup's epilogue overlaps the other's K loop
    static const int gather512 = dev_env("TGP_GATHER_512");
    // development A/B (TGP_ROUTE512=1): every launch WITHOUT gathered residuals on the two-workgroups-per-CU form
    static const int route512 = dev_env("TGP_ROUTE512");
    const bool epi_bound = (gather512 && (p.gres1 || p.gres2) && p.K <= 512) || (route512 && !(p.gres1 || p.gres2));
    // fewer than ~0.3 rounds of the 256 x 128 tiles: the 64 x 128 form on 256 threads (gemm_split256_kernel)
    const int64_t tiles512 = (int64_t)tgp_cdiv(p.M, GEMM_BIG) * tgp_cdiv(p.N, 128) * p.batch;
    // (round 4, measured: raising this threshold so that more -- or all -- of the trainer's fp32-operand launches take the small tile,
    // as the pre-split kernel's measurements suggested, leaves the trainer's step where it is: 17.2-17.6 ms at every setting)
#ifndef TGP_PRED_BIG        // (A/B measurement build: predicated launches on the tile shape their size calls for, as before)
    if (p.pred && p.split_f16 && !p.ksplit && !p.Cp && tgp_split_variant == 7) {
        // a predicated (repair) launch: a small grid of small workgroups walking the tiles (gemm_split256_kernel, PLOOP)
        p.mt_big = 0, p.tiles_big = 0, p.tiles_n_big = 1;
        p.tiles_m_small = tgp_cdiv(p.M, 64), p.tiles_n_small = tgp_cdiv(p.N, 128);
        p.nseg = 0, p.seg_small = 0;
        p.stamps = nullptr;
        const int64_t total = (int64_t)p.tiles_m_small * p.tiles_n_small * p.batch;
        hipLaunchKernelGGL((gemm_split256_kernel<2, true, false, true>), dim3((unsigned)(total < 256 ? total : 256)), dim3(256), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
#endif
    if (p.Cp || (p.split_f16 && !p.ksplit && !p.gres1 && !p.gres2 && !force512 && tgp_split_variant == 7 &&
                 tiles512 * 10 < 3 * 2 * (int64_t)resident_slots())) {
        p.mt_big = 0, p.tiles_big = 0, p.tiles_n_big = 1;
        p.tiles_m_small = tgp_cdiv(p.M, 64), p.tiles_n_small = tgp_cdiv(p.N, 128);
        p.nseg = 0, p.seg_small = 0;
        p.stamps = nullptr;
        const dim3 grid256(p.tiles_m_small * p.tiles_n_small * p.batch);
        if (p.Cp) hipLaunchKernelGGL((gemm_split256_kernel<2, true, true>), grid256, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_split256_kernel<2, true>), grid256, dim3(256), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
    if (p.split_f16 && (force512 || epi_bound || (narrow && !forbid512))) {
        // two 512-thread workgroups per CU, 256 x 128 tiles (+ 128 x 128 tail tiles)
        plan_tiles(p, GEMM_BIG, 2 * (int64_t)resident_slots(), 0.55, 2, 128, 128);
        order_tiles(p, 2 * (int64_t)resident_slots(), false);
        p.stamps = tgp_split_stamps;
        const dim3 grid512(p.tiles_big + p.tiles_m_small * p.tiles_n_small * p.batch);
        if ((tgp_split_variant & 7) == 1) hipLaunchKernelGGL((gemm_split512_kernel<1, true>), grid512, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((gemm_split512_kernel<2, true>), grid512, dim3(512), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
    const int64_t S = resident_slots();
    plan_tiles(p, GEMM_BIG, S, 0.27, 3);
    if (p.split_f16 && tgp_split_variant != 7 && (tgp_split_variant & 64)) {
        order_tiles(p, S, false);
        p.stamps = tgp_split_stamps;
        hipLaunchKernelGGL(gemm_split32_kernel, dim3(p.tiles_big + p.tiles_m_small * p.tiles_n_small * p.batch), dim3(1024), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
    const bool stagger = (tgp_split_variant & 8) != 0 && tgp_split_variant != 7;
    if (stagger && p.tiles_big >= 3 * S) {
        // hand enough M-tile rows to the half-size tiles for the staggered start (1.5 S of them)
        const int64_t per_row = 4 * (int64_t)p.tiles_n_big * p.batch;   // half-size tiles per big M-tile row
        const int64_t have = (int64_t)p.tiles_m_small * p.tiles_n_small * p.batch;
        const int rows = (int)tgp_cdiv(3 * S / 2 + S / 4 - have > 0 ? 3 * S / 2 + S / 4 - have : 0, per_row);
        if (rows > 0 && rows < p.mt_big) {
            p.mt_big -= rows;
            p.tiles_big = p.mt_big * p.tiles_n_big * p.batch;
            p.tiles_m_small = tgp_cdiv(p.M - p.mt_big * GEMM_BIG, GEMM_BIG / 2);
        }
    }
    order_tiles(p, S, stagger);
    p.stamps = tgp_split_stamps;
    const dim3 grid(p.tiles_big + p.tiles_m_small * p.tiles_n_small * p.batch);
#define LAUNCH_SPLIT(PD, F16, SKEW) hipLaunchKernelGGL((gemm_split_kernel<PD, F16, SKEW>), grid, dim3(1024), 0, stream, p)
    if (p.split_f16) {
        switch (tgp_split_variant & 7) {
        case 0: LAUNCH_SPLIT(0, true, false); break;
        case 1: LAUNCH_SPLIT(1, true, false); break;
        case 4: LAUNCH_SPLIT(2, true, true); break;
        default: LAUNCH_SPLIT(2, true, false); break;
        }
    } else {
        switch (tgp_split_variant & 7) {
        case 0: LAUNCH_SPLIT(0, false, false); break;
        default: LAUNCH_SPLIT(1, false, false); break;
        }
    }
#undef LAUNCH_SPLIT
    return TGP_LAUNCH_RESULT();
}

static int launch_main(GemmParams &p, hipStream_t stream)
{
    const int cus = resident_slots();
    // Cost of one round of tiles in microseconds per unit of K, measured inside the forward on MI355X (rocprof,
    // profiles/r01_b and r01_c): 1024-thread schedule 0.285 (wide GEMM 2.99 ms = 8.27 rounds x 1292), 256-thread
    // schedule 0.142 (3.01 ms = 16.3 rounds x 1292; decoder conv0 421 us = 2.3 rounds).  The 1024-thread tiles win
    // when their round count is favourable (e.g. the K=128 projection GEMMs), the 256-thread ones otherwise.
    GemmParams q = p;
    const double t1024 = plan_tiles(p, GEMM_BIG, cus, 0.27, 3) * 0.285;
    const double t256 = plan_tiles(q, 128, 2 * (int64_t)cus, 0.3, 3) * 0.142;
    if (t1024 <= t256) {
        const int total = p.tiles_big + p.tiles_m_small * p.tiles_n_small * p.batch;
        hipLaunchKernelGGL(gemm_main_kernel, dim3(total), dim3(1024), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
    const int total = q.tiles_big + q.tiles_m_small * q.tiles_n_small * q.batch;
    const size_t lds = (size_t)2 * (128 + 128) * (32 + GEMM_LDPAD) * sizeof(float);
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(gemm_main256_kernel), (int)lds)) return e;
    hipLaunchKernelGGL(gemm_main256_kernel, dim3(total), dim3(256), lds, stream, q);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_gemm_f32(const tgp_gemm_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && (a->A || a->A_planes) && a->W && (a->C || a->colmax_keys || a->C_planes));
    TGP_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0);
    TGP_REQUIRE((a->K & 3) == 0 && (a->ldw & 3) == 0 && a->ldw >= a->K && (!a->A || ((a->lda & 3) == 0 && (a->lda >= a->K || (a->a_wrap > 0 && a->lda >= a->a_wrap)))));
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(a->A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->W) & 15) == 0);
    TGP_REQUIRE(a->c_col0 >= 0 && a->c_col0 <= a->N && (!a->C || a->ldc >= a->N - a->c_col0));
    TGP_REQUIRE(!(a->rowbias || a->colmax_keys) || a->rows_per_obj > 0);
    TGP_REQUIRE(a->act == 0 || a->act == 1);
    TGP_REQUIRE(a->batch >= 0 && a->cm_cols >= 0 && a->cm_cols <= a->N);
    GemmParams p = {};
    p.A = a->A, p.W = a->W, p.C = a->C;
    p.lda = a->lda, p.ldw = a->ldw, p.ldc = a->ldc, p.M = a->M, p.N = a->N, p.K = a->K;
    p.bias = a->bias, p.rowbias = a->rowbias, p.ldrb = a->ldrb, p.rows_per_obj = a->rows_per_obj > 0 ? a->rows_per_obj : 1;
    TGP_REQUIRE(a->row_base >= 0);
    p.row_base = a->row_base;
    p.res1 = a->res1, p.ldr1 = a->ldr1, p.res2 = a->res2, p.ldr2 = a->ldr2;
    p.scale = a->scale, p.shift = a->shift, p.slope_vec = a->slope_vec, p.act = a->act, p.slope = a->slope;
    p.cm = a->colmax_keys, p.ldcm = a->ldcm;
    p.cm_cols = a->cm_cols > 0 ? a->cm_cols : a->N;
    p.c_col0 = a->c_col0;
    p.batch = a->batch > 0 ? a->batch : 1;
    p.sA = a->batch_stride_a, p.sW = a->batch_stride_w, p.sC = a->batch_stride_c, p.sV = a->batch_stride_vec;
    p.sCM = a->batch_stride_colmax;
    const bool gather = a->gres1 || a->gres2;
    TGP_REQUIRE(!gather || (a->M > 32 && !a->res1 && !a->res2 && (!a->gres1 || (a->gidx1 && a->ldg1 >= a->N)) &&
                            (!a->gres2 || (a->gidx2 && a->ldg2 >= a->N)) &&
                            (!(a->rowbias || a->colmax_keys) || a->rows_per_obj >= 64) && (a->batch <= 1)));
    p.gres1 = a->gres1, p.ldg1 = a->ldg1, p.gidx1 = a->gidx1, p.gres2 = a->gres2, p.ldg2 = a->ldg2, p.gidx2 = a->gidx2;
    TGP_REQUIRE(a->epilogue == 0 || (a->epilogue == 1 && !gather));
    p.scalar_epi = a->epilogue;
    p.pred = a->pred;
    // (ABI 5) blocked fp16 planes: result planes from the LDS-staged epilogue, operand planes on the pre-split kernel
    TGP_REQUIRE(!a->A_planes == !a->W_planes);
    if (a->C_planes) {
        TGP_REQUIRE(p.batch == 1 && (a->cp_col0 & 15) == 0 && ((a->N - a->c_col0) & 15) == 0 && a->cp_col0 >= 0 &&
                    a->c_kt >= (a->cp_col0 + a->N - a->c_col0) / 16 && (reinterpret_cast<uintptr_t>(a->C_planes) & 15) == 0);
        p.Cp = reinterpret_cast<char *>(a->C_planes), p.c_kt = a->c_kt, p.cp_col0 = a->cp_col0, p.c_amax = a->c_amax;
    }
    if (a->A_planes) {
        TGP_REQUIRE((reinterpret_cast<uintptr_t>(a->A_planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->W_planes) & 15) == 0);
        TGP_REQUIRE(!a->a_scale && !a->ksplit_chunk && p.batch == 1 && a->epilogue == 0 && a->M > 32);
        p.Ap = reinterpret_cast<const char *>(a->A_planes), p.a_kt = a->a_kt, p.a_amax = a->a_amax;
        p.Wp = reinterpret_cast<const char *>(a->W_planes), p.w_kt = a->w_kt;
        p.c_scale = a->c_scale;
        p.range_flag = a->range_flag;
        TGP_REQUIRE(vec_epilogue_ok(p));
        p.vec_epi = 1;
        return tgp_launch_gemm_pp(p, a->pp_config, tgp_hs(stream));
    }
    const bool plain = !a->rowbias && !a->res2 && !a->colmax_keys && !a->slope_vec && a->c_col0 == 0 &&
                       (p.batch == 1 || !a->res1);
    const int64_t mid_tiles = (int64_t)tgp_cdiv(a->M, GEMM_MID) * tgp_cdiv(a->N, GEMM_MID) * p.batch;
    // (round 3) launches below half a round of 128 x 128 tiles used to fall to the exact-fp32 64 x 64 kernel whatever their
    // size (conv_4's last GEMM, M = 2048, N = K = 512: 41 us); with a fp16-split weight and at least 32 of the 64 x 128 tiles
    // they now take the small-tile split kernel
    const bool small_split = a->W_split && a->w_split_kind == 1 && a->M > 32 && a->N > 64 && !a->ksplit_chunk &&
                             (int64_t)tgp_cdiv(a->M, 64) * tgp_cdiv(a->N, 128) * p.batch >= 32;
    // a_scale / c_scale / ksplit_chunk are implemented by the fp16 split tile kernels only: refuse launches that route elsewhere
    TGP_REQUIRE(!(a->a_scale || a->c_scale || a->ksplit_chunk) ||
                (a->W_split && a->w_split_kind == 1 && a->M > 32 && a->N > 64 && (mid_tiles >= resident_slots() / 2 || small_split)));
    TGP_REQUIRE(!(a->a_keys || a->a_wrap || a->C_sigmoid) || (a->M <= 32 && a->C && plain && p.batch == 1));
    TGP_REQUIRE(a->a_wrap >= 0 && (a->a_wrap & 3) == 0 && (!a->a_wrap || a->lda >= a->a_wrap));
    if (a->M <= 32 && a->C && plain) {
        TGP_REQUIRE(!p.Cp);
        p.a_keys = a->a_keys, p.a_wrap = a->a_wrap, p.Csig = a->C_sigmoid;
        if ((int64_t)tgp_cdiv(a->N, 32) * p.batch >= resident_slots() / 2)
            hipLaunchKernelGGL((skinny_gemm_kernel<2, 8>), dim3(tgp_cdiv(a->N, 32), p.batch), dim3(512), 0, tgp_hs(stream), p);
        else
            hipLaunchKernelGGL((skinny_gemm_kernel<1, 16>), dim3(tgp_cdiv(a->N, 16), p.batch), dim3(1024), 0, tgp_hs(stream), p);
        return TGP_LAUNCH_RESULT();
    }
    if ((mid_tiles >= resident_slots() / 2 && a->N > 64) || small_split) {
        if (a->W_split) {
            TGP_REQUIRE(a->ldws >= ((a->K + 15) & ~15) && (a->ldws & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(a->W_split) & 15) == 0);
            TGP_REQUIRE(a->w_split_kind == 0 || a->w_split_kind == 1);
            p.Wsplit = a->W_split, p.ldws = a->ldws;
            p.plane = 0;
            p.split_f16 = a->w_split_kind;
            p.sWS = (a->w_split_kind ? 2 : 3) * (int64_t)a->N * a->ldws;
            if (a->a_scale || a->c_scale || a->ksplit_chunk) {
                // the scaled / K-split forms exist in the fp16 tile kernels only (the backward's use)
                TGP_REQUIRE(a->w_split_kind == 1);
                p.a_scale = a->a_scale, p.c_scale = a->c_scale;
                if (a->ksplit_chunk) {
                    // batch b covers the K columns [b * chunk, b * chunk + K): chunk % 16 == 0, operands zero padded so that
                    // every chunk holds K columns
                    TGP_REQUIRE(a->ksplit_chunk % 16 == 0 && a->ksplit_chunk >= a->K &&
                                (int64_t)a->ksplit_chunk * (p.batch - 1) + a->K <= a->ldws);
                    p.ksplit = 1;
                    p.sA = a->ksplit_chunk;
                    p.sWS = 2 * (int64_t)a->ksplit_chunk;
                }
            }
            return launch_split(p, tgp_hs(stream));
        }
        TGP_REQUIRE(!p.Cp);
        return launch_main(p, tgp_hs(stream));
    }
    TGP_REQUIRE(!p.Cp);
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
    if (n >= 512) {
        hipLaunchKernelGGL(gemm_dist_kernel, dim3(tgp_cdiv(n, 128), tgp_cdiv(n, 128), B), dim3(1024), 0, stream, p);
        return TGP_LAUNCH_RESULT();
    }
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

// out[b,c] = max_i x[b,i,c]; block = (b, 64 columns) x 16 row-slices, combined through LDS (round 3: 4 slices in 256 threads
// left a 1286-column buffer at B = 32 with 672 four-wave workgroups walking 257 rows each: 78 us for 170 MB)
__global__ __launch_bounds__(1024) void colmax_kernel(const float *__restrict__ x, int ld, int n, int C,
                                                      float *__restrict__ out)
{
    __shared__ float part[16][64];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    float m = -INFINITY;
    if (c < C) {
        const float *xb = x + (int64_t)b * n * ld + c;
#pragma unroll 8
        for (int i = slice; i < n; i += 16) m = fmaxf(m, xb[(int64_t)i * ld]);
    }
    part[slice][threadIdx.x & 63] = m;
    __syncthreads();
    if (slice == 0 && c < C) {
#pragma unroll
        for (int s = 1; s < 16; ++s) m = fmaxf(m, part[s][threadIdx.x]);
        out[(int64_t)b * C + c] = m;
    }
}

extern "C" int tgp_colmax(const float *x, int ld, int B, int n, int C, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(x && out && B > 0 && n > 0 && C > 0 && ld >= C);
    hipLaunchKernelGGL(colmax_kernel, dim3(tgp_cdiv(C, 64), B), dim3(1024), 0, tgp_hs(stream), x, ld, n, C, out);
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
                                 const float *__restrict__ ts, int ldg, int ldr, int ldt, const float *__restrict__ mean, int B,
                                 float *__restrict__ pg, float *__restrict__ pr, float *__restrict__ fg,
                                 float *__restrict__ fr, float *__restrict__ pT, float *__restrict__ ps)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *g = green + b * ldg, *rd = red + b * ldr;
    ts += b * ldt - b * 6;
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

extern "C" int tgp_head_post(const float *green, const float *red, const float *ts, int ldg, int ldr, int ldt, const float *mean,
                             int B, float *p_green, float *p_red, float *f_green, float *f_red, float *pred_T, float *pred_s,
                             tgp_stream_t stream)
{
    TGP_REQUIRE(green && red && ts && mean && p_green && p_red && f_green && f_red && pred_T && pred_s && B > 0);
    TGP_REQUIRE(ldg >= 4 && ldr >= 4 && ldt >= 6);
    hipLaunchKernelGGL(head_post_kernel, dim3(tgp_cdiv(B, 64)), dim3(64), 0, tgp_hs(stream), green, red, ts, ldg, ldr, ldt, mean, B,
                       p_green, p_red, f_green, f_red, pred_T, pred_s);
    return TGP_LAUNCH_RESULT();
}

// The three pose heads after their max over points, per (head, object) in ONE launch (round 4; was colmax_decode + two batched
// skinny GEMMs + head_post): the pooled keys -> conv3 (+ bias, BatchNorm fold, ReLU) -> conv4 (+ bias) -> the head's share of
// PoseNet9D.py:57-66 (PoseR.py:37-43, PoseTs.py:43-49; dropout is the identity in eval mode).  256 threads: thread o owns conv3's
// output channel o -- one fmaf chain over the 256 inputs in ascending order, reading W3 transposed (consecutive threads, consecutive
// floats) --, conv4's 8 x 256 products are split over 32 lanes per output and added by xor-shuffles in a fixed tree.
__global__ __launch_bounds__(256) void pose_tail_kernel(const uint32_t *__restrict__ keys2, const float *__restrict__ w3t,
                                                        const float *__restrict__ b3, const float *__restrict__ scale3,
                                                        const float *__restrict__ shift3, const float *__restrict__ w4,
                                                        const float *__restrict__ b4, const float *__restrict__ mean, int B,
                                                        float *__restrict__ pg, float *__restrict__ pr, float *__restrict__ fg,
                                                        float *__restrict__ fr, float *__restrict__ pT, float *__restrict__ ps,
                                                        float *__restrict__ raw)
{
    __shared__ float x[256], y[256], o[8];
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    x[t] = tgp_key_float(keys2[((int64_t)h * B + b) * 256 + t]);
    __syncthreads();
    float acc = 0.f;
    const float *w = w3t + (int64_t)h * 256 * 256 + t;
#pragma unroll 16
    for (int k = 0; k < 256; ++k) acc = fmaf(x[k], w[k * 256], acc);
    float v = acc + b3[h * 256 + t];
    v = v * scale3[h * 256 + t] + shift3[h * 256 + t];
    v = v > 0.f ? v : v * 0.f;
    y[t] = v;
    __syncthreads();
    const int j = t >> 5, l = t & 31;
    float part = 0.f;
    const float *w4r = w4 + ((int64_t)h * 8 + j) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) part = fmaf(y[l + 32 * i], w4r[l + 32 * i], part);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (l == 0) {
        const float r = part + b4[h * 8 + j];
        o[j] = r;
        if (raw) raw[((int64_t)h * B + b) * 8 + j] = r;
    }
    __syncthreads();
    if (t != 0) return;
    if (h < 2) {
        // torch.norm(v, dim=1): sqrt of the sum of squares; PoseNet9D.py:57-58 divides by (norm + 1e-6)
        const float nrm = sqrtf((o[1] * o[1] + o[2] * o[2]) + o[3] * o[3]) + 1e-6f;
        float *pv = h == 0 ? pg : pr, *fv = h == 0 ? fg : fr;
        for (int c = 0; c < 3; ++c) pv[b * 3 + c] = o[1 + c] / nrm;
        fv[b] = 1.0f / (1.0f + expf(-o[0]));
    } else {
        for (int c = 0; c < 3; ++c) {
            pT[b * 3 + c] = o[c] + mean[b * 3 + c];
            ps[b * 3 + c] = o[3 + c];
        }
    }
}

extern "C" int tgp_pose_tail(const uint32_t *keys2, const float *w3t, const float *b3, const float *scale3, const float *shift3,
                             const float *w4, const float *b4, const float *mean, int B, float *p_green, float *p_red, float *f_green,
                             float *f_red, float *pred_T, float *pred_s, float *raw, tgp_stream_t stream)
{
    TGP_REQUIRE(keys2 && w3t && b3 && scale3 && shift3 && w4 && b4 && mean && B > 0);
    TGP_REQUIRE(p_green && p_red && f_green && f_red && pred_T && pred_s);
    hipLaunchKernelGGL(pose_tail_kernel, dim3(3, B), dim3(256), 0, tgp_hs(stream), keys2, w3t, b3, scale3, shift3, w4, b4, mean, B,
                       p_green, p_red, f_green, f_red, pred_T, pred_s, raw);
    return TGP_LAUNCH_RESULT();
}

// A per-point layer with at most four outputs whose rows leave in another order (round 4; the decoder's recon_head.3,
// FaceRecon.py:117, behind the factored layers' row sort: was a 64 x 64-tile GEMM launch on N = 3 plus a scatter):
// out[obj, map[obj, i], j] = bias[j] + sum_k x[obj, i, k] W[j, k].  32 lanes per row, 16 bytes per lane and step (a row of 128
// floats is one coalesced 512-byte read), partial sums added by xor-shuffles in a fixed tree.
template <int NO>
__global__ __launch_bounds__(256) void rows_out_kernel(const float *__restrict__ x, int ld, int64_t rows, int K,
                                                       const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                       const int64_t *__restrict__ map, int rows_per_obj, float *__restrict__ out,
                                                       const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;              // a repair launch whose condition did not arise
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int64_t rr = row < rows ? row : rows - 1;            // (whole waves reach the shuffles)
    float acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) acc[j] = 0.f;
    for (int k4 = l; k4 < K / 4; k4 += 32) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + rr * ld + 4 * k4);
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const float4 wv = *reinterpret_cast<const float4 *>(W + (int64_t)j * ldw + 4 * k4);
            acc[j] = fmaf(xv.w, wv.w, fmaf(xv.z, wv.z, fmaf(xv.y, wv.y, fmaf(xv.x, wv.x, acc[j]))));
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 64);
    if (l != 0 || row >= rows) return;
    const int64_t dst = map ? (row / rows_per_obj) * rows_per_obj + map[row] : row;
#pragma unroll
    for (int j = 0; j < NO; ++j) out[dst * NO + j] = acc[j] + (bias ? bias[j] : 0.f);
}

extern "C" int tgp_rows_out_pred(const float *x, int ld, int64_t rows, int K, const float *W, int ldw, const float *bias, int n_out,
                                 const int64_t *map, int rows_per_obj, float *out, const int *pred, tgp_stream_t stream);
extern "C" int tgp_rows_out(const float *x, int ld, int64_t rows, int K, const float *W, int ldw, const float *bias, int n_out,
                            const int64_t *map, int rows_per_obj, float *out, tgp_stream_t stream)
{
    return tgp_rows_out_pred(x, ld, rows, K, W, ldw, bias, n_out, map, rows_per_obj, out, nullptr, stream);
}

extern "C" int tgp_rows_out_pred(const float *x, int ld, int64_t rows, int K, const float *W, int ldw, const float *bias, int n_out,
                                 const int64_t *map, int rows_per_obj, float *out, const int *pred, tgp_stream_t stream)
{
    TGP_REQUIRE(x && W && out && rows > 0 && K > 0 && (K & 3) == 0 && ld >= K && ldw >= K && (ld & 3) == 0 && (ldw & 3) == 0);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0);
    TGP_REQUIRE(!map || (rows_per_obj > 0 && rows % rows_per_obj == 0));
    if (n_out < 1 || n_out > 4) return TGP_EUNSUPPORTED;
    const dim3 grid((unsigned)tgp_cdiv(rows, 8));
#define RO_GO(NO) hipLaunchKernelGGL(rows_out_kernel<NO>, grid, dim3(256), 0, tgp_hs(stream), x, ld, rows, K, W, ldw, bias, map, rows_per_obj, out, pred)
    if (n_out == 1) RO_GO(1); else if (n_out == 2) RO_GO(2); else if (n_out == 3) RO_GO(3); else RO_GO(4);
#undef RO_GO
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
