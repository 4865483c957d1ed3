// fp32-accurate GEMM on the fp16 matrix cores with BOTH operands pre-split (round 4).
//
// gemm.hip's split kernels take the activations as fp32 and split them into fp16 hi / lo planes while they are staged into LDS:
// ~45 vector instructions per wave and K-step beside 12 MFMAs, through a ring of prefetch registers that pins the kernel at its
// 128-VGPR cap.  Here the producer of an activation (a GEMM epilogue, the row gather, tgp_planes_split) has already written the
// two planes -- a deterministic function of the fp32 value, so every product below equals the one the in-loop split feeds the
// matrix cores -- in a BLOCKED layout made for LDS-DMA:
//
//     chunk (rb = row / 32, kt = k / 16), 2 KB:  [plane: hi | lo][h = (k % 16) / 8][r = row % 32][8 fp16]
//
// One plane of a chunk is 1 KB, contiguous in memory, and in exactly the order in which the 64 lanes of a
// v_mfma_f32_32x32x16_f16 hold it (lane = 32 h + r supplies row r, k = 8 h .. 8 h + 7): one global_load_lds_dwordx4 per wave moves
// it into LDS as it lies, and the fragment read is ds_read_b128 at base + 16 * lane -- linear, conflict-free, no padding, no
// swizzle.  No VGPR staging, no conversion in the loop: the K loop is DMA issue, fragment reads and MFMAs.
//
// Arithmetic: per 32 x 32 output block, K-tiles ascending, per K-tile the three terms A_hi W_lo, A_lo W_hi, A_hi W_hi -- the
// sequence of gemm_split_tile<..., F16> -- so results are bit-identical to that kernel whatever the tile shape
// (tests/test_gpu_parity.py::test_gemm_pp_bit_identical_to_split_kernel).
#include "gemm_epi.h"

#define PP_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(0x0f70 | ((N) & 15) | (((N) >> 4) << 14))

#ifdef TGP_DEV   // development build: wall-clock stamps (100 MHz) per workgroup -- entry, first stage landed, K loop done, epilogue done
__device__ unsigned long long *tgp_pp_stamps = nullptr;
extern "C" int tgp_debug_set_pp_stamps(void *buf)
{
    unsigned long long *b = reinterpret_cast<unsigned long long *>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tgp_pp_stamps), &b, sizeof(b));
}
// (stamps 0 .. 3: wall clock; the shader clock at stamps 0 and 3 goes to slots 4 and 5 of a second table behind the first:
// shader cycles per microsecond of the workgroup's lifetime = the clock the kernel actually ran at)
#define PP_STAMP(I)                                                                                   \
    if (tgp_pp_stamps && threadIdx.x == 0) {                                                          \
        tgp_pp_stamps[(size_t)blockIdx.x * 4 + (I)] = wall_clock64();                                 \
        if ((I) == 0 || (I) == 3) tgp_pp_stamps[(size_t)(1 << 18) + (size_t)blockIdx.x * 2 + ((I) == 3)] = __builtin_readcyclecounter(); \
    }
// timing-only knobs (results are garbage): 1 = A fragments read once, 2 = W fragments read once, 4 = no LDS-DMA after the prologue,
// 8 = no fragment reads and no MFMAs (the staging pipeline alone)
__device__ int tgp_pp_knobs = 0;
extern "C" int tgp_debug_set_pp_knobs(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tgp_pp_knobs), &v, sizeof(v)); }
#define PP_KNOB(B) (tgp_pp_knobs & (B))
#else
#define PP_STAMP(I)
#define PP_KNOB(B) 0
#endif

template <int BM, int BN, int NWM, int NWN, int KTS, int STAGES>
__device__ __forceinline__ void gemm_pp_tile(const GemmParams &p, const int m0, const int n0, char *smem)
{
    constexpr int NW = NWM * NWN;
    constexpr int WTM = BM / NWM, WTN = BN / NWN, TM = WTM / 32, TN = WTN / 32;
    constexpr int ABLK = BM / 32, WBLK = BN / 32;
    constexpr int NCH = (ABLK + WBLK) * KTS * 2;           // 1 KB pieces per stage: [A blocks | W blocks][K-tile of the step][plane]
    constexpr int NI = NCH / NW;                           // LDS-DMA instructions per wave and step
    constexpr int STAGE_BYTES = NCH * 1024;
    constexpr int D = STAGES - 1;                          // steps of prefetch
    static_assert(NCH % NW == 0 && STAGES >= 2 && STAGES <= 5 && (STAGES - 2) * NI <= 63, "piece / wave mapping");
    static_assert(STAGES * STAGE_BYTES >= NW * 4096, "the epilogue turns blocks through 4 KB per wave");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    const int KT = (p.K + 15) >> 4;
    const int numS = (KT + KTS - 1) / KTS;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fp16 range guard, decided per tile from what the PRODUCER of the planes recorded (bits of max |a| per 32-row block): a
    // tile holding a magnitude >= 65504 (or a NaN), or nothing at or above 2^-4, is computed in exact fp32 from the fp32 copy
    // of the operand -- the same rule, tile by tile, as gemm_split_tile's in-loop guard.
    bool exact = false;
    if (p.a_amax) {
        uint32_t am = 0u;
        const int nblk = (p.M + 31) >> 5;
#pragma unroll
        for (int i = 0; i < ABLK; ++i) {
            const int rb = (m0 >> 5) + i;
            const uint32_t v = rb < nblk ? p.a_amax[rb] : 0u;
            am = v > am ? v : am;
        }
        exact = am >= 0x477fe000u || (am != 0u && am < 0x3d800000u);       // >= 65504 | all below 2^-4 (and not all zero)
        if (exact && !p.A) {
            // a planes-only operand: nothing to recompute from.  The tile runs on the split (its result is wrong or imprecise) and
            // the flag tells the caller's predicated fp32 chain to redo the layer(s)
            if (threadIdx.x == 0) atomicOr(p.range_flag, 1);
            exact = false;
        }
    }
    if (!exact) {
        // ---- staging: piece c = j * NW + wave of a stage is this wave's j-th instruction
        const int a_blocks = (p.M + 31) >> 5, w_blocks = (p.N + 31) >> 5;
        const char *cbase[NI];
        int ct[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int c = j * NW + wave, plane = c & 1, t = (c >> 1) % KTS, ob = (c >> 1) / KTS;
            const char *base;
            if (ob < ABLK) {
                const int rb = min((m0 >> 5) + ob, a_blocks - 1);          // blocks past the end re-read the last one (never stored)
                base = p.Ap + (int64_t)rb * p.a_kt * 2048;
            } else {
                const int nb = min((n0 >> 5) + ob - ABLK, w_blocks - 1);
                base = p.Wp + (int64_t)nb * p.w_kt * 2048;
            }
            cbase[j] = base + plane * 1024 + lane * 16;
            ct[j] = t;
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * 1024);
        auto dma = [&](const int s, const int stage) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int kt = min(s * KTS + ct[j], KT - 1);               // a K-tile past the end: any valid piece (its MFMAs are skipped)
                const char *src = cbase[j] + (int64_t)kt * 2048;
                const uint32_t lds = lds0 + stage * STAGE_BYTES + j * NW * 1024;
                // inline assembly: opaque to the compiler's counters, so no vmcnt(0) appears before the fragment reads that
                // follow; the waits are written by hand below (as in heads_fused.hip)
                asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "{m0}"(lds) : "memory");
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < numS) dma(d, d);
#ifdef TGP_DEV
        uint4 keep_a[TM][2] = {}, keep_b[TN][2] = {};
#endif
        for (int s = 0; s < numS; ++s) {
            // step s's pieces have landed (this wave's; after the barrier everybody's), and everybody has finished reading the
            // stage that step s + D is about to overwrite (it held step s - 1)
            // (still in flight behind step s's pieces: the steps issued after it, min(D - 1, numS - 1 - s) of them)
            const int ahead = min(D - 1, numS - 1 - s);
            if (ahead <= 0) PP_WAIT_VM(0);
            else if (ahead == 1) PP_WAIT_VM(NI);
            else if (ahead == 2) PP_WAIT_VM(2 * NI);
            else PP_WAIT_VM(3 * NI);
            __builtin_amdgcn_s_barrier();
            if (s == 0) { PP_STAMP(1) }
            if (s + D < numS && !PP_KNOB(4)) dma(s + D, (s + D) % STAGES);
            const char *st = smem + (s % STAGES) * STAGE_BYTES + lane * 16;
#pragma unroll
            for (int t = 0; t < KTS; ++t) {
                if (KTS > 1 && s * KTS + t >= KT) break;                   // workgroup-uniform
                if (PP_KNOB(8)) break;                                     // (timing only: the staging pipeline alone)
                uint4 a[TM][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                    {
#ifdef TGP_DEV
                        if (PP_KNOB(1) && s > 0) { a[i][q] = keep_a[i][q]; continue; }
#endif
                        a[i][q] = *reinterpret_cast<const uint4 *>(st + (((wm * TM + i) * KTS + t) * 2 + q) * 1024);
#ifdef TGP_DEV
                        keep_a[i][q] = a[i][q];
#endif
                    }
                // The step's W fragments: all of them before the first MFMA where the registers allow (the small tile: +8), so that a
                // step waits for LDS once.  Loaded column by column into ONE pair of registers -- what the larger wave tiles still do
                // -- every column's reads sit behind the previous column's MFMAs with an lgkmcnt(0) in between: four LDS round
                // trips per step of six MFMAs on the 64 x 128 tile.
                constexpr bool ALLB = TM * TN <= 2;
                uint4 bb[ALLB ? TN : 1][2];
                auto load_b = [&](const int j, uint4 (&b)[2]) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                    {
#ifdef TGP_DEV
                        if (PP_KNOB(2) && s > 0) { b[q] = keep_b[j][q]; continue; }
#endif
                        b[q] = *reinterpret_cast<const uint4 *>(st + (((ABLK + wn * TN + j) * KTS + t) * 2 + q) * 1024);
#ifdef TGP_DEV
                        keep_b[j][q] = b[q];
#endif
                    }
                };
                if constexpr (ALLB) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) load_b(j, bb[j]);
                    __builtin_amdgcn_sched_barrier(0);                     // keep the reads up here: the scheduler sinks them to their uses
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (!ALLB) load_b(j, bb[0]);
                    const uint4 (&b)[2] = bb[ALLB ? j : 0];
                    // smallest terms first, as in gemm_split_tile: hi x lo, lo x hi, hi x hi; consecutive MFMAs walk the column's accumulators
#define PP_TERM(QA, QB)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                     \
        __builtin_bit_cast(f16x8, a[i][QA]), __builtin_bit_cast(f16x8, b[QB]), acc[i][j], 0, 0, 0);
                    PP_TERM(0, 1)
                    PP_TERM(1, 0)
                    PP_TERM(0, 0)
#undef PP_TERM
                }
            }
        }
        __builtin_amdgcn_s_barrier();                                      // every fragment read is done: the stages become epilogue scratch
        PP_STAMP(2)
    } else {
        // exact fp32 recomputation straight from global memory (v_mfma_f32_32x32x2_f32; lane (r, h) supplies k = 8 t + 4 h + s to
        // sub-step s of the 8-wide group t), same accumulator layout: gemm_split_tile's fallback
        const float *ag[TM], *wg[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * WTM + i * 32 + r;
            ag[i] = p.A + (int64_t)(row < p.M ? row : p.M - 1) * p.lda + 4 * h;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + r;
            wg[j] = p.W + (int64_t)(col < p.N ? col : p.N - 1) * p.ldw + 4 * h;
        }
#pragma unroll 1
        for (int k0 = 0; k0 < p.K; k0 += 8) {
            const bool ok = k0 + 4 * h < p.K;
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
    if (p.c_scale && !exact) {                     // a pre-scaled weight (ops.split_w); the exact path used the unscaled operands
        const float cs = p.c_scale[0];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= cs;
    }
    // (one instance whose planes output is a run-time branch: two inlined instances spilled 73 registers at the 128-register shapes)
    gemm_epilogue_lds<TM, TN, WTM, WTN, true>(p, acc, m0, n0, 0, wm, wn, r, h, reinterpret_cast<float *>(smem) + wave * 1024);
    PP_STAMP(3)
}

// Workgroups are dealt round-robin to the 8 XCDs; the remap gives each XCD a contiguous range of tiles (N fastest), so the tiles
// that share A rows meet in one L2 (bijective for any tile count).
__device__ __forceinline__ int pp_xcd_remap(const int b, const int n)
{
    const int q = n >> 3, rr = n & 7, x = b & 7;
    return (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (b >> 3);
}

template <int BM, int BN, int NWM, int NWN, int KTS, int STAGES, int WPE>
__global__ __launch_bounds__(64 * NWM * NWN, WPE) void gemm_pp_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(1024))) char pp_smem[];
    if (p.pred && *p.pred == 0) return;
    PP_STAMP(0)
    const int L = pp_xcd_remap((int)blockIdx.x, p.pp_tiles_m * p.pp_tiles_n);
    gemm_pp_tile<BM, BN, NWM, NWN, KTS, STAGES>(p, (L / p.pp_tiles_n) * BM, (L % p.pp_tiles_n) * BN, pp_smem);
}

// ---- fp32 rows -> blocked planes (weights at pack time; activations whose producer does not write planes itself), optionally
// through a row gather (dst[b][p] = src[b][idx[b][p]], the factored layers' row order: engine.encoder_forward) that also leaves
// the fp32 copy -- one pass over the rows instead of a gather and a split
__global__ __launch_bounds__(256) void planes_split_kernel(const float *__restrict__ X, int rows, int K, int ld, char *__restrict__ out,
                                                           int kts, uint32_t *__restrict__ amax, const int32_t *__restrict__ idx,
                                                           int n_src, int n_out, float *__restrict__ dst, int ldd, int ccopy, int kt0,
                                                           int nkt)
{
    // thread = (row block rb, K-tile kt, half h, row r): one 16-byte piece of each plane; a wave covers both halves of one (rb, kt).
    // nkt K-tiles are written per row block, at tiles kt0 .. kt0 + nkt - 1 of its kts (a column range of a wider planes buffer).
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nblk = (rows + 31) >> 5;
    if (t >= (int64_t)nblk * nkt * 64) return;              // (whole waves: the grid is a multiple of 64 threads)
    const int r = (int)(t & 31), h = (int)((t >> 5) & 1);
    const int kt = (int)((t >> 6) % nkt), rb = (int)((t >> 6) / nkt);
    const int row = rb * 32 + r, k0 = kt * 16 + h * 8;
    int64_t srow = row;
    if (idx && row < rows) srow = (int64_t)(row / n_out) * n_src + idx[row];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (row < rows && k0 + e < ld) ? X[srow * ld + k0 + e] : 0.f;
    if (dst && row < rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < ccopy) dst[(int64_t)row * ldd + k0 + e] = v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = k0 + e < K ? v[e] : 0.f;          // columns K .. ld - 1 are not the caller's to define
    uint2 h0, l0, h1, l1;
    split2(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    split2(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    char *o = out + ((int64_t)rb * kts + kt0 + kt) * 2048 + h * 512 + r * 16;
    *reinterpret_cast<uint4 *>(o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(o + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    if (amax) {
        uint32_t m = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
            m = b > m ? b : m;
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o2 = (uint32_t)__shfl_xor((int)m, s);
            m = o2 > m ? o2 : m;
        }
        if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + rb, m);
    }
}

// The same through LDS, one 256-thread workgroup per block of 32 rows: the rows are read along their length (float4 per lane,
// coalesced; the fp32 copy of a gather is written the same way), transposed through a 32 x (16 nkt + 4)-float tile -- the 4-float pad
// makes the row stride 16 x odd bytes mod 256, so the 16-lane groups of the ds_read_b128 below meet 16 distinct bank quads -- and
// leave as 512-byte runs of 16-byte plane pieces.  The direct kernel above reads 32 B per lane from 32 different rows per wave
// instruction: 53 us for the sorted fine buffer (36 MB in, 72 MB out) against 20 here.
#define PLANES_ROWS_MAX_KT 64
__global__ __launch_bounds__(256) void planes_rows_kernel(const float *__restrict__ X, int rows, int K, int ld, int C,
                                                          char *__restrict__ out, int kts, uint32_t *__restrict__ amax,
                                                          const int32_t *__restrict__ idx, int n_src, int n_out,
                                                          float *__restrict__ dst, int ldd, int kt0, int nkt)
{
    extern __shared__ __attribute__((aligned(16))) float pr_tile[];
    __shared__ uint32_t pr_amax[4];
    const int rb = blockIdx.x, tid = threadIdx.x;
    const int LDP = nkt * 16 + 4, W4 = nkt * 4;                 // tile row stride in floats; float4 per tile row
    const int C4 = C >> 2;                                      // float4 per source row (C % 4 == 0)
    for (int i = tid; i < 32 * W4; i += 256) {
        const int r = i / W4, c4 = i - r * W4;
        const int row = rb * 32 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && c4 < C4) {
            const int64_t srow = idx ? (int64_t)(row / n_out) * n_src + idx[row] : row;
            v = *reinterpret_cast<const float4 *>(X + srow * ld + 4 * c4);
            if (dst) *reinterpret_cast<float4 *>(dst + (int64_t)row * ldd + 4 * c4) = v;
            const int k = 4 * c4;                                // columns K .. C - 1 travel in the fp32 copy only
            v.x = k + 0 < K ? v.x : 0.f, v.y = k + 1 < K ? v.y : 0.f, v.z = k + 2 < K ? v.z : 0.f, v.w = k + 3 < K ? v.w : 0.f;
        }
        *reinterpret_cast<float4 *>(pr_tile + r * LDP + 4 * c4) = v;
    }
    __syncthreads();
    uint32_t m = 0u;
    for (int q = tid; q < nkt * 64; q += 256) {
        const int r = q & 31, h = (q >> 5) & 1, kt = q >> 6;
        const float4 v0 = *reinterpret_cast<const float4 *>(pr_tile + r * LDP + kt * 16 + h * 8);
        const float4 v1 = *reinterpret_cast<const float4 *>(pr_tile + r * LDP + kt * 16 + h * 8 + 4);
        uint2 h0, l0, h1, l1;
        split2(v0, h0, l0);
        split2(v1, h1, l1);
        char *o = out + ((int64_t)rb * kts + kt0 + kt) * 2048 + h * 512 + r * 16;
        *reinterpret_cast<uint4 *>(o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4 *>(o + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        const uint32_t b[8] = {__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v0.z), __float_as_uint(v0.w),
                               __float_as_uint(v1.x), __float_as_uint(v1.y), __float_as_uint(v1.z), __float_as_uint(v1.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) m = (b[e] & 0x7fffffffu) > m ? (b[e] & 0x7fffffffu) : m;
    }
    if (amax) {
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o2 = (uint32_t)__shfl_xor((int)m, s);
            m = o2 > m ? o2 : m;
        }
        if ((tid & 63) == 0) pr_amax[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const uint32_t a = max(max(pr_amax[0], pr_amax[1]), max(pr_amax[2], pr_amax[3]));
            if (a) atomicMax(amax + rb, a);
        }
    }
}

// rows -> planes: the LDS form when the rows are float4-addressable and at most 1024 columns wide, else the direct form
static int planes_launch(const float *X, int rows, int K, int ld, int C, char *out, int kts, uint32_t *amax, const int32_t *idx, int n_src,
                         int n_out, float *dst, int ldd, int kt0, int nkt, hipStream_t stream)
{
    const int nblk = (rows + 31) / 32;
    const bool vec = nkt <= PLANES_ROWS_MAX_KT && (ld & 3) == 0 && (C & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                     (!dst || ((ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0)) && C <= nkt * 16;
    if (vec) {
        const size_t lds = (size_t)32 * (nkt * 16 + 4) * sizeof(float);
        static TgpLdsAttr attr;
        if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(planes_rows_kernel), 32 * (PLANES_ROWS_MAX_KT * 16 + 4) * 4)) return e;
        hipLaunchKernelGGL(planes_rows_kernel, dim3(nblk), dim3(256), lds, stream, X, rows, K, ld, C, out, kts, amax, idx, n_src, n_out, dst,
                           ldd, kt0, nkt);
        return TGP_LAUNCH_RESULT();
    }
    const int64_t threads = (int64_t)nblk * nkt * 64;
    hipLaunchKernelGGL(planes_split_kernel, dim3(tgp_cdiv(threads, 256)), dim3(256), 0, stream, X, rows, K, ld, out, kts, amax, idx, n_src,
                       n_out, dst, ldd, C, kt0, nkt);
    return TGP_LAUNCH_RESULT();
}

extern "C" int64_t tgp_planes_bytes(int64_t rows, int K)
{
    if (rows <= 0 || K <= 0) return 0;
    return ((rows + 31) / 32) * ((K + 15) / 16) * 2048;
}

extern "C" int tgp_planes_split(const float *X, int rows, int K, int ld, void *out, int kts, uint32_t *amax, tgp_stream_t stream)
{
    TGP_REQUIRE(X && out && rows > 0 && K > 0 && ld >= K && kts >= (K + 15) / 16);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // (the direct form reads columns < ld; the LDS form takes the K columns, rounded up to whole float4 when the row stride allows)
    const int C = (ld & 3) == 0 ? min((K + 3) & ~3, ld) : K;
    return planes_launch(X, rows, K, ld, C, reinterpret_cast<char *>(out), kts, amax, nullptr, 0, 0, nullptr, 0, 0, kts, tgp_hs(stream));
}

extern "C" int tgp_planes_split_cols(const float *X, int rows, int K, int ld, void *out, int kts, int col0, uint32_t *amax,
                                     tgp_stream_t stream)
{
    TGP_REQUIRE(X && out && rows > 0 && K > 0 && ld >= K && col0 >= 0 && (col0 & 15) == 0 && kts >= col0 / 16 + (K + 15) / 16);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int nkt = (K + 15) / 16;
    const int C = (ld & 3) == 0 ? min((K + 3) & ~3, ld) : K;
    return planes_launch(X, rows, K, ld, C, reinterpret_cast<char *>(out), kts, amax, nullptr, 0, 0, nullptr, 0, col0 / 16, nkt, tgp_hs(stream));
}

extern "C" int tgp_planes_gather(const float *src, int lds, const int32_t *idx, int B, int n_src, int n_out, int K, int C, float *dst,
                                 int ldd, void *out, int kts, uint32_t *amax, tgp_stream_t stream)
{
    TGP_REQUIRE(src && idx && out && B > 0 && n_src > 0 && n_out > 0 && K > 0 && C >= K && lds >= C && kts >= (K + 15) / 16);
    TGP_REQUIRE(!dst || ldd >= C);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (int64_t)B * n_out < (1ll << 31));
    const int rows = B * n_out;
    // (columns [K, C) travel in the fp32 copy only; the K-tiles must cover them so that every column of the copy is visited)
    TGP_REQUIRE(!dst || kts * 16 >= C);
    return planes_launch(src, rows, K, lds, C, reinterpret_cast<char *>(out), kts, amax, idx, n_src, n_out, dst, ldd, 0, kts, tgp_hs(stream));
}

// ---- launch
// config: 1 = 256 x 256 on 8 waves (128 x 64 wave tiles), 32-wide steps, 2 stages (128 KB): one workgroup per CU
//         2 = 256 x 256 on 16 waves (64 x 64), 32-wide steps, 2 stages
//         3 = 256 x 128 on 8 waves (64 x 64), 16-wide steps, 3 stages (72 KB): two workgroups per CU
//         4 = 128 x 128 on 4 waves (64 x 64), 16-wide steps, 2 stages (32 KB): four workgroups per CU
//         5 = 64 x 128 on 4 waves (32 x 64), 16-wide steps, 2 stages (24 KB): the few-tile launches
//         6 = 128 x 128 on 4 waves, 32-wide steps, 2 stages (64 KB): two workgroups per CU
//         7 = 128 x 256 on 8 waves (64 x 64), 16-wide steps, 3 stages (72 KB): two workgroups per CU
template <int BM, int BN, int NWM, int NWN, int KTS, int STAGES, int WPE>
static int pp_launch(GemmParams &p, hipStream_t stream)
{
    constexpr int lds = STAGES * ((BM + BN) / 32) * KTS * 2 * 1024;
    static TgpLdsAttr attr;                    // (one per instance of this template)
    if (lds > 64 * 1024)
        if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(gemm_pp_kernel<BM, BN, NWM, NWN, KTS, STAGES, WPE>), lds)) return e;
    p.pp_tiles_m = tgp_cdiv(p.M, BM), p.pp_tiles_n = tgp_cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_pp_kernel<BM, BN, NWM, NWN, KTS, STAGES, WPE>), dim3(p.pp_tiles_m * p.pp_tiles_n), dim3(64 * NWM * NWN),
                       lds, stream, p);
    return TGP_LAUNCH_RESULT();
}

// the library's choice of tile shape for a launch (0 in tgp_gemm_args.pp_config).  Measured on the eval forward's launches, every
// shape against every launch (scripts/gemm_pp_ab.py, profiles/r04_a_gemm_pp_ab.txt): the 256 x 256 tiles -- one workgroup per CU --
// lose everywhere on this network's short-K, epilogue-heavy layers (K = 128 .. 512: a tile is prologue + 4-16 steps + a 256 KB
// store burst, nothing overlaps it); four 128 x 128 workgroups per CU overlap one another's phases and quantise the 128.5 row tiles
// of 32896 rows finely; narrow outputs (N <= 512) do best on 64 x 128; only the long, wide level-1 coarse product (K = 512, 38 M
// outputs) is better on 256 x 128 with three stages.
static int pp_auto_config(const GemmParams &p)
{
    // N <= 512: the 64 x 128 tile; with more tiles than four per CU hold, its 96-register build (five per CU: decoder's first
    // layer 65.7 -> 59.1 us, its 512-wide layers 2-4 %; the single-round launches of the encoder are 1-6 % slower on it)
    if (p.N <= 512) return (int64_t)((p.M + 63) / 64) * ((p.N + 127) / 128) > 1024 ? 8 : 5;
    if (p.K >= 512 && (int64_t)p.M * p.N >= (1ll << 25)) return 3;
    return 4;
}

int tgp_launch_gemm_pp(GemmParams &p, int config, hipStream_t stream)
{
    TGP_REQUIRE(p.Ap && p.Wp && p.batch == 1 && !p.ksplit && !p.a_scale);
    TGP_REQUIRE(p.a_kt >= (p.K + 15) / 16 && p.w_kt >= (p.K + 15) / 16);
    TGP_REQUIRE(!p.a_amax || (p.A && p.W) || (!p.A && p.range_flag));    // the guard's exact path reads the fp32 operands
    if (!config) config = pp_auto_config(p);
    // the epilogue keeps two objects per wave tile: its per-object bias / max over points need rows_per_obj >= the wave tile's rows
    if (p.rowbias || p.cm) {
        TGP_REQUIRE(p.rows_per_obj >= 32);
        if (p.rows_per_obj < 64 && config != 8) config = 5;
        else if (p.rows_per_obj < 128 && config == 1) config = 4;
    }
    switch (config) {
    case 1: return pp_launch<256, 256, 2, 4, 2, 2, 2>(p, stream);
    case 2: return pp_launch<256, 256, 4, 4, 2, 2, 4>(p, stream);
    case 3: return pp_launch<256, 128, 4, 2, 1, 3, 4>(p, stream);
    case 4: return pp_launch<128, 128, 2, 2, 1, 2, 4>(p, stream);
    case 5: return pp_launch<64, 128, 2, 2, 1, 2, 4>(p, stream);
    case 6: return pp_launch<128, 128, 2, 2, 2, 2, 2>(p, stream);
    case 7: return pp_launch<128, 256, 2, 4, 1, 3, 4>(p, stream);
    case 8: return pp_launch<64, 128, 2, 2, 1, 2, 5>(p, stream);     // config 5 held to 96 registers: five workgroups per CU
    // (measured and removed, round 4: <64, 128, 2, 2, 1, 3, 4> -- three stages at four workgroups per CU -- and <192, 128, 2, 2, 1, 2, 3>
    // -- 688 workgroups for 768 slots, i.e. the decoder's 32896 rows without a last partial round: both within 2 % of configs 4 / 5 / 8
    // on every launch of the forward (profiles/r04_g_gemm_pp_knobs.txt has the knob runs that go with it))
    default: return TGP_EINVAL;
    }
}
