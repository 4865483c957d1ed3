// GemmParams and the fused epilogues shared by every tile GEMM kernel (gemm.hip, gemm_pp.hip).
#pragma once
#include "tgp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(const float4 v, uint2 &hi, uint2 &lo)
{
    // vector form so that the packed conversions of gfx950 (v_cvt_pk_f16_f32) and v_pk_add_f32 are used
    const f32x4 x = {v.x, v.y, v.z, v.w};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    const f32x4 rest = x - __builtin_convertvector(h, f32x4);
    const f16x4 l = __builtin_convertvector(rest, f16x4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

struct GemmParams {
    const float *A;
    const float *W;
    float *C;
    int lda, ldw, ldc, M, N, K;
    const float *bias;
    const float *rowbias;
    int ldrb, rows_per_obj;
    int row_base;    // row r of this launch is row r + row_base of the batch for the per-object bias / max (tgp_gemm_args.row_base)
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
    const uint16_t *Wsplit;      // bf16 (hi, mid, lo) of W as [N][ldws / 16][3][16], ldws % 16 == 0, zero padded
    int ldws;
    int split_f16;               // 0: three bf16 planes, 1: two fp16 planes
    // block order of the split kernel: segment g covers blocks [seg_end[g-1], seg_end[g]) and maps them to consecutive
    // tiles of one kind starting at seg_base[g] (bit g of seg_small: half-size tiles)
    int nseg, seg_small;
    int seg_end[8], seg_base[8];
    unsigned long long *stamps;  // development only: per block {start, loop start, loop end, end} of s_memrealtime (100 MHz)
    int64_t sWS;                 // per-batch stride of Wsplit in bf16 elements of one plane
    // backward GEMMs on the fp16 split: A is multiplied by *a_scale (a power of two that lifts gradients into fp16's range)
    // before it is split, the accumulators by *c_scale before the epilogue; ksplit: the batch index walks K-chunks of one
    // problem (sA, sWS are K offsets), so the W descriptor's extent shrinks by the chunk offset
    const float *a_scale, *c_scale;
    int ksplit;
    // gathered residuals: v += gres1[gidx1[row]][col] + gres2[gidx2[row]][col] (the factored wide layers: products of the
    // coarse levels' features, computed once per coarse point and fetched by each point's nearest coarse point)
    const float *gres1, *gres2;
    const int32_t *gidx1, *gidx2;
    int ldg1, ldg2;
    int a_keys, a_wrap;          // skinny kernel: A holds max keys (decoded on load), column k of the operand is key k % a_wrap
    float *Csig;                 // skinny kernel: sigmoid of the stored values
    int vec_epi;                 // every epilogue operand is 16-byte addressable: the split kernels use gemm_epilogue_lds
    int scalar_epi;              // caller asked for the register-direct epilogue (tgp_gemm_args.epilogue == 1)
    const int *pred;             // device flag: the launch does nothing while *pred == 0 (tgp_gemm_args.pred; split tile kernels)
    int64_t plane;               // elements between planes
    // tile schedule of the main kernel: per batch, M-tile rows [0, mt_big) use 128x128 tiles, the rest 64x64
    int mt_big, tiles_n_big, tiles_big, tiles_m_small, tiles_n_small;
    // (round 4) operands / results as BLOCKED fp16 planes (tgp_gemm_args.A_planes ...; layout: tgp_planes_bytes in tgpose.h):
    // chunk (row block rb = row / 32, K-tile kt = k / 16) is 2 KB = [plane hi | lo][half h = (k % 16) / 8][row % 32][8 fp16]
    const char *Ap, *Wp;         // operand planes of the pre-split kernel (gemm_pp.hip)
    int a_kt, w_kt;              // K-tiles per row block in Ap / Wp
    const uint32_t *a_amax;      // per 32-row block of A: bits of max |a| (NaN / inf read as huge); NULL = unguarded
    char *Cp;                    // result planes written by the LDS-staged epilogue beside (or instead of) C
    int c_kt, cp_col0;           // K-tiles per row block of Cp; plane column of the launch's output column c_col0
    uint32_t *c_amax;            // per 32-row block of the result: atomicMax of the bits of |v| over the columns written
    int pp_tiles_m, pp_tiles_n;  // tile grid of the pre-split kernel
    int *range_flag;             // planes-only operand (A == NULL): raised instead of the exact recomputation
};

// byte offset of (row, 4 consecutive plane columns pc .. pc + 3, pc % 4 == 0) in the hi plane of a blocked-planes buffer; lo is + 1024
__device__ __forceinline__ int64_t tgp_plane_off(const int row, const int pc, const int kts)
{
    return ((int64_t)(row >> 5) * kts + (pc >> 4)) * 2048 + ((pc >> 3) & 1) * 512 + (row & 31) * 16 + (pc & 4) * 2;
}

// Fused epilogue shared by every tile kernel.  C/D layout of a 32x32 MFMA tile (any input dtype):
// col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
template <int TM, int TN, int WTM, int WTN, bool DIST>
__device__ __forceinline__ void gemm_epilogue(const GemmParams &p, f32x16 (&acc)[TM][TN], const int m0, const int n0,
                                              const int z, const int wm, const int wn, const int r, const int h)
{
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + r;
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
                    const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
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
                const int rbase = m0 + wm * WTM + i * 32 + 4 * h;
                int obj = 0, bound = 0x7fffffff;
                if (p.rowbias || p.cm) {
                    obj = (rbase + p.row_base) / p.rows_per_obj;
                    bound = (obj + 1) * p.rows_per_obj - p.row_base;
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

// The same epilogue restructured for instruction count (the generic form above costs ~50 instructions per element, and
// with 64 elements per lane and 4 waves per SIMD that was 20 us per 256 x 256 tile -- 15 % of the wide GEMM):
//  * addresses are (wave-uniform base + compile-time row offset * ld) + one per-lane offset, so a store or a residual load
//    is one instruction with a scalar base;
//  * a wave tile (<= 64 rows) touches at most two objects when rows_per_obj >= its height: the per-object bias is two
//    preloaded values and a select, the max over points two running keys and two atomics per 16 rows.
// Preconditions (checked by the caller, else the generic form runs): rows_per_obj >= WTM when rowbias / colmax are used.
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmParams &p, f32x16 (&acc)[TM][TN], const int m0, const int n0,
                                                   const int z, const int wm, const int wn, const int r, const int h)
{
    const int64_t vo = (int64_t)z * p.sV;
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;       // wave-uniform
    const bool full_rows = row0 + WTM <= p.M;
    int obj0 = 0, bound = 0x7fffffff;
    if (p.rowbias || p.cm) {
        obj0 = (row0 + p.row_base) / p.rows_per_obj;
        bound = (obj0 + 1) * p.rows_per_obj - p.row_base;                     // rows >= bound belong to object obj0 + 1
    }
    const int last_row = (row0 + WTM < p.M ? row0 + WTM : p.M) - 1;
    const bool two_objs = bound <= last_row;
    float *Cb = p.C ? p.C + (int64_t)z * p.sC + (int64_t)row0 * p.ldc + (col0 - p.c_col0) : nullptr;
    const float *R1 = p.res1 ? p.res1 + (int64_t)row0 * p.ldr1 + col0 : nullptr;
    const float *R2 = p.res2 ? p.res2 + (int64_t)row0 * p.ldr2 + col0 : nullptr;
    const int lane_c = 4 * h * p.ldc + r, lane_r1 = 4 * h * p.ldr1 + r, lane_r2 = 4 * h * p.ldr2 + r;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32 + r;
        const bool colok = col < p.N;
        const float bias = (colok && p.bias) ? p.bias[vo + col] : 0.f;
        const float sc = (colok && p.scale) ? p.scale[vo + col] : 1.f;
        const float sh = (colok && p.shift) ? p.shift[vo + col] : 0.f;
        const float slope = (colok && p.slope_vec) ? p.slope_vec[vo + col] : p.slope;
        const bool store_c = Cb && colok && col >= p.c_col0;
        const bool do_cm = p.cm && colok && col < p.cm_cols;
        float rb0 = 0.f, rb1 = 0.f;
        if (p.rowbias && colok && row0 < p.M) {   // a wave tile wholly past the last row has no object: nothing to read
            rb0 = p.rowbias[(int64_t)obj0 * p.ldrb + col];
            if (two_objs) rb1 = p.rowbias[(int64_t)(obj0 + 1) * p.ldrb + col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint32_t key0 = 0, key1 = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int roff = i * 32 + (e & 3) + 8 * (e >> 2);            // compile-time
                const int row = row0 + roff + 4 * h;
                const bool ok = full_rows || row < p.M;
                const bool second = row >= bound;
                float v = acc[i][j][e] + bias;
                if (p.rowbias) v += second ? rb1 : rb0;
                if (R1) v += (colok && ok) ? R1[(int64_t)roff * p.ldr1 + (lane_r1 + j * 32)] : 0.f;
                if (R2) v += (colok && ok) ? R2[(int64_t)roff * p.ldr2 + (lane_r2 + j * 32)] : 0.f;
                if (p.scale) v = v * sc + sh;
                if (p.act == 1) v = v > 0.f ? v : v * slope;
                if (store_c && ok) Cb[(int64_t)roff * p.ldc + (lane_c + j * 32)] = v;
                if (do_cm && ok) {
                    const uint32_t key = tgp_float_key(v);
                    if (second) key1 = key > key1 ? key : key1;
                    else key0 = key > key0 ? key : key0;
                }
            }
            if (do_cm) {
                uint32_t *cm = p.cm + (int64_t)z * p.sCM + col;
                if (key0) atomicMax(cm + (int64_t)obj0 * p.ldcm, key0);
                if (key1) atomicMax(cm + (int64_t)(obj0 + 1) * p.ldcm, key1);
            }
        }
    }
}

// The fast epilogue with gathered residuals (factored wide layers).  Row-block outermost so that the 16 gather indices a lane
// needs for a 32-row block are loaded once and serve every column block; otherwise the same arithmetic, in the same order,
// as gemm_epilogue_fast: bias, gathered residuals, per-object bias, BN scale / shift, activation, store, max over points.
// Preconditions: no plain residuals (res1 / res2), rows_per_obj >= WTM when rowbias / colmax are used.
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue_gather(const GemmParams &p, f32x16 (&acc)[TM][TN], const int m0, const int n0,
                                                     const int z, const int wm, const int wn, const int r, const int h)
{
    const int64_t vo = (int64_t)z * p.sV;
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;       // wave-uniform
    const bool full_rows = row0 + WTM <= p.M;
    int obj0 = 0, bound = 0x7fffffff;
    if (p.rowbias || p.cm) {
        obj0 = (row0 + p.row_base) / p.rows_per_obj;
        bound = (obj0 + 1) * p.rows_per_obj - p.row_base;
    }
    const int last_row = (row0 + WTM < p.M ? row0 + WTM : p.M) - 1;
    const bool two_objs = bound <= last_row;
    float *Cb = p.C ? p.C + (int64_t)z * p.sC + (int64_t)row0 * p.ldc + (col0 - p.c_col0) : nullptr;
    const int lane_c = 4 * h * p.ldc + r;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int g1[16], g2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const bool ok = full_rows || row < p.M;
            g1[e] = (p.gres1 && ok) ? p.gidx1[row] : 0;
            g2[e] = (p.gres2 && ok) ? p.gidx2[row] : 0;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 32 + r;
            const bool colok = col < p.N;
            const float bias = (colok && p.bias) ? p.bias[vo + col] : 0.f;
            const float sc = (colok && p.scale) ? p.scale[vo + col] : 1.f;
            const float sh = (colok && p.shift) ? p.shift[vo + col] : 0.f;
            const float slope = (colok && p.slope_vec) ? p.slope_vec[vo + col] : p.slope;
            const bool store_c = Cb && colok && col >= p.c_col0;
            const bool do_cm = p.cm && colok && col < p.cm_cols;
            float rb0 = 0.f, rb1 = 0.f;
            if (p.rowbias && colok && row0 < p.M) {
                rb0 = p.rowbias[(int64_t)obj0 * p.ldrb + col];
                if (two_objs) rb1 = p.rowbias[(int64_t)(obj0 + 1) * p.ldrb + col];
            }
            const float *G1 = p.gres1 ? p.gres1 + col : nullptr;
            const float *G2 = p.gres2 ? p.gres2 + col : nullptr;
            uint32_t key0 = 0, key1 = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int roff = i * 32 + (e & 3) + 8 * (e >> 2);
                const int row = row0 + roff + 4 * h;
                const bool ok = full_rows || row < p.M;
                const bool second = row >= bound;
                float v = acc[i][j][e] + bias;
                if (G1) v += (colok && ok) ? G1[(int64_t)g1[e] * p.ldg1] : 0.f;
                if (G2) v += (colok && ok) ? G2[(int64_t)g2[e] * p.ldg2] : 0.f;
                if (p.rowbias) v += second ? rb1 : rb0;
                if (p.scale) v = v * sc + sh;
                if (p.act == 1) v = v > 0.f ? v : v * slope;
                if (store_c && ok) Cb[(int64_t)roff * p.ldc + (lane_c + j * 32)] = v;
                if (do_cm && ok) {
                    const uint32_t key = tgp_float_key(v);
                    if (second) key1 = key > key1 ? key : key1;
                    else key0 = key > key0 ? key : key0;
                }
            }
            if (do_cm) {
                uint32_t *cm = p.cm + (int64_t)z * p.sCM + col;
                if (key0) atomicMax(cm + (int64_t)obj0 * p.ldcm, key0);
                if (key1) atomicMax(cm + (int64_t)(obj0 + 1) * p.ldcm, key1);
            }
        }
    }
}

// LDS-staged epilogue of the split kernels.  The MFMA accumulator layout gives a lane ONE column and 16 rows of a 32 x 32
// block, so the register-direct epilogues move 4 bytes per lane and instruction: 64 stores (plus one load per residual) per
// lane and tile, and gathered residuals -- two more loads per element whose rows differ from instruction to instruction --
// cost 60-90 us per 256 x 256 tile that way.  Here each wave turns its blocks through a private 4 KB LDS region
// (32 x 32 floats, unpadded: the 16-lane groups of ds_read_b128 cover two whole rows = all 64 banks): written in accumulator
// layout, read back as lane = (row = pass * 8 + lane / 8, four consecutive columns), so every global access of the epilogue
// -- bias / scale / shift vectors, residual rows, gathered rows, the store -- is 16 bytes per lane and there are a quarter as
// many of them.  Arithmetic per element is that of gemm_epilogue_fast, in the same order (results are bit-identical).
// The max over an object's points is reduced over the 8 lanes that share a column quad with 3 xor-shuffles before the
// atomics.  Preconditions (host: vec_epi): N, ldc, c_col0, residual / gather / per-object-bias strides multiples of 4 and
// their bases 16-byte aligned; rows_per_obj >= WTM with rowbias / colmax; all waves are past their last operand read.
template <int TM, int TN, int WTM, int WTN, bool PLANES = false>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams &p, f32x16 (&acc)[TM][TN], const int m0, const int n0, const int z,
                                                  const int wm, const int wn, const int r, const int h, float *stage)
{
    const int lane = r + 32 * h;
    const int64_t vo = (int64_t)z * p.sV;
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;       // wave-uniform
    int obj0 = 0, bound = 0x7fffffff;
    if (p.rowbias || p.cm) {
        obj0 = (row0 + p.row_base) / p.rows_per_obj;
        bound = (obj0 + 1) * p.rows_per_obj - p.row_base;
    }
    const int last_row = (row0 + WTM < p.M ? row0 + WTM : p.M) - 1;
    const bool two_objs = bound <= last_row;
    const int lr = lane >> 3, cq = (lane & 7) * 4;               // row within a pass, first of the lane's four columns
    uint32_t amax_bits[TM];                                      // planes output: bits of the largest |v| per 32-row block
#pragma unroll
    for (int i = 0; i < TM; ++i) amax_bits[i] = 0u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32 + cq;
        const bool colok = col < p.N;                            // N % 4 == 0: the quad is wholly inside or outside
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bias = (colok && p.bias) ? *reinterpret_cast<const float4 *>(p.bias + vo + col) : zero4;
        const float4 sc = (colok && p.scale) ? *reinterpret_cast<const float4 *>(p.scale + vo + col) : one4;
        const float4 sh = (colok && p.shift) ? *reinterpret_cast<const float4 *>(p.shift + vo + col) : zero4;
        const float4 slope = (colok && p.slope_vec) ? *reinterpret_cast<const float4 *>(p.slope_vec + vo + col)
                                                    : make_float4(p.slope, p.slope, p.slope, p.slope);
        const bool store_c = p.C && colok && col >= p.c_col0;
        const bool do_cm = p.cm && col < p.cm_cols;              // cm_cols % 4 == 0 (host)
        float4 rb0 = zero4, rb1 = zero4;
        if (p.rowbias && colok && row0 < p.M) {
            rb0 = *reinterpret_cast<const float4 *>(p.rowbias + (int64_t)obj0 * p.ldrb + col);
            if (two_objs) rb1 = *reinterpret_cast<const float4 *>(p.rowbias + (int64_t)(obj0 + 1) * p.ldrb + col);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int gi1[4], gi2[4];                                  // fetched while the accumulators are being staged
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + i * 32 + q * 8 + lr;
                gi1[q] = (p.gres1 && row < p.M) ? p.gidx1[row] : 0;
                gi2[q] = (p.gres2 && row < p.M) ? p.gidx2[row] : 0;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][j][e];
            uint32_t k0[4] = {0, 0, 0, 0}, k1[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + i * 32 + q * 8 + lr;
                const bool ok = row < p.M;
                const bool second = row >= bound;
                float4 v = *reinterpret_cast<const float4 *>(stage + (q * 8 + lr) * 32 + cq);
                v.x += bias.x, v.y += bias.y, v.z += bias.z, v.w += bias.w;
                if (p.gres1 && colok && ok) {
                    const float4 g = *reinterpret_cast<const float4 *>(p.gres1 + (int64_t)gi1[q] * p.ldg1 + col);
                    v.x += g.x, v.y += g.y, v.z += g.z, v.w += g.w;
                }
                if (p.gres2 && colok && ok) {
                    const float4 g = *reinterpret_cast<const float4 *>(p.gres2 + (int64_t)gi2[q] * p.ldg2 + col);
                    v.x += g.x, v.y += g.y, v.z += g.z, v.w += g.w;
                }
                if (p.rowbias) {
                    const float4 rb = second ? rb1 : rb0;
                    v.x += rb.x, v.y += rb.y, v.z += rb.z, v.w += rb.w;
                }
                if (p.res1 && colok && ok) {
                    const float4 g = *reinterpret_cast<const float4 *>(p.res1 + (int64_t)row * p.ldr1 + col);
                    v.x += g.x, v.y += g.y, v.z += g.z, v.w += g.w;
                }
                if (p.res2 && colok && ok) {
                    const float4 g = *reinterpret_cast<const float4 *>(p.res2 + (int64_t)row * p.ldr2 + col);
                    v.x += g.x, v.y += g.y, v.z += g.z, v.w += g.w;
                }
                if (p.scale) v.x = v.x * sc.x + sh.x, v.y = v.y * sc.y + sh.y, v.z = v.z * sc.z + sh.z, v.w = v.w * sc.w + sh.w;
                if (p.act == 1) {
                    v.x = v.x > 0.f ? v.x : v.x * slope.x, v.y = v.y > 0.f ? v.y : v.y * slope.y;
                    v.z = v.z > 0.f ? v.z : v.z * slope.z, v.w = v.w > 0.f ? v.w : v.w * slope.w;
                }
                if (store_c && ok)
                    *reinterpret_cast<float4 *>(p.C + (int64_t)z * p.sC + (int64_t)row * p.ldc + (col - p.c_col0)) = v;
                if (PLANES && p.Cp && colok && ok && col >= p.c_col0) {
                    // the same value as fp16 hi / lo planes, in the blocked layout the pre-split GEMM stages by LDS-DMA: a
                    // deterministic function of the fp32 value, so the consumer's products equal those of an in-loop split
                    uint2 ph, pl;
                    split2(v, ph, pl);
                    char *dst = p.Cp + tgp_plane_off(row, col - p.c_col0 + p.cp_col0, p.c_kt);
                    *reinterpret_cast<uint2 *>(dst) = ph;
                    *reinterpret_cast<uint2 *>(dst + 1024) = pl;
                    const uint32_t b0 = __float_as_uint(v.x) & 0x7fffffffu, b1 = __float_as_uint(v.y) & 0x7fffffffu;
                    const uint32_t b2 = __float_as_uint(v.z) & 0x7fffffffu, b3 = __float_as_uint(v.w) & 0x7fffffffu;
                    const uint32_t m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3, m = m01 > m23 ? m01 : m23;
                    amax_bits[i] = m > amax_bits[i] ? m : amax_bits[i];
                }
                if (do_cm && colok && ok) {
                    const uint32_t key[4] = {tgp_float_key(v.x), tgp_float_key(v.y), tgp_float_key(v.z), tgp_float_key(v.w)};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (second) k1[c] = key[c] > k1[c] ? key[c] : k1[c];
                        else k0[c] = key[c] > k0[c] ? key[c] : k0[c];
                    }
                }
            }
            if (p.cm) {                                          // wave-uniform branch: the shuffles need every lane
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int m = 8; m < 64; m <<= 1) {
                        const uint32_t o0 = (uint32_t)__shfl_xor((int)k0[c], m), o1 = (uint32_t)__shfl_xor((int)k1[c], m);
                        k0[c] = o0 > k0[c] ? o0 : k0[c], k1[c] = o1 > k1[c] ? o1 : k1[c];
                    }
                }
                if (do_cm && colok && lane < 8) {
                    uint32_t *cm = p.cm + (int64_t)z * p.sCM + col;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (k0[c]) atomicMax(cm + (int64_t)obj0 * p.ldcm + c, k0[c]);
                        if (k1[c]) atomicMax(cm + (int64_t)(obj0 + 1) * p.ldcm + c, k1[c]);
                    }
                }
            }
        }
    }
    if (PLANES && p.Cp && p.c_amax) {                            // wave-uniform
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint32_t m = amax_bits[i];
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)m, s);
                m = o > m ? o : m;
            }
            const int rb = (row0 + i * 32) >> 5;
            if (lane == 0 && m && row0 + i * 32 < p.M) atomicMax(p.c_amax + rb, m);
        }
    }
}
