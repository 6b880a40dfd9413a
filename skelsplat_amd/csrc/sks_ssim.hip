// sks_ssim.hip -- fused SSIM forward / backward for MI355X (gfx950).
// Reference: submodules/fused-ssim/ssim.cu:187-444 (kernels), fused_ssim/__init__.py:8-41 (autograd wrapper).
//
// 11-tap separable Gaussian over five (forward) / three (backward) quantities per pixel.  The arithmetic, not HBM,
// is what this op has to get through (>= 110 FMA per pixel against 12..28 bytes; the vector ALU of the forward kernel is
// busy two thirds of the time, rocprofv3 SQ_ACTIVE_INST_VALU), so the kernels are built around the instruction count:
//   * a workgroup (256 threads) walks a strip of 64x32 output tiles DOWN the image and keeps the horizontally filtered
//     rows in LDS: a tile needs 42 of them, the last 10 of the tile above are reused, so only 32 new image rows are
//     fetched and filtered per tile (the 5-pixel halo costs 1.31x horizontal work when every tile starts from scratch);
//   * both images interleaved in LDS as (img1, img2) pairs, the horizontal sums interleaved as (mu1, mu2) and
//     (E[x^2], E[y^2]) pairs: every tap of a pair is ONE packed FMA (v_pk_fma_f32) without any register shuffling and
//     every LDS access is 8 or 16 bytes; the vertical pass pairs two adjacent columns the same way;
//   * register blocking: 4 adjacent outputs per horizontal item (14 inputs), 2 columns x 4 rows per vertical thread;
//   * the next tile's 32 image rows are fetched into registers (aligned 16-byte pieces, columns x0-8 .. x0+71, when
//     W % 4 == 0) while the current tile is convolved.
// Built with -fno-slp-vectorize (build.py): the SLP vectoriser packs the remaining scalar FMAs of neighbouring outputs
// into v_pk_fma_f32 and pays for it with ~60 v_mov per item.
// Taps are accumulated left-to-right / top-to-bottom like ssim.cu:100-185 with explicit FMAs (nvcc contracts the
// reference's `sum += g * v` the same way); sigma = E[x^2] - mu^2 and the map / partial-derivative expressions are
// ssim.cu:262-283 as written (their quotients through div_by below: same bits, fewer instructions).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <initializer_list>

#include "../../include/skelsplat_hip.h"
#include "sks_err.h"
#include "sks_math.h"

namespace {

using sks::wave_sum_d;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int TW = 64, TH = 32, HALO = 5;
constexpr int HR = TH + 2 * HALO;        // 42 horizontally filtered rows per tile
constexpr int IW = TW + 16;              // 80 staged columns: x0-8 .. x0+71, a 16-byte aligned window around the halo
constexpr int HG = TW / 4;               // horizontal items per row (16)
// LDS images are arrays of 16-byte chunks, and chunk `slot` of row `row` is stored at slot ^ (row & 1) (at
// slot ^ ((row & 1) << 3) where a row has only 16 chunks).  Work items are dealt to lanes as 8 neighbours of one row then
// the same 8 of the next row, so the 16 lanes the LDS serves together touch 16 different chunks of a 256-byte bank
// window although neighbours in a row are 32 bytes apart.
constexpr int IN_SLOTS = IW / 2;         // chunks per row of an (a, b) pair image of the staged rows (40)
constexpr int H2_SLOTS = TW / 2;         // chunks per row of a pair image after the horizontal pass (32)
constexpr int H1_SLOTS = TW / 4;         // chunks per row of a single-float image after the horizontal pass (16)
constexpr int STAGE_RP = 48;             // staging slots per row pair: 3 groups of (8 pieces x 2 rows), 20 pieces per row
constexpr int NLD = (TH / 2) * STAGE_RP / NT;   // staging pieces per thread and tile (3)
static_assert((TH / 2) * STAGE_RP == NLD * NT, "staging slots must fill the block");
static_assert(HALO * STAGE_RP <= NT, "the strip prologue (10 rows) is one piece per thread");

__device__ __forceinline__ float gk(int t)   // ssim.cu:9-19
{
    switch (t) {
    case 0: case 10: return 0.001028380123898387f;
    case 1: case 9: return 0.0075987582094967365f;
    case 2: case 8: return 0.036000773310661316f;
    case 3: case 7: return 0.10936068743467331f;
    case 4: case 6: return 0.21300552785396576f;
    default: return 0.26601171493530273f;
    }
}

// IEEE quotients that share a denominator: one refined reciprocal per denominator, two residual corrections per
// numerator (the hardware's own division sequence without the range scaling).  Bit-identical to `n / d` for
// d in [2^-40, 2^8), |n| in [2^-60, 2^12) -- 4.3e9 random pairs, tools/div_check.hip -- which covers the denominators
// A B, A A B, A B B (A >= C1, B >= C2 up to rounding) and numerators of ssim.cu:262-283; 44 instead of 77 instructions for
// the seven quotients of the training form.
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_by(float n, float d, float r)
{
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}

__device__ __forceinline__ v2f pk_fma(float g, v2f a, v2f c)
{
    const v2f gg = { g, g };
    return __builtin_elementwise_fma(gg, a, c);
}

// one 16-byte piece of an image row: columns x .. x+3 of `row` (nullptr = the row is outside the image)
template <bool VEC>
__device__ __forceinline__ v4f load_piece(const float* __restrict__ row, int x, int W)
{
    v4f r = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (row) {
        if (VEC) {
            if (x >= 0 && x < W) r = *reinterpret_cast<const v4f*>(row + x);   // W % 4 == 0: all four inside
        } else {
            if (x >= 0 && x < W) r.x = row[x];
            if (x + 1 >= 0 && x + 1 < W) r.y = row[x + 1];
            if (x + 2 >= 0 && x + 2 < W) r.z = row[x + 2];
            if (x + 3 >= 0 && x + 3 < W) r.w = row[x + 3];
        }
    }
    return r;
}

// this thread's k-th staging piece: staged row, first tile-local column, LDS chunk index of its first two pixels in a
// pair image (-1 = no piece; the chunk of pixels 2, 3 is that index ^ 1)
struct Piece {
    int r, col, dst;
};
__device__ __forceinline__ Piece piece_of(int k)
{
    const int s = threadIdx.x + k * NT;
    const int rp = s / STAGE_RP, w = s - rp * STAGE_RP;
    const int c4 = (w >> 4) * 8 + (w & 7);
    Piece p;
    p.r = 2 * rp + ((w >> 3) & 1);
    p.col = c4 * 4 - 8;
    p.dst = c4 < IW / 4 ? p.r * IN_SLOTS + ((2 * c4) ^ (p.r & 1)) : -1;
    return p;
}

// horizontal item i -> (row, group of 4 outputs): lanes 0-7 one row, lanes 8-15 the next row
__device__ __forceinline__ void item_of(int i, int& row, int& g, int& par)
{
    par = (i >> 3) & 1;
    row = 2 * (i >> 5) + par;
    g = (i & 7) | (((i >> 4) & 1) << 3);
}

// the last 10 filtered rows of a tile are the first 10 of the tile below (row parity, hence the swizzle, is unchanged)
__device__ __forceinline__ void carry_rows(v4f* h, int slots)
{
    for (int c = threadIdx.x; c < 2 * HALO * slots; c += NT) h[c] = h[TH * slots + c];
}

// ---- forward ---------------------------------------------------------------------------------------------------
// horizontal pass (ssim.cu:100-164) over `nrows` staged rows -> filtered rows hoff .. hoff+nrows-1 (hoff even)
template <int NROWS>
__device__ __forceinline__ void fwd_rows(const v4f* s_in, v4f* s_h01, v4f* s_h23, v4f* s_h4, int hoff)
{
#pragma unroll
    for (int i = threadIdx.x; i < NROWS * HG; i += NT) {   // 2 items per thread per tile: the second one's LDS reads are
                                                            // in flight under the first one's FMAs
        int row, g, par;
        item_of(i, row, g, par);
        const int base = row * IN_SLOTS + 2 * g + 1;   // chunks 2g+1 .. 2g+8 = tile-local columns 4g-6 .. 4g+9
        v2f in[16];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const v4f t = s_in[base + q + ((q & 1) ? par : -par)];   // (2g + 1 + q) ^ par
            in[2 * q] = (v2f){ t.x, t.y };
            in[2 * q + 1] = (v2f){ t.z, t.w };
        }
        v2f m[4], s[4];
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { m[k] = (v2f){ 0.0f, 0.0f }; s[k] = (v2f){ 0.0f, 0.0f }; c[k] = 0.0f; }
#pragma unroll
        for (int jn = 0; jn < 14; jn++) {
            const v2f uw = in[jn + 1];
            const v2f sq = uw * uw;
            const float p = uw.x * uw.y;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = jn - k;
                if (t >= 0 && t < 11) {
                    m[k] = pk_fma(gk(t), uw, m[k]);
                    s[k] = pk_fma(gk(t), sq, s[k]);
                    c[k] = __builtin_fmaf(gk(t), p, c[k]);
                }
            }
        }
        const int hr = row + hoff;
        const int d2 = hr * H2_SLOTS + ((2 * g) ^ par);
        s_h01[d2] = (v4f){ m[0].x, m[0].y, m[1].x, m[1].y };
        s_h01[d2 ^ 1] = (v4f){ m[2].x, m[2].y, m[3].x, m[3].y };
        s_h23[d2] = (v4f){ s[0].x, s[0].y, s[1].x, s[1].y };
        s_h23[d2 ^ 1] = (v4f){ s[2].x, s[2].y, s[3].x, s[3].y };
        s_h4[hr * H1_SLOTS + (g ^ (par << 3))] = (v4f){ c[0], c[1], c[2], c[3] };
    }
}

// MODE bits: 1 = write ssim_map, 2 = write the three partial-derivative maps (train), 4 = add the sum of the map over the
// image shrunk by `crop` pixels per side to total[] (fused_ssim()'s `.mean()` without a pass over the map).
// grid (x tiles, strips of S tiles, B*CH planes).
template <bool VEC, int MODE>
__global__ __launch_bounds__(NT) void k_ssim_fwd(int H, int W, int S, float C1, float C2, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                  float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                  float* __restrict__ dm_dsigma12, int crop, double* __restrict__ total)
{
    __shared__ v4f s_in[TH * IN_SLOTS];                          // (img1, img2) per pixel, the 32 (10) rows being filtered
    __shared__ v4f s_h01[HR * H2_SLOTS], s_h23[HR * H2_SLOTS];   // (mu1, mu2), (E11, E22) after x
    __shared__ v4f s_h4[HR * H1_SLOTS];                          // E12 after x
    const int tid = threadIdx.x;
    const int tiles_y = (H + TH - 1) / TH;
    const int ty0 = blockIdx.y * S;
    const int njobs = min(S, tiles_y - ty0);
    const int x0 = blockIdx.x * TW;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* p1 = img1 + plane;
    const float* p2 = img2 + plane;

    Piece pc[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) pc[k] = piece_of(k);
    v4f r1[NLD], r2[NLD];
    auto fetch = [&](int ybase, int k) {
        const int y = ybase + pc[k].r;
        const bool in = y >= 0 && y < H;
        const size_t ro = (size_t)(in ? y : 0) * W;
        r1[k] = load_piece<VEC>(in ? p1 + ro : nullptr, x0 + pc[k].col, W);
        r2[k] = load_piece<VEC>(in ? p2 + ro : nullptr, x0 + pc[k].col, W);
    };
    auto stage = [&](int k) {
        if (pc[k].dst >= 0) {
            s_in[pc[k].dst] = (v4f){ r1[k].x, r2[k].x, r1[k].y, r2[k].y };
            s_in[pc[k].dst ^ 1] = (v4f){ r1[k].z, r2[k].z, r1[k].w, r2[k].w };
        }
    };
    // strip prologue: the 10 image rows above the first tile's own 32 (one piece per thread; rows 0-9 of piece 0)
    const int ytop = ty0 * TH - HALO;
    if (tid < HALO * STAGE_RP) {
        fetch(ytop, 0);
        stage(0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; k++) fetch(ytop + 2 * HALO, k);   // in flight during the prologue's pass
    fwd_rows<2 * HALO>(s_in, s_h01, s_h23, s_h4, 0);
    __syncthreads();

    double acc = 0.0;
    for (int j = 0; j < njobs; j++) {
        const int y0 = (ty0 + j) * TH;
#pragma unroll
        for (int k = 0; k < NLD; k++) stage(k);
        if (j > 0) {
            carry_rows(s_h01, H2_SLOTS);
            carry_rows(s_h23, H2_SLOTS);
            carry_rows(s_h4, H1_SLOTS);
        }
        __syncthreads();
        if (j + 1 < njobs) {
#pragma unroll
            for (int k = 0; k < NLD; k++) fetch(y0 + TH + HALO, k);   // the next tile's rows, in flight during the passes
        }
        fwd_rows<TH>(s_in, s_h01, s_h23, s_h4, 2 * HALO);
        __syncthreads();
        {
            // vertical pass (ssim.cu:166-185, 218-260): columns 2cx, 2cx+1; rows 4rg .. 4rg+3
            const int cx = tid & 31, rg = tid >> 5;
            v2f M[2][4], Sg[2][4], Cc[4];
#pragma unroll
            for (int o = 0; o < 4; o++) M[0][o] = M[1][o] = Sg[0][o] = Sg[1][o] = Cc[o] = (v2f){ 0.0f, 0.0f };
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const int row = 4 * rg + r, par = r & 1;
                const v4f ta = s_h01[row * H2_SLOTS + (cx ^ par)], tb = s_h23[row * H2_SLOTS + (cx ^ par)];
                const v2f te = reinterpret_cast<const v2f*>(s_h4)[(row * H1_SLOTS + ((cx >> 1) ^ (par << 3))) * 2 + (cx & 1)];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int t = r - o;
                    if (t >= 0 && t < 11) {
                        M[0][o] = pk_fma(gk(t), (v2f){ ta.x, ta.y }, M[0][o]);
                        M[1][o] = pk_fma(gk(t), (v2f){ ta.z, ta.w }, M[1][o]);
                        Sg[0][o] = pk_fma(gk(t), (v2f){ tb.x, tb.y }, Sg[0][o]);
                        Sg[1][o] = pk_fma(gk(t), (v2f){ tb.z, tb.w }, Sg[1][o]);
                        Cc[o] = pk_fma(gk(t), te, Cc[o]);
                    }
                }
            }
            const int x = x0 + 2 * cx;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int y = y0 + 4 * rg + o;
                float mv[2], d1[2], d2[2], d3[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float mu1 = M[h][o].x, mu2 = M[h][o].y;
                    const float sigma1_sq = Sg[h][o].x - mu1 * mu1;
                    const float sigma2_sq = Sg[h][o].y - mu2 * mu2;
                    const float sigma12 = (h ? Cc[o].y : Cc[o].x) - mu1 * mu2;
                    // ssim.cu:262-283
                    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
                    const float Cn = (2.0f * mu1_mu2 + C1);
                    const float D = (2.0f * sigma12 + C2);
                    const float A = (mu1_sq + mu2_sq + C1);
                    const float B = (sigma1_sq + sigma2_sq + C2);
                    const float AB = (A * B), rAB = refined_rcp(AB);
                    mv[h] = div_by((Cn * D), AB, rAB);
                    if (MODE & 2) {
                        const float AAB = (A * A * B), ABB = (A * B * B), rAAB = refined_rcp(AAB), rABB = refined_rcp(ABB);
                        d1[h] = (div_by((mu2 * 2.0f * D), AB, rAB) - div_by((mu2 * 2.0f * Cn), AB, rAB) -
                                 div_by((mu1 * 2.0f * Cn * D), AAB, rAAB) + div_by((mu1 * 2.0f * Cn * D), ABB, rABB));
                        d2[h] = div_by((-Cn * D), ABB, rABB);
                        d3[h] = div_by((2 * Cn), AB, rAB);
                    }
                }
                if ((MODE & 4) && y >= crop && y < H - crop) {
                    if (x >= crop && x < W - crop) acc += (double)mv[0];
                    if (x + 1 >= crop && x + 1 < W - crop) acc += (double)mv[1];
                }
                if ((MODE & 3) && y < H) {
                    const size_t gi = plane + (size_t)y * W + x;
                    if (VEC) {
                        if (x < W) {
                            if (MODE & 1) *reinterpret_cast<v2f*>(ssim_map + gi) = (v2f){ mv[0], mv[1] };
                            if (MODE & 2) {
                                *reinterpret_cast<v2f*>(dm_dmu1 + gi) = (v2f){ d1[0], d1[1] };
                                *reinterpret_cast<v2f*>(dm_dsigma1_sq + gi) = (v2f){ d2[0], d2[1] };
                                *reinterpret_cast<v2f*>(dm_dsigma12 + gi) = (v2f){ d3[0], d3[1] };
                            }
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (x + h < W) {
                                if (MODE & 1) ssim_map[gi + h] = mv[h];
                                if (MODE & 2) {
                                    dm_dmu1[gi + h] = d1[h];
                                    dm_dsigma1_sq[gi + h] = d2[h];
                                    dm_dsigma12[gi + h] = d3[h];
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();   // the next tile's staging / carried rows must not overtake this tile's LDS reads
    }
    if (MODE & 4) {
        __shared__ double s_red[NT / 64];
        acc = wave_sum_d(acc);
        if ((tid & 63) == 0) s_red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0)
            atomicAdd(total + ((blockIdx.x + blockIdx.y + blockIdx.z) & (SKS_SSIM_SUM_SLOTS - 1)),
                      (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    }
}

// adds the slots, writes the mean and leaves the scratch zeroed for the next call.  (Folding this into the kernel above as
// "the last workgroup to arrive" costs a device-scope fence per workgroup: 86 -> 145 us on 5 x 1500 x 1500.)
__global__ __launch_bounds__(64) void k_ssim_mean_finish(double* __restrict__ total, float* __restrict__ mean_out, double inv_count)
{
    static_assert(SKS_SSIM_SUM_SLOTS == 64, "one slot per lane");
    const double t = wave_sum_d(total[threadIdx.x]);
    total[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) *mean_out = (float)(t * inv_count);
}

// ---- backward (ssim.cu:288-366) ----------------------------------------------------------------------------------
// dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12)
constexpr int I1_SLOTS = 24;   // chunk pitch of the single-float staged image (20 used): the next row starts 8 chunks
                               // further mod 16, which separates the two rows of a lane group without a swizzle

template <int NROWS>
__device__ __forceinline__ void bwd_rows(const v4f* s_in01, const v4f* s_in2, v4f* s_h01, v4f* s_h2, int hoff)
{
#pragma unroll
    for (int i = threadIdx.x; i < NROWS * HG; i += NT) {   // horizontal pass (ssim.cu:318-340)
        int row, g, par;
        item_of(i, row, g, par);
        const int base = row * IN_SLOTS + 2 * g + 1;   // tile-local columns 4g-6 .. 4g+9
        const int base2 = row * I1_SLOTS + g;          // tile-local columns 4g-8 .. 4g+11
        v2f in[16];
        float in2[20];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const v4f t = s_in01[base + q + ((q & 1) ? par : -par)];
            in[2 * q] = (v2f){ t.x, t.y };
            in[2 * q + 1] = (v2f){ t.z, t.w };
        }
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const v4f t = s_in2[base2 + q];
            in2[4 * q] = t.x; in2[4 * q + 1] = t.y; in2[4 * q + 2] = t.z; in2[4 * q + 3] = t.w;
        }
        v2f m[4];
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { m[k] = (v2f){ 0.0f, 0.0f }; c[k] = 0.0f; }
#pragma unroll
        for (int jn = 0; jn < 14; jn++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = jn - k;
                if (t >= 0 && t < 11) {
                    m[k] = pk_fma(gk(t), in[jn + 1], m[k]);
                    c[k] = __builtin_fmaf(gk(t), in2[jn + 3], c[k]);
                }
            }
        }
        const int hr = row + hoff;
        const int d2 = hr * H2_SLOTS + ((2 * g) ^ par);
        s_h01[d2] = (v4f){ m[0].x, m[0].y, m[1].x, m[1].y };
        s_h01[d2 ^ 1] = (v4f){ m[2].x, m[2].y, m[3].x, m[3].y };
        s_h2[hr * H1_SLOTS + (g ^ (par << 3))] = (v4f){ c[0], c[1], c[2], c[3] };
    }
}

// UNIFORM: dL_dmap is one value (*dL_value) inside the image shrunk by `crop` pixels per side and zero outside --
// what `.mean()` of the ("valid"-cropped) map hands back -- so no gradient image is materialised or read.
template <bool VEC, bool UNIFORM>
__global__ __launch_bounds__(NT) void k_ssim_bwd(int H, int W, int S, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                  const float* __restrict__ dL_value, float dL_scale, int crop,
                                                  const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                  const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1)
{
    __shared__ v4f s_in01[TH * IN_SLOTS];   // (dL dm_dmu1, dL dm_dsigma1_sq)
    __shared__ v4f s_in2[TH * I1_SLOTS];    // dL dm_dsigma12
    __shared__ v4f s_h01[HR * H2_SLOTS];
    __shared__ v4f s_h2[HR * H1_SLOTS];
    const int tid = threadIdx.x;
    const int tiles_y = (H + TH - 1) / TH;
    const int ty0 = blockIdx.y * S;
    const int njobs = min(S, tiles_y - ty0);
    const int x0 = blockIdx.x * TW;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float dval = UNIFORM ? *dL_value * dL_scale : 0.0f;
    const int cr = UNIFORM ? crop : 0;

    Piece pc[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) pc[k] = piece_of(k);
    v4f rd[NLD], r0[NLD], r1[NLD], r2[NLD];
    auto fetch = [&](int ybase, int k) {
        const int y = ybase + pc[k].r;
        const bool in = y >= cr && y < H - cr;
        const size_t ro = plane + (size_t)(in ? y : 0) * W;
        const int x = x0 + pc[k].col;
        if (UNIFORM) {
            rd[k] = (v4f){ (in && x >= cr && x < W - cr) ? dval : 0.0f, (in && x + 1 >= cr && x + 1 < W - cr) ? dval : 0.0f,
                           (in && x + 2 >= cr && x + 2 < W - cr) ? dval : 0.0f,
                           (in && x + 3 >= cr && x + 3 < W - cr) ? dval : 0.0f };
        } else {
            rd[k] = load_piece<VEC>(in ? dL_dmap + ro : nullptr, x, W);
        }
        r0[k] = load_piece<VEC>(in ? dm_dmu1 + ro : nullptr, x, W);
        r1[k] = load_piece<VEC>(in ? dm_dsigma1_sq + ro : nullptr, x, W);
        r2[k] = load_piece<VEC>(in ? dm_dsigma12 + ro : nullptr, x, W);
    };
    auto stage = [&](int k) {
        if (pc[k].dst >= 0) {
            const v4f a = r0[k] * rd[k], b = r1[k] * rd[k];
            s_in01[pc[k].dst] = (v4f){ a.x, b.x, a.y, b.y };
            s_in01[pc[k].dst ^ 1] = (v4f){ a.z, b.z, a.w, b.w };
            s_in2[pc[k].r * I1_SLOTS + ((pc[k].col + 8) >> 2)] = r2[k] * rd[k];
        }
    };
    const int ytop = ty0 * TH - HALO;
    if (tid < HALO * STAGE_RP) {
        fetch(ytop, 0);
        stage(0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; k++) fetch(ytop + 2 * HALO, k);
    bwd_rows<2 * HALO>(s_in01, s_in2, s_h01, s_h2, 0);
    __syncthreads();

    for (int j = 0; j < njobs; j++) {
        const int y0 = (ty0 + j) * TH;
#pragma unroll
        for (int k = 0; k < NLD; k++) stage(k);
        if (j > 0) {
            carry_rows(s_h01, H2_SLOTS);
            carry_rows(s_h2, H1_SLOTS);
        }
        __syncthreads();
        if (j + 1 < njobs) {
#pragma unroll
            for (int k = 0; k < NLD; k++) fetch(y0 + TH + HALO, k);
        }
        bwd_rows<TH>(s_in01, s_in2, s_h01, s_h2, 2 * HALO);
        __syncthreads();
        {
            const int cx = tid & 31, rg = tid >> 5;   // vertical pass (ssim.cu:342-365)
            const int x = x0 + 2 * cx;
            v2f p1[4], p2[4];   // the images at this thread's outputs: requested now, used after the 132 FMAs below
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int y = y0 + 4 * rg + o;
                p1[o] = p2[o] = (v2f){ 0.0f, 0.0f };
                if (y < H && x < W) {
                    const size_t gi = plane + (size_t)y * W + x;
                    if (VEC) {
                        p1[o] = *reinterpret_cast<const v2f*>(img1 + gi);
                        p2[o] = *reinterpret_cast<const v2f*>(img2 + gi);
                    } else {
                        p1[o].x = img1[gi]; p2[o].x = img2[gi];
                        if (x + 1 < W) { p1[o].y = img1[gi + 1]; p2[o].y = img2[gi + 1]; }
                    }
                }
            }
            v2f A[2][4], Cc[4];
#pragma unroll
            for (int o = 0; o < 4; o++) A[0][o] = A[1][o] = Cc[o] = (v2f){ 0.0f, 0.0f };
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const int row = 4 * rg + r, par = r & 1;
                const v4f ta = s_h01[row * H2_SLOTS + (cx ^ par)];
                const v2f te = reinterpret_cast<const v2f*>(s_h2)[(row * H1_SLOTS + ((cx >> 1) ^ (par << 3))) * 2 + (cx & 1)];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int t = r - o;
                    if (t >= 0 && t < 11) {
                        A[0][o] = pk_fma(gk(t), (v2f){ ta.x, ta.y }, A[0][o]);
                        A[1][o] = pk_fma(gk(t), (v2f){ ta.z, ta.w }, A[1][o]);
                        Cc[o] = pk_fma(gk(t), te, Cc[o]);
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int y = y0 + 4 * rg + o;
                if (y < H && x < W) {
                    const size_t gi = plane + (size_t)y * W + x;
                    v2f o2;
                    o2.x = (A[0][o].x + p1[o].x * 2.0f * A[0][o].y) + p2[o].x * Cc[o].x;
                    o2.y = (A[1][o].x + p1[o].y * 2.0f * A[1][o].y) + p2[o].y * Cc[o].y;
                    if (VEC) {
                        *reinterpret_cast<v2f*>(dL_dimg1 + gi) = o2;
                    } else {
                        dL_dimg1[gi] = o2.x;
                        if (x + 1 < W) dL_dimg1[gi + 1] = o2.y;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- host ------------------------------------------------------------------------------------------------------
// tiles per strip: the longer the strip, the less of the halo is filtered twice (1 tile: 1.31x, 2: 1.16x, 8: 1.04x),
// but the chip wants a few thousand workgroups
int tiles_per_strip(int tiles_x, int tiles_y, int planes)
{
    const char* e = getenv("SKS_SSIM_MIN_BLOCKS");   // tuning / test switch (read per call: tests force long strips)
    const long long min_blocks = e ? atoll(e) : 2048ll;
    int S = 8;
    while (S > 1 && (long long)tiles_x * ((tiles_y + S - 1) / S) * planes < min_blocks) S /= 2;
    return S;
}

bool aligned16(std::initializer_list<const void*> ps)
{
    for (const void* p : ps)
        if (p && ((uintptr_t)p & 15)) return false;
    return true;
}

int check_shape(const char* what, int B, int CH, int H, int W)
{
    if (B < 0 || CH < 0 || H < 1 || W < 1) return fail2(-1, "%s: bad shape", what);
    return 0;
}

struct Launch {
    dim3 grid;
    int S;
};
Launch plan(int B, int CH, int H, int W)
{
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    Launch l;
    l.S = tiles_per_strip(tiles_x, tiles_y, B * CH);
    l.grid = dim3(tiles_x, (tiles_y + l.S - 1) / l.S, B * CH);
    return l;
}

template <int MODE>
void launch_fwd(bool vec, const Launch& l, hipStream_t st, int H, int W, float C1, float C2, const float* img1,
                const float* img2, float* ssim_map, float* d1, float* d2, float* d3, int crop, double* total)
{
    if (vec)
        hipLaunchKernelGGL((k_ssim_fwd<true, MODE>), l.grid, dim3(NT), 0, st, H, W, l.S, C1, C2, img1, img2, ssim_map, d1, d2, d3,
                           crop, total);
    else
        hipLaunchKernelGGL((k_ssim_fwd<false, MODE>), l.grid, dim3(NT), 0, st, H, W, l.S, C1, C2, img1, img2, ssim_map, d1, d2, d3,
                           crop, total);
}

template <bool UNIFORM>
void launch_bwd(bool vec, const Launch& l, hipStream_t st, int H, int W, const float* img1, const float* img2,
                const float* dL_dmap, const float* dL_value, float dL_scale, int crop, const float* d1, const float* d2,
                const float* d3, float* dL_dimg1)
{
    if (vec)
        hipLaunchKernelGGL((k_ssim_bwd<true, UNIFORM>), l.grid, dim3(NT), 0, st, H, W, l.S, img1, img2, dL_dmap, dL_value, dL_scale,
                           crop, d1, d2, d3, dL_dimg1);
    else
        hipLaunchKernelGGL((k_ssim_bwd<false, UNIFORM>), l.grid, dim3(NT), 0, st, H, W, l.S, img1, img2, dL_dmap, dL_value,
                           dL_scale, crop, d1, d2, d3, dL_dimg1);
}

}  // namespace

extern "C" {

int sks_fused_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream)
{
    if (int rc = check_shape("ssim", B, CH, H, W)) return rc;
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !ssim_map) return fail2(-2, "ssim: missing pointer");
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))
        return fail2(-2, "ssim: provide all three partial-derivative maps or none");
    const Launch l = plan(B, CH, H, W);
    const bool vec = W % 4 == 0 && aligned16({ img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 });
    if (dm_dmu1)
        launch_fwd<3>(vec, l, (hipStream_t)stream, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, 0,
                      nullptr);
    else
        launch_fwd<1>(vec, l, (hipStream_t)stream, H, W, C1, C2, img1, img2, ssim_map, nullptr, nullptr, nullptr, 0, nullptr);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_mean(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, int crop,
                        float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, double* scratch, float* mean_out,
                        void* stream)
{
    if (int rc = check_shape("ssim mean", B, CH, H, W)) return rc;
    if (crop < 0) return fail2(-1, "ssim mean: crop negative");
    if (!scratch || !mean_out) return fail2(-2, "ssim mean: missing pointer");
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))
        return fail2(-2, "ssim mean: provide all three partial-derivative maps or none");
    const double count = (double)B * CH * (H > 2 * crop ? H - 2 * crop : 0) * (W > 2 * crop ? W - 2 * crop : 0);
    if (B * CH == 0) {   // the mean of an empty map is nan (0 / 0), like torch's; with planes but no pixels the
                         // kernels below write 0 * nan
        HIP_TRY2(hipMemsetD32Async((hipDeviceptr_t)mean_out, 0x7fc00000, 1, (hipStream_t)stream));
        return 0;
    }
    if (!img1 || !img2) return fail2(-2, "ssim mean: missing pointer");
    const Launch l = plan(B, CH, H, W);
    const bool vec = W % 4 == 0 && aligned16({ img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 });
    const double inv = count > 0.0 ? 1.0 / count : __builtin_nan("");
    if (dm_dmu1)
        launch_fwd<6>(vec, l, (hipStream_t)stream, H, W, C1, C2, img1, img2, nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, crop,
                      scratch);
    else
        launch_fwd<4>(vec, l, (hipStream_t)stream, H, W, C1, C2, img1, img2, nullptr, nullptr, nullptr, nullptr, crop, scratch);
    HIP_TRY2(hipGetLastError());
    hipLaunchKernelGGL(k_ssim_mean_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, mean_out, inv);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_bwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                       float* dL_dimg1, void* stream)
{
    (void)C1; (void)C2;
    if (int rc = check_shape("ssim backward", B, CH, H, W)) return rc;
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail2(-2, "ssim backward: missing pointer");
    launch_bwd<false>(W % 4 == 0 && aligned16({ img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1 }),
                      plan(B, CH, H, W), (hipStream_t)stream, H, W, img1, img2, dL_dmap, nullptr, 0.0f, 0, dm_dmu1, dm_dsigma1_sq,
                      dm_dsigma12, dL_dimg1);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_bwd_uniform(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_value,
                               float dL_scale, int crop, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                               float* dL_dimg1, void* stream)
{
    if (int rc = check_shape("ssim backward", B, CH, H, W)) return rc;
    if (crop < 0) return fail2(-1, "ssim backward: crop negative");
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !dL_value || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail2(-2, "ssim backward: missing pointer");
    launch_bwd<true>(W % 4 == 0 && aligned16({ img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1 }), plan(B, CH, H, W),
                     (hipStream_t)stream, H, W, img1, img2, nullptr, dL_value, dL_scale, crop, dm_dmu1, dm_dsigma1_sq,
                     dm_dsigma12, dL_dimg1);
    HIP_TRY2(hipGetLastError());
    return 0;
}

}  // extern "C"
