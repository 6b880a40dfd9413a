// sks_ssim.hip -- fused SSIM forward / backward for MI355X (gfx950).
// Reference: submodules/fused-ssim/ssim.cu:187-444 (kernels), fused_ssim/__init__.py:8-41 (autograd wrapper).
//
// 11-tap separable Gaussian over five (forward) / three (backward) quantities per pixel.  The arithmetic, not HBM,
// is what this op has to get through (~110 FMA per pixel against 12..28 bytes), so the kernels are built around
// the vector ALU:
//   * 64x32 output tile per 256-thread workgroup: the 5-pixel halo costs 1.31x horizontal-pass work (1.72x traffic
//     at 32x32); the halo tile is fetched as aligned 16-byte pieces (columns x0-8 .. x0+71) when W % 4 == 0;
//   * both images interleaved in LDS as (img1, img2) pairs, the horizontal sums interleaved as (mu1, mu2) and
//     (E[x^2], E[y^2]) pairs: every tap of a pair is ONE packed FMA (v_pk_fma_f32) and every LDS access is 8 or 16
//     bytes; the vertical pass pairs two adjacent columns the same way;
//   * register blocking: 4 adjacent outputs per horizontal item (14 inputs), 2 columns x 4 rows per vertical thread;
//   * a block runs a sequence of jobs (channel, x-tile) and prefetches the next halo tile into registers while the
//     current one is convolved.
// Taps are accumulated left-to-right / top-to-bottom like ssim.cu:100-185 with explicit FMAs (nvcc contracts the
// reference's `sum += g * v` the same way); sigma = E[x^2] - mu^2 and the map / partial-derivative expressions are
// ssim.cu:262-283 as written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/skelsplat_hip.h"
#include "sks_err.h"
#include "sks_math.h"

namespace {

using sks::wave_sum_d;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int TW = 64, TH = 32, HALO = 5;
constexpr int IH = TH + 2 * HALO;        // 42 input rows
constexpr int IW = TW + 16;              // 80 input columns: x0-8 .. x0+71, a 16-byte aligned window around the halo
constexpr int NF4 = IH * (IW / 4);       // 840 float4 pieces per image per tile
constexpr int NLD = (NF4 + 255) / 256;   // pieces per thread (4)
constexpr int HG = TW / 4;               // horizontal items per row (16)

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

__device__ __forceinline__ v2f pk_fma(float g, v2f a, v2f c)
{
    const v2f gg = { g, g };
    return __builtin_elementwise_fma(gg, a, c);
}

struct Job {
    int c, tx;
};
__device__ __forceinline__ Job job_of(int j, int txb, int tiles_x)
{
    Job jb;
    jb.c = j / txb;
    jb.tx = blockIdx.x * txb + (j - jb.c * txb);
    if (jb.tx >= tiles_x) jb.tx = -1;
    return jb;
}

// one 16-byte piece of a halo row: columns x .. x+3 of `row` (nullptr = the row is outside the image)
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

// MODE bits: 1 = write ssim_map, 2 = write the three partial-derivative maps (train), 4 = add the sum of the map over the
// image shrunk by `crop` pixels per side to *total (fused_ssim()'s `.mean()` without a pass over the map)
template <bool VEC, int MODE>
__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, int CH, int txb, float C1, float C2,
                                                   const float* __restrict__ img1, const float* __restrict__ img2,
                                                   float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
                                                   float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
                                                   int crop, double* __restrict__ total)
{
    __shared__ __attribute__((aligned(16))) v2f s_in[IH][IW];                  // (img1, img2) per pixel
    __shared__ __attribute__((aligned(16))) v2f s_h01[IH][TW], s_h23[IH][TW];  // (mu1, mu2), (E11, E22) after x
    __shared__ __attribute__((aligned(16))) float s_h4[IH][TW];                // E12 after x
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW;
    const int y0 = blockIdx.y * TH;
    const int njobs = CH * txb;

    int prow[NLD], pcol[NLD];   // this thread's pieces: image row (-1 = outside / unused), first tile-local column
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int i = tid + k * 256;
        const int r = i / (IW / 4);
        const int y = y0 + r - HALO;
        pcol[k] = (i - r * (IW / 4)) * 4 - 8;
        prow[k] = (i < NF4 && y >= 0 && y < H) ? y : -1;
    }
    v4f r1[NLD], r2[NLD];
    auto fetch = [&](int j) {
        const Job jb = job_of(j, txb, tiles_x);
        const size_t plane = ((size_t)blockIdx.z * CH + jb.c) * H * W;
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const bool in = jb.tx >= 0 && prow[k] >= 0;
            const size_t ro = plane + (size_t)(in ? prow[k] : 0) * W;
            r1[k] = load_piece<VEC>(in ? img1 + ro : nullptr, jb.tx * TW + pcol[k], W);
            r2[k] = load_piece<VEC>(in ? img2 + ro : nullptr, jb.tx * TW + pcol[k], W);
        }
    };
    double acc = 0.0;
    fetch(0);
    for (int j = 0; j < njobs; j++) {
        const Job jb = job_of(j, txb, tiles_x);
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + k * 256;
            if (i < NF4) {
                v4f* dst = reinterpret_cast<v4f*>(&s_in[0][0]) + 2 * i;   // 4 pixels = 2 x 16 bytes
                dst[0] = (v4f){ r1[k].x, r2[k].x, r1[k].y, r2[k].y };
                dst[1] = (v4f){ r1[k].z, r2[k].z, r1[k].w, r2[k].w };
            }
        }
        __syncthreads();
        if (j + 1 < njobs) fetch(j + 1);   // in flight during the two passes below
        if (jb.tx >= 0) {
            for (int i = tid; i < IH * HG; i += 256) {   // horizontal pass (ssim.cu:100-164): 4 outputs from 14 inputs
                const int row = i / HG, g4 = (i - row * HG) * 4;
                const v4f* src = reinterpret_cast<const v4f*>(&s_in[row][g4 + 2]);   // tile-local columns g4-6 .. g4+9
                v2f in[16];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const v4f t = src[q];
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
                v4f* d01 = reinterpret_cast<v4f*>(&s_h01[row][g4]);
                v4f* d23 = reinterpret_cast<v4f*>(&s_h23[row][g4]);
                d01[0] = (v4f){ m[0].x, m[0].y, m[1].x, m[1].y };
                d01[1] = (v4f){ m[2].x, m[2].y, m[3].x, m[3].y };
                d23[0] = (v4f){ s[0].x, s[0].y, s[1].x, s[1].y };
                d23[1] = (v4f){ s[2].x, s[2].y, s[3].x, s[3].y };
                *reinterpret_cast<v4f*>(&s_h4[row][g4]) = (v4f){ c[0], c[1], c[2], c[3] };
            }
        }
        __syncthreads();
        if (jb.tx >= 0) {
            // vertical pass (ssim.cu:166-185, 218-260): columns 2cp, 2cp+1; rows 4rg .. 4rg+3
            const int cp = tid & 31, rg = tid >> 5;
            v2f Ma[4], Mb[4], Sa[4], Sb[4], Cc[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                Ma[o] = Mb[o] = Sa[o] = Sb[o] = Cc[o] = (v2f){ 0.0f, 0.0f };
            }
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const v4f a = *reinterpret_cast<const v4f*>(&s_h01[4 * rg + r][2 * cp]);
                const v4f b = *reinterpret_cast<const v4f*>(&s_h23[4 * rg + r][2 * cp]);
                const v2f e = *reinterpret_cast<const v2f*>(&s_h4[4 * rg + r][2 * cp]);
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int t = r - o;
                    if (t >= 0 && t < 11) {
                        Ma[o] = pk_fma(gk(t), (v2f){ a.x, a.y }, Ma[o]);
                        Mb[o] = pk_fma(gk(t), (v2f){ a.z, a.w }, Mb[o]);
                        Sa[o] = pk_fma(gk(t), (v2f){ b.x, b.y }, Sa[o]);
                        Sb[o] = pk_fma(gk(t), (v2f){ b.z, b.w }, Sb[o]);
                        Cc[o] = pk_fma(gk(t), e, Cc[o]);
                    }
                }
            }
            const size_t plane = ((size_t)blockIdx.z * CH + jb.c) * H * W;
            const int x = jb.tx * TW + 2 * cp;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int y = y0 + 4 * rg + o;
                float mv[2], d1[2], d2[2], d3[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float mu1 = h ? Mb[o].x : Ma[o].x, mu2 = h ? Mb[o].y : Ma[o].y;
                    const float e11 = h ? Sb[o].x : Sa[o].x, e22 = h ? Sb[o].y : Sa[o].y;
                    const float e12 = h ? Cc[o].y : Cc[o].x;
                    const float sigma1_sq = e11 - mu1 * mu1;
                    const float sigma2_sq = e22 - mu2 * mu2;
                    const float sigma12 = e12 - mu1 * mu2;
                    // ssim.cu:262-283
                    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
                    const float Cn = (2.0f * mu1_mu2 + C1);
                    const float D = (2.0f * sigma12 + C2);
                    const float A = (mu1_sq + mu2_sq + C1);
                    const float B = (sigma1_sq + sigma2_sq + C2);
                    mv[h] = (Cn * D) / (A * B);
                    if (MODE & 2) {
                        d1[h] = ((mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * Cn) / (A * B) - (mu1 * 2.0f * Cn * D) / (A * A * B) +
                                 (mu1 * 2.0f * Cn * D) / (A * B * B));
                        d2[h] = ((-Cn * D) / (A * B * B));
                        d3[h] = ((2 * Cn) / (A * B));
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
        __syncthreads();   // the next job's staging writes must not overtake this job's LDS reads
    }
    if (MODE & 4) {
        __shared__ double s_red[4];
        acc = wave_sum_d(acc);
        if ((tid & 63) == 0) s_red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) atomicAdd(total, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    }
}

// backward (ssim.cu:288-366): dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12)
// UNIFORM: dL_dmap is one value (*dL_value) inside the image shrunk by `crop` pixels per side and zero outside --
// what `.mean()` of the ("valid"-cropped) map hands back -- so no gradient image is materialised or read.
template <bool VEC, bool UNIFORM>
__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, int CH, int txb, const float* __restrict__ img1,
                                                   const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                   const float* __restrict__ dL_value, int crop,
                                                   const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                   const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1)
{
    __shared__ __attribute__((aligned(16))) v2f s_in01[IH][IW];   // (dL dm_dmu1, dL dm_dsigma1_sq)
    __shared__ __attribute__((aligned(16))) float s_in2[IH][IW];  // dL dm_dsigma12
    __shared__ __attribute__((aligned(16))) v2f s_h01[IH][TW];
    __shared__ __attribute__((aligned(16))) float s_h2[IH][TW];
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW;
    const int y0 = blockIdx.y * TH;
    const int njobs = CH * txb;
    const float dval = UNIFORM ? *dL_value : 0.0f;

    int prow[NLD], pcol[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int i = tid + k * 256;
        const int r = i / (IW / 4);
        const int y = y0 + r - HALO;
        pcol[k] = (i - r * (IW / 4)) * 4 - 8;
        prow[k] = (i < NF4 && y >= (UNIFORM ? crop : 0) && y < H - (UNIFORM ? crop : 0)) ? y : -1;
    }
    v4f rd[NLD], r0[NLD], r1[NLD], r2[NLD];
    auto fetch = [&](int j) {
        const Job jb = job_of(j, txb, tiles_x);
        const size_t plane = ((size_t)blockIdx.z * CH + jb.c) * H * W;
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const bool in = jb.tx >= 0 && prow[k] >= 0;
            const size_t ro = plane + (size_t)(in ? prow[k] : 0) * W;
            const int x = jb.tx * TW + pcol[k];
            if (UNIFORM) {
                rd[k] = (v4f){ (in && x >= crop && x < W - crop) ? dval : 0.0f,
                               (in && x + 1 >= crop && x + 1 < W - crop) ? dval : 0.0f,
                               (in && x + 2 >= crop && x + 2 < W - crop) ? dval : 0.0f,
                               (in && x + 3 >= crop && x + 3 < W - crop) ? dval : 0.0f };
            } else {
                rd[k] = load_piece<VEC>(in ? dL_dmap + ro : nullptr, x, W);
            }
            r0[k] = load_piece<VEC>(in ? dm_dmu1 + ro : nullptr, x, W);
            r1[k] = load_piece<VEC>(in ? dm_dsigma1_sq + ro : nullptr, x, W);
            r2[k] = load_piece<VEC>(in ? dm_dsigma12 + ro : nullptr, x, W);
        }
    };
    fetch(0);
    for (int j = 0; j < njobs; j++) {
        const Job jb = job_of(j, txb, tiles_x);
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + k * 256;
            if (i < NF4) {
                const v4f a = r0[k] * rd[k], b = r1[k] * rd[k];
                v4f* dst = reinterpret_cast<v4f*>(&s_in01[0][0]) + 2 * i;
                dst[0] = (v4f){ a.x, b.x, a.y, b.y };
                dst[1] = (v4f){ a.z, b.z, a.w, b.w };
                reinterpret_cast<v4f*>(&s_in2[0][0])[i] = r2[k] * rd[k];
            }
        }
        __syncthreads();
        if (j + 1 < njobs) fetch(j + 1);
        if (jb.tx >= 0) {
            for (int i = tid; i < IH * HG; i += 256) {   // horizontal pass (ssim.cu:318-340)
                const int row = i / HG, g4 = (i - row * HG) * 4;
                const v4f* src = reinterpret_cast<const v4f*>(&s_in01[row][g4 + 2]);   // tile-local columns g4-6 .. g4+9
                const v4f* src2 = reinterpret_cast<const v4f*>(&s_in2[row][g4]);        // tile-local columns g4-8 .. g4+11
                v2f in[16];
                float in2[20];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const v4f t = src[q];
                    in[2 * q] = (v2f){ t.x, t.y };
                    in[2 * q + 1] = (v2f){ t.z, t.w };
                }
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const v4f t = src2[q];
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
                v4f* d01 = reinterpret_cast<v4f*>(&s_h01[row][g4]);
                d01[0] = (v4f){ m[0].x, m[0].y, m[1].x, m[1].y };
                d01[1] = (v4f){ m[2].x, m[2].y, m[3].x, m[3].y };
                *reinterpret_cast<v4f*>(&s_h2[row][g4]) = (v4f){ c[0], c[1], c[2], c[3] };
            }
        }
        __syncthreads();
        if (jb.tx >= 0) {
            const int cp = tid & 31, rg = tid >> 5;   // vertical pass (ssim.cu:342-365)
            v2f Aa[4], Ab[4], Cc[4];
#pragma unroll
            for (int o = 0; o < 4; o++) Aa[o] = Ab[o] = Cc[o] = (v2f){ 0.0f, 0.0f };
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const v4f a = *reinterpret_cast<const v4f*>(&s_h01[4 * rg + r][2 * cp]);
                const v2f e = *reinterpret_cast<const v2f*>(&s_h2[4 * rg + r][2 * cp]);
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int t = r - o;
                    if (t >= 0 && t < 11) {
                        Aa[o] = pk_fma(gk(t), (v2f){ a.x, a.y }, Aa[o]);
                        Ab[o] = pk_fma(gk(t), (v2f){ a.z, a.w }, Ab[o]);
                        Cc[o] = pk_fma(gk(t), e, Cc[o]);
                    }
                }
            }
            const size_t plane = ((size_t)blockIdx.z * CH + jb.c) * H * W;
            const int x = jb.tx * TW + 2 * cp;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int y = y0 + 4 * rg + o;
                if (y < H && x < W) {
                    const size_t gi = plane + (size_t)y * W + x;
                    if (VEC) {
                        const v2f p1 = *reinterpret_cast<const v2f*>(img1 + gi);
                        const v2f p2 = *reinterpret_cast<const v2f*>(img2 + gi);
                        v2f o2;
                        o2.x = (0.0f + Aa[o].x + p1.x * 2.0f * Aa[o].y) + p2.x * Cc[o].x;
                        o2.y = (0.0f + Ab[o].x + p1.y * 2.0f * Ab[o].y) + p2.y * Cc[o].y;
                        *reinterpret_cast<v2f*>(dL_dimg1 + gi) = o2;
                    } else {
                        dL_dimg1[gi] = (0.0f + Aa[o].x + img1[gi] * 2.0f * Aa[o].y) + img2[gi] * Cc[o].x;
                        if (x + 1 < W)
                            dL_dimg1[gi + 1] = (0.0f + Ab[o].x + img1[gi + 1] * 2.0f * Ab[o].y) + img2[gi + 1] * Cc[o].y;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// x-tiles per block: every block runs CH * txb pipelined jobs; keep >= ~3000 blocks on the chip when the image allows
int tiles_per_block(int tiles_x, int tiles_y, int B, int CH)
{
    int txb = 1;
    while (txb < 8 && CH * txb < 4 && (long long)((tiles_x + 2 * txb - 1) / (2 * txb)) * tiles_y * B >= 3000) txb *= 2;
    return txb;
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

template <int MODE>
void launch_fwd(bool vec, dim3 grid, hipStream_t st, int H, int W, int CH, int txb, float C1, float C2, const float* img1,
                const float* img2, float* ssim_map, float* d1, float* d2, float* d3, int crop, double* total)
{
    if (vec)
        hipLaunchKernelGGL((k_ssim_fwd<true, MODE>), grid, dim3(256), 0, st, H, W, CH, txb, C1, C2, img1, img2, ssim_map, d1, d2,
                           d3, crop, total);
    else
        hipLaunchKernelGGL((k_ssim_fwd<false, MODE>), grid, dim3(256), 0, st, H, W, CH, txb, C1, C2, img1, img2, ssim_map, d1,
                           d2, d3, crop, total);
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
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int txb = tiles_per_block(tiles_x, tiles_y, B, CH);
    const dim3 grid((tiles_x + txb - 1) / txb, tiles_y, B);
    const bool vec = W % 4 == 0 && aligned16({ img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 });
    if (dm_dmu1)
        launch_fwd<3>(vec, grid, (hipStream_t)stream, H, W, CH, txb, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq,
                      dm_dsigma12, 0, nullptr);
    else
        launch_fwd<1>(vec, grid, (hipStream_t)stream, H, W, CH, txb, C1, C2, img1, img2, ssim_map, nullptr, nullptr, nullptr, 0,
                      nullptr);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_sum(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, int crop,
                       float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, double* total, void* stream)
{
    if (int rc = check_shape("ssim sum", B, CH, H, W)) return rc;
    if (crop < 0) return fail2(-1, "ssim sum: crop negative");
    if (!total) return fail2(-2, "ssim sum: missing pointer");
    HIP_TRY2(hipMemsetAsync(total, 0, sizeof(double), (hipStream_t)stream));
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))
        return fail2(-2, "ssim sum: provide all three partial-derivative maps or none");
    if (B * CH == 0) return 0;
    if (!img1 || !img2) return fail2(-2, "ssim sum: missing pointer");
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int txb = tiles_per_block(tiles_x, tiles_y, B, CH);
    const dim3 grid((tiles_x + txb - 1) / txb, tiles_y, B);
    const bool vec = W % 4 == 0 && aligned16({ img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 });
    if (dm_dmu1)
        launch_fwd<6>(vec, grid, (hipStream_t)stream, H, W, CH, txb, C1, C2, img1, img2, nullptr, dm_dmu1, dm_dsigma1_sq,
                      dm_dsigma12, crop, total);
    else
        launch_fwd<4>(vec, grid, (hipStream_t)stream, H, W, CH, txb, C1, C2, img1, img2, nullptr, nullptr, nullptr, nullptr,
                      crop, total);
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
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int txb = tiles_per_block(tiles_x, tiles_y, B, CH);
    const dim3 grid((tiles_x + txb - 1) / txb, tiles_y, B);
    if (W % 4 == 0 && aligned16({ img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1 }))
        hipLaunchKernelGGL((k_ssim_bwd<true, false>), grid, dim3(256), 0, (hipStream_t)stream, H, W, CH, txb, img1, img2, dL_dmap,
                           nullptr, 0, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    else
        hipLaunchKernelGGL((k_ssim_bwd<false, false>), grid, dim3(256), 0, (hipStream_t)stream, H, W, CH, txb, img1, img2,
                           dL_dmap, nullptr, 0, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_bwd_uniform(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_value,
                               int crop, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                               float* dL_dimg1, void* stream)
{
    if (int rc = check_shape("ssim backward", B, CH, H, W)) return rc;
    if (crop < 0) return fail2(-1, "ssim backward: crop negative");
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !dL_value || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail2(-2, "ssim backward: missing pointer");
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int txb = tiles_per_block(tiles_x, tiles_y, B, CH);
    const dim3 grid((tiles_x + txb - 1) / txb, tiles_y, B);
    if (W % 4 == 0 && aligned16({ img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1 }))
        hipLaunchKernelGGL((k_ssim_bwd<true, true>), grid, dim3(256), 0, (hipStream_t)stream, H, W, CH, txb, img1, img2, nullptr,
                           dL_value, crop, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    else
        hipLaunchKernelGGL((k_ssim_bwd<false, true>), grid, dim3(256), 0, (hipStream_t)stream, H, W, CH, txb, img1, img2, nullptr,
                           dL_value, crop, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    HIP_TRY2(hipGetLastError());
    return 0;
}

}  // extern "C"
