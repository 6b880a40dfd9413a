// sks_ops.hip -- the smaller ops of the hot path for MI355X (gfx950):
//   * fused masked-L2 heat-map loss + gradient          (reference: utils/loss_utils.py:86-100, train.py:150-161)
//   * fused SSIM forward / backward                      (reference: submodules/fused-ssim/ssim.cu:187-444)
//   * mean squared distance to the 3 nearest neighbours  (reference: submodules/simple-knn/simple_knn.cu:132-222)
// All HBM-bound elementwise / stencil work: 16-byte coalesced accesses, LDS-staged tiles, no MFMA.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/skelsplat_hip.h"

// the error text lives in sks_raster.hip's thread-local buffer (sks_last_error); this TU reports through it
extern "C" void sks_set_error_(const char* msg);

namespace {

thread_local char g_err2[512] = "";

int fail2(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err2, sizeof(g_err2), fmt, ap);
    va_end(ap);
    sks_set_error_(g_err2);
    return code;
}

#define HIP_TRY2(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail2((int)e_, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// masked L2: per view  N = #{gt > 0 or render > 0},  S = sum over that mask of (render - gt)^2,
// dL = 2 (render - gt) on the mask (NOT divided by N: the caller scales the parameter gradients by 1/N, which is
// exact because everything downstream of dL/d(render) is linear in it).  One pass: read render + gt, write dL.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_masked_l2(size_t n, const float* __restrict__ render, const float* __restrict__ gt,
                                                    float* __restrict__ dL, double* __restrict__ sums)
{
    __shared__ double s_red[2][4];
    const int v = blockIdx.y, tid = threadIdx.x;
    const float* r = render + (size_t)v * n;
    const float* g = gt + (size_t)v * n;
    float* d = dL ? dL + (size_t)v * n : nullptr;
    double S = 0.0, N = 0.0;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(r)[i];
        const float4 b = reinterpret_cast<const float4*>(g)[i];
        float4 o;
        float e;
        bool m;
        m = b.x > 0.0f || a.x > 0.0f; e = a.x - b.x; o.x = m ? 2.0f * e : 0.0f; if (m) { S += (double)(e * e); N += 1.0; }
        m = b.y > 0.0f || a.y > 0.0f; e = a.y - b.y; o.y = m ? 2.0f * e : 0.0f; if (m) { S += (double)(e * e); N += 1.0; }
        m = b.z > 0.0f || a.z > 0.0f; e = a.z - b.z; o.z = m ? 2.0f * e : 0.0f; if (m) { S += (double)(e * e); N += 1.0; }
        m = b.w > 0.0f || a.w > 0.0f; e = a.w - b.w; o.w = m ? 2.0f * e : 0.0f; if (m) { S += (double)(e * e); N += 1.0; }
        if (d) reinterpret_cast<float4*>(d)[i] = o;
    }
    if (blockIdx.x == 0) {  // scalar tail (n % 4 elements)
        for (size_t i = n4 * 4 + tid; i < n; i += 256) {
            const float a = r[i], b = g[i];
            const bool m = b > 0.0f || a > 0.0f;
            const float e = a - b;
            if (d) d[i] = m ? 2.0f * e : 0.0f;
            if (m) { S += (double)(e * e); N += 1.0; }
        }
    }
    S = wave_sum_d(S);
    N = wave_sum_d(N);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = S; s_red[1][tid >> 6] = N; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&sums[2 * v], (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
        atomicAdd(&sums[2 * v + 1], (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    }
}

// ------------------------------------------------------------------------------------------------------------
// fused SSIM.  32x32 output tile per 256-thread workgroup (4 rows per thread), 11-tap separable Gaussian,
// zero ("same") padding like get_pix_value (ssim.cu:36-42).  Tap order and the sigma = E[x^2] - mu^2 form follow
// ssim.cu:100-185, 218-283.  grid (ceil(W/32), ceil(H/32), B*CH).
// ------------------------------------------------------------------------------------------------------------
__constant__ float c_gauss[11] = { 0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                   0.10936068743467331f,  0.21300552785396576f,  0.26601171493530273f,
                                   0.21300552785396576f,  0.10936068743467331f,  0.036000773310661316f,
                                   0.0075987582094967365f, 0.001028380123898387f };  // ssim.cu:9-19
constexpr int ST = 32, SH = ST + 10;

__device__ __forceinline__ float pix_or_zero(const float* __restrict__ img, int y, int x, int H, int W)
{
    return (x >= 0 && y >= 0 && x < W && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

template <int NQ>
__device__ __forceinline__ void conv_y(const float (*hx)[SH][ST], int ly, int lx, float (&out)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        float val = 0.0f;
#pragma unroll
        for (int t = 0; t < 11; t++) val += c_gauss[t] * hx[q][ly + t][lx];
        out[q] = val;
    }
}

__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                   const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                   float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                   float* __restrict__ dm_dsigma12)
{
    __shared__ float p1[SH][SH + 1], p2[SH][SH + 1];
    __shared__ float hx[5][SH][ST];
    const int tid = threadIdx.x;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* a = img1 + plane;
    const float* b = img2 + plane;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    for (int i = tid; i < SH * SH; i += 256) {
        const int ly = i / SH, lx = i - ly * SH;
        p1[ly][lx] = pix_or_zero(a, y0 + ly - 5, x0 + lx - 5, H, W);
        p2[ly][lx] = pix_or_zero(b, y0 + ly - 5, x0 + lx - 5, H, W);
    }
    __syncthreads();
    for (int i = tid; i < SH * ST; i += 256) {  // horizontal pass (ssim.cu:100-164)
        const int ly = i / ST, lx = i - ly * ST;
        float m1 = 0.0f, m2 = 0.0f, s11 = 0.0f, s22 = 0.0f, s12 = 0.0f;
#pragma unroll
        for (int t = 0; t < 11; t++) {
            const float u = p1[ly][lx + t], w = p2[ly][lx + t], gk = c_gauss[t];
            m1 += gk * u;
            m2 += gk * w;
            s11 += gk * (u * u);
            s22 += gk * (w * w);
            s12 += gk * (u * w);
        }
        hx[0][ly][lx] = m1; hx[1][ly][lx] = m2; hx[2][ly][lx] = s11; hx[3][ly][lx] = s22; hx[4][ly][lx] = s12;
    }
    __syncthreads();
    const int lx = tid & 31;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = (tid >> 5) + 8 * r;
        const int x = x0 + lx, y = y0 + ly;
        float q[5];
        conv_y<5>(hx, ly, lx, q);
        const float mu1 = q[0], mu2 = q[1];
        const float sigma1_sq = q[2] - mu1 * mu1;
        const float sigma2_sq = q[3] - mu2 * mu2;
        const float sigma12 = q[4] - mu1 * mu2;
        // ssim.cu:262-283
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const float Cc = (2.0f * mu1_mu2 + C1);
        const float D = (2.0f * sigma12 + C2);
        const float A = (mu1_sq + mu2_sq + C1);
        const float B = (sigma1_sq + sigma2_sq + C2);
        const float m = (Cc * D) / (A * B);
        if (x < W && y < H) {
            const size_t gi = plane + (size_t)y * W + x;
            ssim_map[gi] = m;
            if (dm_dmu1) {
                dm_dmu1[gi] = ((mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * Cc) / (A * B) - (mu1 * 2.0f * Cc * D) / (A * A * B) +
                               (mu1 * 2.0f * Cc * D) / (A * B * B));
                dm_dsigma1_sq[gi] = ((-Cc * D) / (A * B * B));
                dm_dsigma12[gi] = ((2 * Cc) / (A * B));
            }
        }
    }
}

// backward (ssim.cu:288-366): dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12)
__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                   const float* __restrict__ dL_dmap, const float* __restrict__ dm_dmu1,
                                                   const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                   float* __restrict__ dL_dimg1)
{
    __shared__ float pq[3][SH][SH + 1];
    __shared__ float hx[3][SH][ST];
    const int tid = threadIdx.x;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    for (int i = tid; i < SH * SH; i += 256) {
        const int ly = i / SH, lx = i - ly * SH;
        const int y = y0 + ly - 5, x = x0 + lx - 5;
        const float d = pix_or_zero(dL_dmap + plane, y, x, H, W);
        pq[0][ly][lx] = pix_or_zero(dm_dmu1 + plane, y, x, H, W) * d;
        pq[1][ly][lx] = pix_or_zero(dm_dsigma1_sq + plane, y, x, H, W) * d;
        pq[2][ly][lx] = pix_or_zero(dm_dsigma12 + plane, y, x, H, W) * d;
    }
    __syncthreads();
    for (int i = tid; i < SH * ST; i += 256) {
        const int ly = i / ST, lx = i - ly * ST;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float val = 0.0f;
#pragma unroll
            for (int t = 0; t < 11; t++) val += c_gauss[t] * pq[q][ly][lx + t];
            hx[q][ly][lx] = val;
        }
    }
    __syncthreads();
    const int lx = tid & 31;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = (tid >> 5) + 8 * r;
        const int x = x0 + lx, y = y0 + ly;
        float q[3];
        conv_y<3>(hx, ly, lx, q);
        if (x < W && y < H) {
            const size_t gi = plane + (size_t)y * W + x;
            float dL_dpix = 0.0f;
            dL_dpix += q[0];
            dL_dpix += img1[gi] * 2.0f * q[1];
            dL_dpix += img2[gi] * q[2];
            dL_dimg1[gi] = dL_dpix;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 3-NN mean squared distance.  The reference's Morton/box search is conservative (simple_knn.cu:169-182), i.e. it
// returns the exact 3 nearest neighbours; this is the exact search as an LDS-tiled all-pairs sweep with the same
// updateKBest insertion (simple_knn.cu:132-146) and the same final (b0+b1+b2)/3 (:183).  O(P^2), intended for the
// skeleton-sized clouds of this pipeline (P = 15..19 per skeleton; fine up to ~1e5 points).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn3(int P, const float* __restrict__ pts, float* __restrict__ out)
{
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0, py = 0, pz = 0;
    if (i < P) { px = pts[3 * i]; py = pts[3 * i + 1]; pz = pts[3 * i + 2]; }
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    for (int j0 = 0; j0 < P; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < P) { sx[threadIdx.x] = pts[3 * j]; sy[threadIdx.x] = pts[3 * j + 1]; sz[threadIdx.x] = pts[3 * j + 2]; }
        __syncthreads();
        const int cnt = min(256, P - j0);
        if (i < P) {
            for (int k = 0; k < cnt; k++) {
                if (j0 + k == i) continue;
                const float dx = sx[k] - px, dy = sy[k] - py, dz = sz[k] - pz;
                float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (best[q] > dist) { const float t = best[q]; best[q] = dist; dist = t; }
                }
            }
        }
    }
    if (i < P) out[i] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

extern "C" {

int sks_masked_l2(int V, size_t n_per_view, const float* render, const float* gt, float* dL_unscaled, double* sums,
                  void* stream)
{
    if (V < 1 || !render || !gt || !sums) return fail2(-2, "masked_l2: missing pointer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY2(hipMemsetAsync(sums, 0, (size_t)V * 2 * sizeof(double), st));
    if (n_per_view == 0) return 0;
    size_t blocks = (n_per_view / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_masked_l2, dim3((unsigned)blocks, V), dim3(256), 0, st, n_per_view, render, gt, dL_unscaled, sums);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream)
{
    if (B < 0 || CH < 0 || H < 1 || W < 1) return fail2(-1, "ssim: bad shape");
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !ssim_map) return fail2(-2, "ssim: missing pointer");
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))
        return fail2(-2, "ssim: provide all three partial-derivative maps or none");
    dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * CH);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(256), 0, (hipStream_t)stream, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_fused_ssim_bwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                       float* dL_dimg1, void* stream)
{
    (void)C1; (void)C2;
    if (B < 0 || CH < 0 || H < 1 || W < 1) return fail2(-1, "ssim: bad shape");
    if (B * CH == 0) return 0;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail2(-2, "ssim backward: missing pointer");
    dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * CH);
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), 0, (hipStream_t)stream, H, W, img1, img2, dL_dmap, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_knn3_meandist2(int P, const float* points, float* mean_dist2, void* stream)
{
    if (P < 0) return fail2(-1, "knn: P negative");
    if (P == 0) return 0;
    if (!points || !mean_dist2) return fail2(-2, "knn: missing pointer");
    hipLaunchKernelGGL(k_knn3, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, points, mean_dist2);
    HIP_TRY2(hipGetLastError());
    return 0;
}

}  // extern "C"
