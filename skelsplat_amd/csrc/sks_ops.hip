// sks_ops.hip -- the smaller ops of the hot path for MI355X (gfx950):
//   * fused masked-L2 heat-map loss + gradient          (reference: utils/loss_utils.py:86-100, train.py:150-161)
//   * mean squared distance to the 3 nearest neighbours  (reference: submodules/simple-knn/simple_knn.cu:132-222)
//   * pseudo-GT heat-map planes                          (reference: utils/general_utils.py:175-304)
// (fused SSIM: sks_ssim.hip)
// All HBM-bound elementwise / stencil work: 16-byte coalesced accesses, LDS-staged tiles, no MFMA.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#include "../../include/skelsplat_hip.h"
#include "sks_err.h"
#include "sks_math.h"

namespace {

using sks::wave_sum_d;
typedef float hm_v4f __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// masked L2: per view  N = #{gt > 0 or render > 0},  S = sum over that mask of (render - gt)^2,
// dL = 2 (render - gt) on the mask (NOT divided by N: the caller scales the parameter gradients by 1/N, which is
// exact because everything downstream of dL/d(render) is linear in it).  One pass: read render + gt, write dL.
// ------------------------------------------------------------------------------------------------------------
// Launch shape: at most L2_BLOCKS workgroups of 1024 threads per view.  Every
// workgroup ends with two atomics on its view's {S, N} -- all views' sums share a cache line or two, and atomics on one line are
// executed one after the other at the memory side (~15 ns each, measured: with 2 048 workgroups of 256 threads per view the
// kernel took 63 us for one H36M view, L2-resident, and exactly as long without writing dL -- 4 096 atomics in a row; 234 us for
// four views).  256 x 2 per view disappear behind the streaming.
constexpr int L2_BLOCKS = 256;
constexpr int L2_THREADS = 1024;
__global__ __launch_bounds__(L2_THREADS) void k_masked_l2(size_t n, const float* __restrict__ render, const float* __restrict__ gt,
                                                           float* __restrict__ dL, double* __restrict__ sums)
{
    __shared__ double s_red[2][L2_THREADS / 64];
    const int v = blockIdx.y, tid = threadIdx.x;
    const float* r = render + (size_t)v * n;
    const float* g = gt + (size_t)v * n;
    float* d = dL ? dL + (size_t)v * n : nullptr;
    // Per thread: the four squared errors of one 16-byte group are added in fp32 (three roundings on a sum of four terms), every
    // group's sum goes into a DOUBLE accumulator -- one conversion and one fp64 add per four elements, nothing next to the three
    // memory streams -- so the loss (and the early-stopping decision made from it, train.py:155) sits within ~1e-7 of the
    // reference's all-at-once `error[mask].mean()` whatever the launch shape; the mask count is an integer
    double Sd = 0.0;
    unsigned Nu = 0u;
    const size_t n4 = n / 4;
    auto one = [&](float a, float b, float& acc) -> float {
        const bool m = b > 0.0f || a > 0.0f;
        const float e = m ? a - b : 0.0f;
        acc = __builtin_fmaf(e, e, acc);
        Nu += m ? 1u : 0u;
        return 2.0f * e;
    };
    auto four = [&](const float4& a, const float4& b) -> float4 {
        float acc = 0.0f;
        const float4 o = make_float4(one(a.x, b.x, acc), one(a.y, b.y, acc), one(a.z, b.z, acc), one(a.w, b.w, acc));
        Sd += (double)acc;
        return o;
    };
    // workgroup-strided: the workgroups running at one time read and write one contiguous window of memory (a contiguous piece per
    // workgroup -- 512 separate streams -- measured 5 % slower at 31 x 1920 x 1080)
    const size_t stride = (size_t)gridDim.x * L2_THREADS;
    size_t i = (size_t)blockIdx.x * L2_THREADS + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four float4 pairs in flight per trip
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = reinterpret_cast<const float4*>(r)[i + u * stride];
            b[u] = reinterpret_cast<const float4*>(g)[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 o = four(a[u], b[u]);
            if (d) __builtin_nontemporal_store((hm_v4f){ o.x, o.y, o.z, o.w }, reinterpret_cast<hm_v4f*>(d) + i + u * stride);
        }
    }
    for (; i < n4; i += stride) {
        const float4 o = four(reinterpret_cast<const float4*>(r)[i], reinterpret_cast<const float4*>(g)[i]);
        if (d) reinterpret_cast<float4*>(d)[i] = o;
    }
    if (blockIdx.x == 0) {  // scalar tail (n % 4 elements)
        for (size_t j = n4 * 4 + tid; j < n; j += L2_THREADS) {
            float acc = 0.0f;
            const float o = one(r[j], g[j], acc);
            Sd += (double)acc;
            if (d) d[j] = o;
        }
    }
    const double S = wave_sum_d(Sd), N = wave_sum_d((double)Nu);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = S; s_red[1][tid >> 6] = N; }
    __syncthreads();
    if (tid == 0) {
        double tS = 0.0, tN = 0.0;
        for (int w = 0; w < L2_THREADS / 64; w++) { tS += s_red[0][w]; tN += s_red[1][w]; }
        if (tN != 0.0) {   // (N == 0: nothing in the mask, S == 0 too)
            atomicAdd(&sums[2 * v], tS);
            atomicAdd(&sums[2 * v + 1], tN);
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


// ------------------------------------------------------------------------------------------------------------
// 3-NN mean squared distance for LARGE clouds: exact search over a uniform grid (the reference sorts by Morton code and
// prunes 1024-point boxes, simple_knn.cu:150-222; both are conservative, i.e. exact).  Points are bucketed into
// ~2-per-cell cells (count -> scan -> scatter), then every point walks the cell shells around its own cell, keeping
// the 3 best squared distances with the reference's updateKBest insertion, until the 3rd best cannot be beaten by
// anything outside the shells already visited.  Distances are evaluated exactly like the all-pairs kernel, so the
// two paths return identical floats.
// ------------------------------------------------------------------------------------------------------------
struct KnnGrid {
    int n;             // cells per axis
    float* part;       // [256][6] partial bounding boxes
    uint32_t* start;   // n^3 + 1 exclusive cell offsets
    uint32_t* cursor;  // n^3
    float4* sorted;    // P points in cell order, .w = original index (bits)
};
inline int knn_cells_per_axis(int P)
{
    int n = (int)ceil(cbrt((double)P / 2.0));   // ~2 points per cell on average: dense regions of a non-uniform cloud
    return n < 1 ? 1 : (n > 160 ? 160 : n);      // stay cheap, at the price of walking more (empty) cells elsewhere
}
inline size_t knn_scratch(int P, KnnGrid* g, void* base)
{
    const int n = knn_cells_per_axis(P);
    const size_t nc = (size_t)n * n * n;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_part = take(256 * 6 * 4), o_start = take((nc + 1) * 4), o_cur = take(nc * 4), o_sorted = take((size_t)P * 16);
    if (g) {
        g->n = n;
        g->part = (float*)((char*)base + o_part);
        g->start = (uint32_t*)((char*)base + o_start);
        g->cursor = (uint32_t*)((char*)base + o_cur);
        g->sorted = (float4*)((char*)base + o_sorted);
    }
    return off;
}

__global__ __launch_bounds__(256) void k_knn_bbox(int P, const float* __restrict__ pts, float* __restrict__ part)
{
    __shared__ float s[6][256];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
        for (int k = 0; k < 3; k++) { const float v = pts[3 * i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
    for (int k = 0; k < 3; k++) { s[k][threadIdx.x] = lo[k]; s[3 + k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; k++) {
                s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + o]);
                s[3 + k][threadIdx.x] = fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + o]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

struct KnnBox { float lo[3], inv[3], h; };   // inv = cells per unit length, h = smallest cell edge

// every block folds the <= 256 partial boxes itself (cheaper than another launch)
__device__ __forceinline__ KnnBox knn_box(const float* __restrict__ part, int nparts, int n, float* s6)
{
    if (threadIdx.x < 6) {
        float v = part[threadIdx.x];
        for (int b = 1; b < nparts; b++) v = threadIdx.x < 3 ? fminf(v, part[b * 6 + threadIdx.x]) : fmaxf(v, part[b * 6 + threadIdx.x]);
        s6[threadIdx.x] = v;
    }
    __syncthreads();
    KnnBox bx;
    bx.h = FLT_MAX;
    for (int k = 0; k < 3; k++) {
        bx.lo[k] = s6[k];
        const float ext = fmaxf(s6[3 + k] - s6[k], 1e-30f);
        bx.inv[k] = (float)n / ext;
        bx.h = fminf(bx.h, ext / (float)n);
    }
    return bx;
}
__device__ __forceinline__ void knn_cell(const KnnBox& bx, int n, float x, float y, float z, int (&c)[3])
{
    const float p[3] = { x, y, z };
    for (int k = 0; k < 3; k++) c[k] = min(n - 1, max(0, (int)((p[k] - bx.lo[k]) * bx.inv[k])));
}

__global__ __launch_bounds__(256) void k_knn_count(int P, const float* __restrict__ pts, KnnGrid g, int nparts)
{
    __shared__ float s6[6];
    const KnnBox bx = knn_box(g.part, nparts, g.n, s6);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    int c[3];
    knn_cell(bx, g.n, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], c);
    atomicAdd(&g.cursor[(c[2] * g.n + c[1]) * g.n + c[0]], 1u);
}

__global__ __launch_bounds__(1024) void k_knn_scan(int ncell, KnnGrid g)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (ncell + 1023) / 1024;
    const int b = min(ncell, tid * per), e = min(ncell, b + per);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += g.cursor[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t t = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    uint32_t run = tid ? s_part[tid - 1] : 0;
    for (int i = b; i < e; i++) {
        const uint32_t c = g.cursor[i];
        g.start[i] = run;
        g.cursor[i] = run;    // scatter cursor
        run += c;
    }
    if (tid == 1023) g.start[ncell] = s_part[1023];
}

__global__ __launch_bounds__(256) void k_knn_scatter(int P, const float* __restrict__ pts, KnnGrid g, int nparts)
{
    __shared__ float s6[6];
    const KnnBox bx = knn_box(g.part, nparts, g.n, s6);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    int c[3];
    knn_cell(bx, g.n, x, y, z, c);
    const uint32_t pos = atomicAdd(&g.cursor[(c[2] * g.n + c[1]) * g.n + c[0]], 1u);
    g.sorted[pos] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
}

__global__ __launch_bounds__(256) void k_knn_search(int P, KnnGrid g, int nparts, float* __restrict__ out)
{
    __shared__ float s6[6];
    const KnnBox bx = knn_box(g.part, nparts, g.n, s6);
    const int si = blockIdx.x * 256 + threadIdx.x;   // position in cell order: neighbouring threads share cells
    if (si >= P) return;
    const float4 me = g.sorted[si];
    const int n = g.n;
    int c[3];
    knn_cell(bx, n, me.x, me.y, me.z, c);
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    for (int r = 0; r < n; r++) {
        // shell r: cells at Chebyshev distance exactly r from (c0, c1, c2)
        const int z0 = max(0, c[2] - r), z1 = min(n - 1, c[2] + r);
        const int y0 = max(0, c[1] - r), y1 = min(n - 1, c[1] + r);
        const int x0 = max(0, c[0] - r), x1 = min(n - 1, c[0] + r);
        for (int zz = z0; zz <= z1; zz++)
            for (int yy = y0; yy <= y1; yy++) {
                const bool face = abs(zz - c[2]) == r || abs(yy - c[1]) == r;
                for (int xx = x0; xx <= x1; xx += (face || r == 0) ? 1 : max(1, x1 - x0)) {
                    if (!face && abs(xx - c[0]) != r) continue;   // interior of the cube was visited by earlier shells
                    const int cell = (zz * n + yy) * n + xx;
                    const uint32_t s0 = g.start[cell], s1 = g.start[cell + 1];
                    for (uint32_t j = s0; j < s1; j++) {
                        if ((int)j == si) continue;
                        const float4 q = g.sorted[j];
                        const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
                        float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            if (best[k] > dist) { const float t = best[k]; best[k] = dist; dist = t; }
                        }
                    }
                }
            }
        // everything not visited yet is at least (r - slack) * h away (slack covers the rounding of the cell index)
        const float reach = ((float)r - 1e-3f) * bx.h;
        if (reach > 0.0f && best[2] <= reach * reach) break;
    }
    out[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

namespace {
// ------------------------------------------------------------------------------------------------------------
// heat-map planes: out[y][x] = (row[y] * col[x] - cmin) / den, one streaming write.  grid (x chunks of 1024 px,
// 16-row bands, V*J); a thread keeps its 4 column weights in registers and walks the band's rows (the row weight is
// wave-uniform -> scalar load); 16-byte non-temporal stores.
// ------------------------------------------------------------------------------------------------------------
// TOTALS: also accumulates, per view, the sum of out^2 and the count of out > 0 (what sks_gt_tile_stats would read back
// from the planes: the masked-L2 loss of an all-zero render), so a frame's heat-maps are written and never re-read.
__global__ void k_heatmap_totals_finish(int V, double* __restrict__ totals);
template <bool TOTALS>
__global__ __launch_bounds__(256) void k_heatmaps(int W, int H, int J, const float* __restrict__ row, const float* __restrict__ col,
                                                   const float* __restrict__ cmin, const float* __restrict__ den,
                                                   float* __restrict__ out, double* __restrict__ totals)
{
    __shared__ double s_t[2][4];
    const int vj = blockIdx.z, y0 = blockIdx.y * 16, x = (blockIdx.x * 256 + threadIdx.x) * 4;
    float S = 0.0f, N = 0.0f;
    if (x < W) {
        const float lo = cmin[vj], d = den[vj];
        const float* r = row + (size_t)vj * H;
        const float* c = col + (size_t)vj * W + x;
        float* o = out + (size_t)vj * H * W + x;
        const int rows = min(16, H - y0);
        if ((W & 3) == 0) {
            const float4 k = *reinterpret_cast<const float4*>(c);
            for (int i = 0; i < rows; i++) {
                const float a = r[y0 + i];
                hm_v4f v = { (a * k.x - lo) / d, (a * k.y - lo) / d, (a * k.z - lo) / d, (a * k.w - lo) / d };
                __builtin_nontemporal_store(v, reinterpret_cast<hm_v4f*>(o + (size_t)(y0 + i) * W));
                if (TOTALS) {
                    S += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    N += ((v.x > 0.0f ? 1.0f : 0.0f) + (v.y > 0.0f ? 1.0f : 0.0f)) + ((v.z > 0.0f ? 1.0f : 0.0f) + (v.w > 0.0f ? 1.0f : 0.0f));
                }
            }
        } else {
            const int n = min(4, W - x);
            for (int i = 0; i < rows; i++) {
                const float a = r[y0 + i];
                for (int k = 0; k < n; k++) {
                    const float v = (a * c[k] - lo) / d;
                    o[(size_t)(y0 + i) * W + k] = v;
                    if (TOTALS) { S += v * v; N += v > 0.0f ? 1.0f : 0.0f; }
                }
            }
        }
    }
    if (TOTALS) {   // <= 64 pixels per thread: S, N exact enough in fp32; across threads in fp64
        const double tS = wave_sum_d((double)S), tN = wave_sum_d((double)N);
        if ((threadIdx.x & 63) == 0) { s_t[0][threadIdx.x >> 6] = tS; s_t[1][threadIdx.x >> 6] = tN; }
        __syncthreads();
        if (threadIdx.x == 0) {
            // like k_heatmap_totals: the blocks of a view finish in any order, so the sum of squares is combined in 2^-32 fixed
            // point with integer atomics (order-independent: the loss constants, and with them the reported loss and the early
            // stopping input, are reproducible bit for bit); k_heatmap_totals_finish converts it back
            const int v = vj / J;
            const double Sb = (s_t[0][0] + s_t[0][1]) + (s_t[0][2] + s_t[0][3]), Nb = (s_t[1][0] + s_t[1][1]) + (s_t[1][2] + s_t[1][3]);
            // (a view's {S, N} are one cache line: its atomics run one after the other at the memory side, ~15 ns each -- the
            // thousands of workgroups that saw nothing but zeros, most of a heat-map, send none)
            if (Sb != 0.0) atomicAdd(reinterpret_cast<unsigned long long*>(&totals[2 * v]), (unsigned long long)llrint(Sb * 4294967296.0));
            if (Nb != 0.0) atomicAdd(&totals[2 * v + 1], Nb);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// heat-map factors: the 1-D impulse responses `row`, `col` and the min-max constants of every (view, joint) plane,
// one workgroup per plane (utils/general_utils.py:175-304; the arithmetic is skelsplat_amd/heatmaps.py's
// ewa_lambdas_views + _impulse_response_1d + heatmap_factors, operation for operation: fp32 up to sqrt(lambda), fp64 for
// the responses).  Note the reference's operand order (R J)^T Sigma^T (R J) -- see heatmaps.py.
// ------------------------------------------------------------------------------------------------------------
struct HmTan {
    float x[SKS_MAX_VIEWS], y[SKS_MAX_VIEWS];
    int w[SKS_MAX_VIEWS], h[SKS_MAX_VIEWS];   // per-view image size (<= the W, H strides of row / col)
};

__device__ __forceinline__ double block_sum_d(double v, double* s_red)
{
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// scipy.ndimage.gaussian_filter1d of a unit impulse at integer p on an axis of n samples, mode='reflect', truncate=4:
// writes scale * response (fp32) and returns this thread's (min, max) of what it wrote
__device__ __forceinline__ void impulse_response(int n, int p, float sigma_f, float scale, float* __restrict__ out,
                                                 double* s_red, float& vmin, float& vmax)
{
    const double sig = (double)sigma_f, ss = sig * sig;
    const double radius = floor(4.0 * sig + 0.5);
    double part = 0.0;   // kernel normalisation: sum over |j| <= radius of exp(-0.5 j^2 / sigma^2)
    for (long long j = 1 + threadIdx.x; (double)j <= radius; j += 256) {
        const double jj = (double)j;
        part += exp(-0.5 * jj * jj / ss);
    }
    const double norm = 1.0 + 2.0 * block_sum_d(part, s_red);
    const double src[3] = { (double)p, -1.0 - (double)p, 2.0 * n - 1.0 - (double)p };   // the impulse and its mirrors
    vmin = FLT_MAX;
    vmax = -FLT_MAX;
    for (int i = threadIdx.x; i < n; i += 256) {
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const double d = (double)i - src[m];
            if (fabs(d) <= radius) acc += exp(-0.5 * d * d / ss);
        }
        const float v = scale * (float)(acc / norm);
        out[i] = v;
        vmin = fminf(vmin, v);
        vmax = fmaxf(vmax, v);
    }
}

__device__ __forceinline__ float block_minmax(float v, bool want_max, float* s_red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = want_max ? fmaxf(v, o) : fminf(v, o);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return want_max ? fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))
                    : fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(256) void k_heatmap_factors(int J, int W, int H, const float* __restrict__ means,
                                                          const float* __restrict__ scales, const float* __restrict__ rots,
                                                          float scale_modifier, const float* __restrict__ poses_2d,
                                                          const float* __restrict__ viewmatrix, HmTan tan,
                                                          float* __restrict__ row, float* __restrict__ col,
                                                          float* __restrict__ cmin, float* __restrict__ den, int vf)
{
    __shared__ double s_red[4];
    __shared__ float s_redf[4];
    const int j = blockIdx.x, v = blockIdx.y, vj = v * J + j;
    const int Ws = W, Hs = H;        // strides of col / row: the largest view
    W = tan.w[v]; H = tan.h[v];      // this view's own size (focal lengths, clamps, reflections)
    if (vf > 0) {   // frames batched: view v belongs to frame v / vf, whose J Gaussians sit frame-th in the stacked tensors
        const size_t fr = (size_t)(v / vf);
        means += fr * J * 3; scales += fr * J * 3; rots += fr * J * 4;
    }
    // ---- lambda1, lambda2 (every thread the same scalars) ----
    const float q0 = rots[4 * j], q1 = rots[4 * j + 1], q2 = rots[4 * j + 2], q3 = rots[4 * j + 3];
    const float qn = sqrtf(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
    const float r = q0 / qn, x = q1 / qn, y = q2 / qn, z = q3 / qn;
    const float R[3][3] = { { 1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y) },
                            { 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x) },
                            { 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y) } };
    float L[3][3], Sg[3][3];
    for (int a = 0; a < 3; a++)
        for (int k = 0; k < 3; k++) L[a][k] = R[a][k] * (scale_modifier * scales[3 * j + k]);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Sg[a][b] = (L[a][0] * L[b][0] + L[a][1] * L[b][1]) + L[a][2] * L[b][2];
    const float* vm = viewmatrix + 16 * v;   // world_view_transform as stored (the matrix transposed): M[i][k] = vm[4k + i]
    const float px = means[3 * j], py = means[3 * j + 1], pz = means[3 * j + 2];
    float t[3];
    for (int i = 0; i < 3; i++) t[i] = ((vm[i] * px + vm[4 + i] * py) + vm[8 + i] * pz) + vm[12 + i] * 1.0f;
    const float tanx = tan.x[v], tany = tan.y[v];
    const float fx = (float)W / (2.0f * tanx), fy = (float)H / (2.0f * tany);
    const float tz = t[2];
    const float tx = fminf(fmaxf(t[0] / tz, -1.3f * tanx), 1.3f * tanx) * tz;
    const float ty = fminf(fmaxf(t[1] / tz, -1.3f * tany), 1.3f * tany) * tz;
    const float Jm[3][3] = { { fx / tz, 0.0f, -(fx * tx) / (tz * tz) }, { 0.0f, fy / tz, -(fy * ty) / (tz * tz) }, { 0.0f, 0.0f, 0.0f } };
    float T[3][3], A[3][3], cov[2][2];
    for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) T[a][c] = (vm[a] * Jm[0][c] + vm[4 + a] * Jm[1][c]) + vm[8 + a] * Jm[2][c];
    for (int a = 0; a < 3; a++)      // A = T^T Sigma^T
        for (int b = 0; b < 3; b++) A[a][b] = (T[0][a] * Sg[b][0] + T[1][a] * Sg[b][1]) + T[2][a] * Sg[b][2];
    for (int a = 0; a < 2; a++)      // cov = A T
        for (int c = 0; c < 2; c++) cov[a][c] = (A[a][0] * T[0][c] + A[a][1] * T[1][c]) + A[a][2] * T[2][c];
    const float cx = cov[0][0] + 0.3f, cy = cov[0][1], cz = cov[1][1] + 0.3f;
    const float det = cx * cz - cy * cy;
    const float mid = 0.5f * (cx + cz);
    const float root = sqrtf(fmaxf(mid * mid - det, 0.1f));
    const float l1 = mid + root, l2 = mid - root;
    // ---- responses: sigma1 filters rows (axis 0), sigma2 columns; the impulse sits at the truncated 2D detection ----
    const int xs = min(max((int)poses_2d[2 * vj], 0), W - 1), ys = min(max((int)poses_2d[2 * vj + 1], 0), H - 1);
    float rmin, rmax, kmin, kmax;
    impulse_response(H, ys, sqrtf(l1), 255.0f, row + (size_t)vj * Hs, s_red, rmin, rmax);
    impulse_response(W, xs, sqrtf(l2), 1.0f, col + (size_t)vj * Ws, s_red, kmin, kmax);
    rmin = block_minmax(rmin, false, s_redf);
    rmax = block_minmax(rmax, true, s_redf);
    kmin = block_minmax(kmin, false, s_redf);
    kmax = block_minmax(kmax, true, s_redf);
    if (threadIdx.x == 0) {
        const float lo = rmin * kmin, hi = rmax * kmax;
        cmin[vj] = lo;
        den[vj] = (hi - lo) + 1e-8f;
    }
}


// ------------------------------------------------------------------------------------------------------------
// Per-view loss constants of heat-maps that are never written: sum of gt^2 and count of gt > 0 over the J planes of a
// view, from the separable factors alone (gt = (row[y] * col[x] - cmin) / den, the very expression k_heatmaps stores).
// Outside the impulse response's 4-sigma support the factors are exactly zero, and then so is cmin and the pixel: only
// the rows with a non-zero factor are walked (a few dozen of a thousand).  grid (row bands of 16, J, V); the counts are
// integers (exact in any order), the sums are accumulated in fp64 like k_heatmaps' own and combined across blocks in
// fixed point (order-independent: reproducible bit for bit).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_heatmap_totals(int J, int Ws, int Hs, HmTan sz, const float* __restrict__ row,
                                                         const float* __restrict__ col, const float* __restrict__ cmin,
                                                         const float* __restrict__ den, double* __restrict__ totals)
{
    __shared__ double s_t[2][4];
    const int j = blockIdx.y, v = blockIdx.z, vj = v * J + j, y0 = blockIdx.x * 16;
    const int W = sz.w[v], H = sz.h[v];
    if (y0 >= H) return;
    const float lo = cmin[vj], d = den[vj];
    const float* r = row + (size_t)vj * Hs;
    const float* c = col + (size_t)vj * Ws;
    const int rows = min(16, H - y0);
    bool any = lo != 0.0f;
    for (int i = 0; i < rows && !any; i++) any = r[y0 + i] != 0.0f;   // (uniform)
    if (!any) return;
    double S = 0.0, N = 0.0;
    for (int x = threadIdx.x; x < W; x += 256) {
        const float k = c[x];
        if (k == 0.0f && lo == 0.0f) continue;
        float s = 0.0f, n = 0.0f;
        for (int i = 0; i < rows; i++) {
            const float g = (r[y0 + i] * k - lo) / d;
            s += g * g;
            n += g > 0.0f ? 1.0f : 0.0f;
        }
        S += (double)s;
        N += (double)n;
    }
    S = wave_sum_d(S);
    N = wave_sum_d(N);
    if ((threadIdx.x & 63) == 0) { s_t[0][threadIdx.x >> 6] = S; s_t[1][threadIdx.x >> 6] = N; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // The blocks of a view finish in any order.  Counts are integers (a double sum of them is exact in any order); the
        // sum of squares is combined in 2^-32 FIXED POINT with integer atomics -- integer addition is associative, so the
        // total is bit-reproducible run to run (it feeds the reported loss and early stopping) -- and converted back by
        // k_heatmap_totals_finish.  A view's sum is < J * H * W < 2^27, a block's rounding 2^-33.
        const double Sb = (s_t[0][0] + s_t[0][1]) + (s_t[0][2] + s_t[0][3]), Nb = (s_t[1][0] + s_t[1][1]) + (s_t[1][2] + s_t[1][3]);
        if (Sb != 0.0) atomicAdd(reinterpret_cast<unsigned long long*>(&totals[2 * v]), (unsigned long long)llrint(Sb * 4294967296.0));
        if (Nb != 0.0) atomicAdd(&totals[2 * v + 1], Nb);   // (all-zero workgroups send nothing: see k_heatmaps)
    }
}

__global__ void k_heatmap_totals_finish(int V, double* __restrict__ totals)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const unsigned long long fx = *reinterpret_cast<const unsigned long long*>(&totals[2 * v]);
    totals[2 * v] = (double)fx * (1.0 / 4294967296.0);
}

// loss and gradient scale of the criterion from a view's {S, N} (utils/loss_utils.py:96-99): mean -> S / N and 1 / N (an empty
// mask gives inf * 0 = NaN like torch's mean of nothing), sum -> S and 1
__global__ void k_masked_l2_finish(int V, const double* __restrict__ sums, float* __restrict__ loss, float* __restrict__ scale,
                                   int mean)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const double S = sums[2 * v], N = sums[2 * v + 1];
    const double sc = mean ? 1.0 / N : 1.0;
    loss[v] = (float)(S * sc);
    scale[v] = (float)sc;
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
    size_t blocks = (n_per_view / 4 + L2_THREADS - 1) / L2_THREADS;
    if (blocks > L2_BLOCKS) blocks = L2_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_masked_l2, dim3((unsigned)blocks, V), dim3(L2_THREADS), 0, st, n_per_view, render, gt, dL_unscaled, sums);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_masked_l2_loss(int V, size_t n_per_view, const float* render, const float* gt, float* dL_unscaled, double* sums,
                       float* loss, float* scale, int mean, void* stream)
{
    if (!loss || !scale) return fail2(-2, "masked_l2_loss: missing pointer");
    if (int rc = sks_masked_l2(V, n_per_view, render, gt, dL_unscaled, sums, stream)) return rc;
    hipLaunchKernelGGL(k_masked_l2_finish, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, V, sums, loss, scale, mean);
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

size_t sks_knn3_scratch_bytes(int P)
{
    return P > 0 ? knn_scratch(P, nullptr, nullptr) : 0;
}

int sks_knn3_meandist2_grid(int P, const float* points, float* mean_dist2, void* scratch, size_t scratch_bytes, void* stream)
{
    if (P < 0) return fail2(-1, "knn: P negative");
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !scratch) return fail2(-2, "knn: missing pointer");
    KnnGrid g;
    if (knn_scratch(P, &g, scratch) > scratch_bytes) return fail2(-1, "knn: scratch too small (see sks_knn3_scratch_bytes)");
    hipStream_t st = (hipStream_t)stream;
    const int ncell = g.n * g.n * g.n;
    const int nb = (P + 255) / 256, nparts = nb < 256 ? nb : 256;
    HIP_TRY2(hipMemsetAsync(g.cursor, 0, (size_t)ncell * 4, st));
    hipLaunchKernelGGL(k_knn_bbox, dim3(nparts), dim3(256), 0, st, P, points, g.part);
    hipLaunchKernelGGL(k_knn_count, dim3(nb), dim3(256), 0, st, P, points, g, nparts);
    hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, st, ncell, g);
    hipLaunchKernelGGL(k_knn_scatter, dim3(nb), dim3(256), 0, st, P, points, g, nparts);
    hipLaunchKernelGGL(k_knn_search, dim3(nb), dim3(256), 0, st, P, g, nparts, mean_dist2);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_heatmaps(int V, int J, int W, int H, const float* row, const float* col, const float* cmin, const float* den,
                 float* out, double* gt_totals, void* stream)
{
    if (V < 1 || J < 1 || W < 1 || H < 1 || (long long)V * J > 65535) return fail2(-1, "heatmaps: bad shape");
    if (!row || !col || !cmin || !den || !out) return fail2(-2, "heatmaps: missing pointer");
    dim3 grid((W + 1023) / 1024, (H + 15) / 16, V * J);
    if (gt_totals) {
        HIP_TRY2(hipMemsetAsync(gt_totals, 0, (size_t)V * 2 * sizeof(double), (hipStream_t)stream));
        hipLaunchKernelGGL(k_heatmaps<true>, grid, dim3(256), 0, (hipStream_t)stream, W, H, J, row, col, cmin, den, out, gt_totals);
        hipLaunchKernelGGL(k_heatmap_totals_finish, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, V, gt_totals);
    } else {
        hipLaunchKernelGGL(k_heatmaps<false>, grid, dim3(256), 0, (hipStream_t)stream, W, H, J, row, col, cmin, den, out, nullptr);
    }
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_heatmap_factors(int V, int J, int W, int H, const float* means3D, const float* scales, const float* rotations,
                        float scale_modifier, const float* poses_2d, const float* viewmatrix, const float* tanfovx,
                        const float* tanfovy, float* row, float* col, float* cmin, float* den, int frames,
                        const int* view_wh, void* stream)
{
    if (V < 1 || V > SKS_MAX_VIEWS || J < 1 || W < 1 || H < 1) return fail2(-1, "heatmap factors: bad shape");
    if (frames < 1 || V % frames) return fail2(-1, "heatmap factors: frames must divide the number of views");
    if (!means3D || !scales || !rotations || !poses_2d || !viewmatrix || !tanfovx || !tanfovy || !row || !col || !cmin || !den)
        return fail2(-2, "heatmap factors: missing pointer");
    HmTan tan;
    for (int v = 0; v < V; v++) {
        tan.x[v] = tanfovx[v]; tan.y[v] = tanfovy[v];
        tan.w[v] = view_wh ? view_wh[2 * v] : W; tan.h[v] = view_wh ? view_wh[2 * v + 1] : H;
        if (tan.w[v] < 1 || tan.w[v] > W || tan.h[v] < 1 || tan.h[v] > H)
            return fail2(-1, "heatmap factors: every view's size must be within [1, W] x [1, H]");
    }
    hipLaunchKernelGGL(k_heatmap_factors, dim3(J, V), dim3(256), 0, (hipStream_t)stream, J, W, H, means3D, scales, rotations,
                       scale_modifier, poses_2d, viewmatrix, tan, row, col, cmin, den, frames > 1 ? V / frames : 0);
    HIP_TRY2(hipGetLastError());
    return 0;
}

int sks_heatmap_totals(int V, int J, int W, int H, const float* row, const float* col, const float* cmin, const float* den,
                       const int* view_wh, double* gt_totals, void* stream)
{
    if (V < 1 || V > SKS_MAX_VIEWS || J < 1 || W < 1 || H < 1) return fail2(-1, "heatmap totals: bad shape");
    if (!row || !col || !cmin || !den || !gt_totals) return fail2(-2, "heatmap totals: missing pointer");
    HmTan sz;
    for (int v = 0; v < V; v++) {
        sz.x[v] = sz.y[v] = 0.0f;
        sz.w[v] = view_wh ? view_wh[2 * v] : W; sz.h[v] = view_wh ? view_wh[2 * v + 1] : H;
        if (sz.w[v] < 1 || sz.w[v] > W || sz.h[v] < 1 || sz.h[v] > H)
            return fail2(-1, "heatmap totals: every view's size must be within [1, W] x [1, H]");
    }
    HIP_TRY2(hipMemsetAsync(gt_totals, 0, (size_t)V * 2 * sizeof(double), (hipStream_t)stream));
    hipLaunchKernelGGL(k_heatmap_totals, dim3((H + 15) / 16, J, V), dim3(256), 0, (hipStream_t)stream, J, W, H, sz, row, col, cmin,
                       den, gt_totals);
    hipLaunchKernelGGL(k_heatmap_totals_finish, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, V, gt_totals);
    HIP_TRY2(hipGetLastError());
    return 0;
}

}  // extern "C"
