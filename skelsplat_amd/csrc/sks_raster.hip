// sks_raster.hip -- skeletal-Gaussian rasterizer for MI355X (gfx950): geometry, binning, compositing, C ABI.
//
// Replaces DGR/cuda_rasterizer/{forward.cu,backward.cu,rasterizer_impl.cu} + DGR/rasterize_points.cu of the
// reference ("DGR/" = submodules/diff-gaussian-rasterization-h36m/).  Design:
//   * one launch sequence renders V views that share the Gaussian parameters (blockIdx.z = view);
//   * P <= SKS_SMALL_P ("skeleton" regime, the reference's real configs have P = 15..19): NO global binning.
//     Every workgroup owns a contiguous chunk of one 16-row tile band of the image, finds the Gaussians whose
//     tile rect crosses the band, depth-sorts them in LDS (key = depth bits, index -- the order the reference
//     gets from its stable radix sort of (tile | depth) keys with index-major emission) and composites /
//     zero-fills its chunk with 16-byte-per-lane plane-contiguous stores;
//   * larger P: tile-centric binning (count -> scan -> scatter -> per-tile LDS bitonic sort) and one workgroup per
//     tile with LDS-staged batches;
//   * backward never needs final_T / n_contrib from HBM: it re-composites its pixels first, then walks back to
//     front exactly like backward.cu:531-637; per-Gaussian partial sums are reduced across the wavefront, then in
//     LDS, and leave the workgroup as one atomic per value;
//   * HBM-bound: no MFMA anywhere (there is no dense contraction in this path).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/skelsplat_hip.h"
#include "sks_math.h"
#include "sks_loop_dev.h"

using namespace sks;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail((int)e_, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

#define STAGE_CHECK(name)                                                                      \
    do {                                                                                       \
        HIP_TRY(hipGetLastError());                                                            \
        if (flags & SKS_DEBUG_SYNC) {                                                          \
            hipError_t e_ = hipStreamSynchronize(st);                                          \
            if (e_ != hipSuccess) return fail((int)e_, "stage %s: %s", name, hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)

// ---- measurement hook (sks_prof_enable / sks_prof_read) ----
constexpr int PROF_MAX = 16384;
struct ProfKind {
    hipEvent_t b[PROF_MAX], e[PROF_MAX];
    int created = 0, n = 0;
};
bool g_prof_on = false;
ProfKind g_prof[2];

struct ProfScope {  // records begin at construction, end at destruction, around one kernel launch
    ProfKind* k = nullptr;
    hipStream_t st;
    ProfScope(int kind, hipStream_t s) : st(s)
    {
        if (!g_prof_on) return;
        ProfKind& p = g_prof[kind];
        if (p.n >= PROF_MAX) return;
        if (p.n >= p.created) {
            if (hipEventCreate(&p.b[p.created]) != hipSuccess || hipEventCreate(&p.e[p.created]) != hipSuccess) return;
            p.created++;
        }
        k = &p;
        (void)hipEventRecord(p.b[p.n], st);
    }
    ~ProfScope()
    {
        if (!k) return;
        (void)hipEventRecord(k->e[k->n], st);
        k->n++;
    }
};

constexpr int LCAP = 64;          // binned path: entries staged in LDS per batch (short lists are the norm; the
                                  // smaller the footprint, the more tile blocks a CU keeps in flight)
constexpr int NACC = 8;           // per-Gaussian accumulators before the feature block:
                                  // 0,1 dL_dmean2D.xy  2,3,4 dL_dconic.{x,y,w}  5 dL_dopacity  6 dL_dinvdepth  7 pad
constexpr int BWD_SPLITS = 16;    // small path backward: partial-sum slots (workgroups) per (view, Gaussian)

struct ViewTan {
    float x[SKS_MAX_VIEWS];
    float y[SKS_MAX_VIEWS];
};

struct Geom {  // per-(view, Gaussian) records kept for backward ("geomBuffer")
    float4* co;    // conic.x, conic.y, conic.z, opacity * h_convolution_scaling   (forward.cu:269)
    float4* xyd;   // pixel centre x, y, view depth, 1/depth
    uint4* rect;   // tile rect xmin, ymin, xmax, ymax (all 0 when culled)
    uint32_t* cover;  // per (view, tile band): word 0 = "some rect crosses this band", then one bit per tile column;
                      // nullptr when not produced (P > SKS_SMALL_P or the image has too many tiles)
};

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// SKS_RAW_PARAMS: the caller passes the LEAF parameters (_opacity logits, _scaling log-scales, _rotation raw
// quaternions) and the activations of scene/gaussian_model.py:39-47 (sigmoid / exp / F.normalize) run in-kernel.
struct Activated {
    float opacity, s[3], q[4], qnorm;
};
__device__ __forceinline__ Activated activate(bool raw, float op, const float s[3], const float q[4])
{
    Activated a;
    if (raw) {
        a.opacity = 1.0f / (1.0f + expf_fixed(-op));
#pragma unroll
        for (int k = 0; k < 3; k++) a.s[k] = expf_fixed(s[k]);
        const float nn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        a.qnorm = fmaxf(sqrtf(nn), 1e-12f);
#pragma unroll
        for (int k = 0; k < 4; k++) a.q[k] = q[k] / a.qnorm;
    } else {
        a.opacity = op;
        a.qnorm = 1.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) a.s[k] = s[k];
#pragma unroll
        for (int k = 0; k < 4; k++) a.q[k] = q[k];
    }
    return a;
}

constexpr int COVER_MAX_WORDS = 4096;  // LDS bitmap of k_geom_fwd (16 KB)
__host__ __device__ inline int cover_cw(int W) { return 1 + (((W + TILE - 1) / TILE) + 31) / 32; }
inline bool cover_enabled(int P, int W, int H)
{
    return P <= SKS_SMALL_P && ((H + TILE - 1) / TILE) * cover_cw(W) <= COVER_MAX_WORDS;
}
inline Geom geom_from(void* base, int V, int P, int W, int H)
{
    char* p = (char*)base;
    Geom g;
    size_t n = (size_t)V * P;
    g.co = (float4*)p; p += align256(n * sizeof(float4));
    g.xyd = (float4*)p; p += align256(n * sizeof(float4));
    g.rect = (uint4*)p; p += align256(n * sizeof(uint4));
    g.cover = cover_enabled(P, W, H) ? (uint32_t*)p : nullptr;
    return g;
}
inline uint32_t* geom_cover_ptr(void* base, int V, int P)   // the cover region exists for every P (geom_bytes)
{
    return (uint32_t*)((char*)base + 3 * align256((size_t)V * P * sizeof(float4)));
}
inline size_t geom_bytes(int V, int P, int W, int H)
{
    return 3 * align256((size_t)V * P * 16) + align256((size_t)V * ((H + TILE - 1) / TILE) * cover_cw(W) * 4);
}

struct Bin {  // binned path scratch ("binningBuffer" + ImageState::ranges)
    uint32_t* count;   // V*NT
    uint32_t* cursor;  // V*NT
    uint2* ranges;     // V*NT
    int* nrend;        // V (+ overflow flag at [V])
    unsigned long long* keys;  // V*cap : (depth bits << 32) | Gaussian index, sorted per tile
};
inline Bin bin_from(void* base, int V, int NT, size_t cap)
{
    char* p = (char*)base;
    Bin b;
    b.count = (uint32_t*)p; p += align256((size_t)V * NT * 4);
    b.cursor = (uint32_t*)p; p += align256((size_t)V * NT * 4);
    b.ranges = (uint2*)p; p += align256((size_t)V * NT * 8);
    b.nrend = (int*)p; p += align256((size_t)(V + 1) * 4);
    b.keys = (unsigned long long*)p;
    return b;
}
inline size_t bin_bytes(int V, int NT, size_t cap)
{
    return 2 * align256((size_t)V * NT * 4) + align256((size_t)V * NT * 8) + align256((size_t)(V + 1) * 4) +
           align256((size_t)V * cap * 8);
}

// ------------------------------------------------------------------------------------------------------------
// geometry forward: preprocessCUDA, DGR/cuda_rasterizer/forward.cu:153-273 (+ in_frustum auxiliary.h:151-176)
// ------------------------------------------------------------------------------------------------------------
// one (view, Gaussian) pair; returns the tile rect (all zero when culled or !live)
__device__ __forceinline__ uint4 geom_fwd_one(int P, int W, int H, const ViewTan& vt, const float* __restrict__ vms,
                                              const float* __restrict__ pms, const float* __restrict__ means,
                                              const float* __restrict__ opac, const float* __restrict__ scales,
                                              const float* __restrict__ rots, const float* __restrict__ cov3Dp,
                                              float smod, unsigned flags, const Geom& g, int* __restrict__ radii,
                                              int v, int idx, bool live)
{
    const size_t o = (size_t)v * P + (live ? idx : 0);
    const float* V = vms + 16 * v;
    const float* PM = pms + 16 * v;
    const float tan_fovx = vt.x[v], tan_fovy = vt.y[v];
    const float focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:224-225
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;

    int radius_out = 0;
    uint4 rect = make_uint4(0, 0, 0, 0);
    float4 co = make_float4(0, 0, 0, 0), xyd = make_float4(0, 0, 1.0f, 1.0f);

    const int li = live ? idx : 0;
    const float p_orig[3] = { means[3 * li], means[3 * li + 1], means[3 * li + 2] };
    float p_view[3];
    transformPoint4x3(p_orig, V, p_view);
    if (live && p_view[2] > 0.2f) {
        float p_hom[4];
        transformPoint4x4(p_orig, PM, p_hom);
        const float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        const float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };
        float cov3D[6];
        if (cov3Dp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3Dp[6 * li + i];
        } else {
            const float s[3] = { scales[3 * li], scales[3 * li + 1], scales[3 * li + 2] };
            const float q[4] = { rots[4 * li], rots[4 * li + 1], rots[4 * li + 2], rots[4 * li + 3] };
            const Activated ac = activate(flags & SKS_RAW_PARAMS, 0.0f, s, q);
            computeCov3D(ac.s, smod, ac.q, cov3D);
        }
        Cov2D c;
        cov2d(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, V, c);
        float cov_x = c.cov.m[0][0], cov_y = c.cov.m[0][1], cov_z = c.cov.m[1][1];
        constexpr float h_var = 0.3f;
        const float det_cov = cov_x * cov_z - cov_y * cov_y;
        cov_x += h_var;
        cov_z += h_var;
        const float det_cov_plus_h_cov = cov_x * cov_z - cov_y * cov_y;
        float h_convolution_scaling = 1.0f;
        if (flags & SKS_ANTIALIASING) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
        const float det = det_cov_plus_h_cov;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float conic[3] = { cov_z * det_inv, -cov_y * det_inv, cov_x * det_inv };
            const float mid = 0.5f * (cov_x + cov_z);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
            int xmin, ymin, xmax, ymax;
            getRect(px, py, (int)my_radius, gx, gy, xmin, ymin, xmax, ymax);
            if ((xmax - xmin) * (ymax - ymin) != 0) {
                radius_out = (int)my_radius;
                rect = make_uint4(xmin, ymin, xmax, ymax);
                const float op_raw = opac[li];
                const float op = (flags & SKS_RAW_PARAMS) ? 1.0f / (1.0f + expf_fixed(-op_raw)) : op_raw;
                co = make_float4(conic[0], conic[1], conic[2], op * h_convolution_scaling);
                xyd = make_float4(px, py, p_view[2], 1 / p_view[2]);
            }
        }
    }
    if (live) {
        radii[o] = radius_out;
        g.co[o] = co;
        g.xyd[o] = xyd;
        g.rect[o] = rect;
    }
    return rect;
}

__global__ __launch_bounds__(256) void k_geom_fwd(int P, int W, int H, ViewTan vt, const float* __restrict__ vms,
                                                   const float* __restrict__ pms, const float* __restrict__ means,
                                                   const float* __restrict__ opac, const float* __restrict__ scales,
                                                   const float* __restrict__ rots, const float* __restrict__ cov3Dp,
                                                   float smod, unsigned flags, Geom g, int* __restrict__ radii)
{
    __shared__ uint32_t s_cov[COVER_MAX_WORDS];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y;
    const bool live = idx < P;
    const uint4 rect = geom_fwd_one(P, W, H, vt, vms, pms, means, opac, scales, rots, cov3Dp, smod, flags, g, radii, v, idx, live);
    if (g.cover) {  // single block per view (P <= 256): bitmap of covered tiles for the forward's fill blocks
        const int gy = (H + TILE - 1) / TILE;
        const int cw = cover_cw(W), nw = gy * cw;
        for (int i = threadIdx.x; i < nw; i += 256) s_cov[i] = 0u;
        __syncthreads();
        if (live) {
            for (unsigned y = rect.y; y < rect.w; y++) {
                atomicOr(&s_cov[y * cw], 1u);
                for (unsigned x = rect.x; x < rect.z; x++) atomicOr(&s_cov[y * cw + 1 + (x >> 5)], 1u << (x & 31));
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nw; i += 256) g.cover[(size_t)v * nw + i] = s_cov[i];
    }
}

__global__ void k_mark_visible(int P, const float* __restrict__ means, const float* __restrict__ V,
                               uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float p[3] = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
    float pv[3];
    transformPoint4x3(p, V, pv);
    present[idx] = pv[2] > 0.2f;
}

// ------------------------------------------------------------------------------------------------------------
// LDS-resident, depth-ordered list of Gaussians for a workgroup's pixels
// ------------------------------------------------------------------------------------------------------------
template <int CG>
struct List {
    float2 xy[LCAP];
    float4 co[LCAP];
    float invd[LCAP];
    int xr[LCAP];    // xmin | xmax << 16 (tile units); the pixel's tile column must fall inside
    int id[LCAP];
    float feat[LCAP * CG];
};

// Forward compositing of one pixel over n LDS entries (forward.cu:346-386), carrying state across batches.
template <int CG, bool XF>
__device__ __forceinline__ void composite_px(const List<CG>& L, int n, float pxf, float pyf, int tx, float& T,
                                             float (&acc)[CG], float& inv, uint32_t& contributor, uint32_t& last,
                                             bool& done)
{
    for (int k = 0; k < n && !done; k++) {
        if (XF) {
            const int xr = L.xr[k];
            if (tx < (xr & 0xffff) || tx >= (xr >> 16)) continue;  // not in this pixel's tile list
        }
        contributor++;
        const float2 xy = L.xy[k];
        const float4 co = L.co[k];
        const float dx = xy.x - pxf, dy = xy.y - pyf;
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co.w * expf_fixed(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) {
            done = true;
            continue;
        }
#pragma unroll
        for (int ch = 0; ch < CG; ch++) acc[ch] += L.feat[k * CG + ch] * alpha * T;
        inv += L.invd[k] * alpha * T;
        T = test_T;
        last = contributor;
    }
}

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

struct FwdArgs {
    int P, C, W, H;
    unsigned flags;
    Geom g;
    const float* features;
    float* out_color;
    float* out_invdepth;
    float* final_T;
    uint32_t* n_contrib;
};

// ------------------------------------------------------------------------------------------------------------
// small path forward, "fill + sparse composite" in ONE launch.  With a skeleton (P = 15..19) a view covers ~100 of
// ~4000 tiles, so the kernel is a dense zero fill of (C+1) planes with a sprinkle of compositing.  Two roles by
// block index (composite blocks come first so their latency hides under the fill):
//   * composite role, block (slot, g, v): walks the tiles of Gaussian g's rect (slot, slot+T_SLOTS, ...).  The
//     tile is rendered by the LOWEST-index Gaussian whose rect covers it (so every covered tile has exactly one
//     owner); the owner gathers the covering Gaussians, orders them by (depth bits, index) -- the order the
//     reference gets from its stable radix sort of (tile | depth) keys with index-major emission -- and composites
//     thread-per-pixel exactly like forward.cu:278-401, writing all C+1 planes of the tile;
//   * fill role, block (plane, band, v): streams zeros over its plane's 16-row band (rows*W contiguous floats,
//     16 B per lane), skipping the tile columns some Gaussian rect covers in that band.
// A pure fill of this shape runs at the memset rate (7+ TB/s for the 288 MB of 4 H36M views on MI355X).
// PPT = 4 needs W % 4 == 0; PPT = 1 handles any W with 4-byte stores.
// ------------------------------------------------------------------------------------------------------------
constexpr int T_SLOTS = 16;     // composite blocks per (view, Gaussian)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d)
{
    v4f v = { a, b, c, d };
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
    else *reinterpret_cast<v4f*>(p) = v;
}

// dynamic-LDS list with capacity `cap` entries (cap = P rounded up to 16)
template <int CG>
struct DynList {
    float2* xy;
    float4* co;
    float* invd;
    float* feat;               // cap * CG
    unsigned long long* key;   // cap
    uint4* rect;               // cap
    __device__ DynList(char* base, int cap)
    {
        co = (float4*)base; base += (size_t)cap * 16;
        rect = (uint4*)base; base += (size_t)cap * 16;
        key = (unsigned long long*)base; base += (size_t)cap * 8;
        xy = (float2*)base; base += (size_t)cap * 8;
        invd = (float*)base; base += (size_t)cap * 4;
        feat = (float*)base;
    }
    static size_t bytes(int cap, int cg) { return (size_t)cap * (16 + 16 + 8 + 8 + 4 + 4 * (size_t)cg); }
};

// ------------------------------------------------------------------------------------------------------------
// fill role shared by the two forward kernels (fill + sparse composite): block (q, band, zid = view * (C+1) + plane)
// streams zeros over pb passes of its plane's 16-row band, skipping the tile columns whose bit is set in the
// band's `cover` row (those tiles are written by a composite block).
// One block = ONE pass: 256 threads x 16 B = 4 KB of one plane.  On MI355X a dispatch of many 4 KB blocks is the
// fastest fill shape measured (tools/fill_bench.hip: 44-47 us for 288 MB vs 48-55 us with 16 KB per block), but only
// if nothing VECTOR-memory sits in front of the store: a dependent global_load queues behind the chip-wide flood of
// stores (measured: 2.4 TB/s).  The band's cover words written by k_geom_fwd are therefore read through the SCALAR
// cache (s_load: `cover` is a const __restrict__ kernel argument and the address is made wave-uniform).
// block order = memory order (pass, band, plane, view): concurrently running blocks write one contiguous window
// ------------------------------------------------------------------------------------------------------------
template <int PPT, bool NT>
__device__ __forceinline__ void fwd_fill_role(const FwdArgs& a, int q, int band, int zid, int gy, int pb,
                                              const uint32_t* __restrict__ cover)
{
    const int tid = threadIdx.x;
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const size_t HW = (size_t)H * W;
    const int v = zid / (C + 1);
    const int plane = zid - v * (C + 1);
    const bool is_inv = plane == C;
    const int cw = cover_cw(W);
    constexpr int PASS = 256 * PPT;
    const int rows = min(TILE, H - band * TILE);
    const int Nb = rows * W;  // this band of this plane: Nb contiguous floats
    const int base0 = q * PASS * pb + tid * PPT;     // this block covers passes q*pb .. q*pb + pb - 1 of the band
    const size_t band0 = (size_t)band * TILE * W;
    float* out = (is_inv ? a.out_invdepth + (size_t)v * HW : a.out_color + ((size_t)v * C + plane) * HW) + band0;
    float* outT = (is_inv && a.final_T) ? a.final_T + (size_t)v * HW + band0 : nullptr;
    uint32_t* outN = (is_inv && a.n_contrib) ? a.n_contrib + (size_t)v * HW + band0 : nullptr;
    bool any = false;
    const uint32_t* __restrict__ cwp = cover;
    if (cover) {
        const int wo = __builtin_amdgcn_readfirstlane((v * gy + band) * cw);
        cwp = cover + wo;
        any = cwp[0] != 0u;   // wave-uniform scalar load; only bands some rect crosses look at the tile bits
    }
    uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;
    if (any && cw <= 5) {  // W <= 2048: the band's <= 4 bit words via the scalar cache, selected per lane below
        w0 = cwp[1]; w1 = cw > 2 ? cwp[2] : 0u; w2 = cw > 3 ? cwp[3] : 0u; w3 = cw > 4 ? cwp[4] : 0u;
    }
    if (pb < 0) {
        // Row-aligned mode (PPT == 4, cover present, launcher: fill_row_mode): block q = (row block, x chunk of 1024
        // pixels); a lane keeps the same 4 pixels' column in every row, so its tile column -- and the skip bit -- is
        // computed ONCE, without the per-pass modulo of the linear mode below.
        const int pbr = -pb;
        const int chunks = (W + PASS - 1) / PASS;
        const int rb = q / chunks, chunk = q - rb * chunks;
        const int x = chunk * PASS + tid * PPT;
        if (x >= W) return;
        if (any) {
            const int tx = x >> 4;
            const int wi = tx >> 5;
            const uint32_t word = cw <= 5 ? (wi == 0 ? w0 : wi == 1 ? w1 : wi == 2 ? w2 : w3) : cwp[1 + wi];
            if ((word >> (tx & 31)) & 1u) return;
        }
        const int r1 = min(rows, (rb + 1) * pbr);
        for (int r = rb * pbr; r < r1; r++) {
            const int base = r * W + x;
            store4<NT>(out + base, 0.0f, 0.0f, 0.0f, 0.0f);
            if (outT || outN) {
#pragma unroll
                for (int p = 0; p < PPT; p++) {
                    if (outT) outT[base + p] = 1.0f;
                    if (outN) outN[base + p] = 0u;
                }
            }
        }
        return;
    }
    for (int k = 0; k < pb; k++) {
        const int base = base0 + k * PASS;
        if (base >= Nb) break;
        bool skip0 = false;  // does a composite block write this thread's tile?
        if (any) {
            const int tx = (base % W) >> 4;  // PPT == 4: W % 4 == 0, the 4 pixels share one tile column
            const int wi = tx >> 5;
            const uint32_t word = cw <= 5 ? (wi == 0 ? w0 : wi == 1 ? w1 : wi == 2 ? w2 : w3) : cwp[1 + wi];
            skip0 = (word >> (tx & 31)) & 1u;
        } else if (!cover) {  // cover not precomputed (very large images): test the rects directly
            const uint4* grect = a.g.rect + (size_t)v * P;
            const int tx = (base % W) >> 4;
            for (int i = 0; i < P; i++) {
                const uint4 r = grect[i];
                skip0 = skip0 || ((int)r.y <= band && band < (int)r.w && (int)r.x <= tx && tx < (int)r.z);
            }
        }
        if (skip0) continue;
        if (PPT == 4) store4<NT>(out + base, 0.0f, 0.0f, 0.0f, 0.0f);
        else out[base] = 0.0f;
        if (outT || outN) {  // debug planes of the reference's ImageState: T = 1, no contributor outside covered tiles
#pragma unroll
            for (int p = 0; p < PPT; p++) {
                if (outT) outT[base + p] = 1.0f;
                if (outN) outN[base + p] = 0u;
            }
        }
    }
}

template <int CG, int PPT, bool NT>
__global__ __launch_bounds__(256) void k_render_fwd_sparse(FwdArgs a, int ncomp, int gy, int fsplit, int pb,
                                                           const uint32_t* __restrict__ cover)
{
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ int s_n, s_min;
    const int tid = threadIdx.x;
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const size_t HW = (size_t)H * W;

    // grid (fsplit + xc, bands, (C+1) * V): x < fsplit are the fill passes of one (band, plane, view) row, the xc extra
    // blocks per row carry the composite role (spread through the dispatch order so that their latency -- list
    // building, barriers -- overlaps the streaming stores).  No integer division on the fill path except z / (C+1).
    const int xq = blockIdx.x, band_id = blockIdx.y, zid = blockIdx.z;
    const int xc = gridDim.x - fsplit;
    const bool comp_role = xq >= fsplit;
    if (comp_role) {
        // ---------------- composite role ----------------
        const int cb = (zid * gy + band_id) * xc + (xq - fsplit);
        if (cb >= ncomp) return;
        const int slot = cb % T_SLOTS;
        const int g = (cb / T_SLOTS) % P;
        const int v = cb / (T_SLOTS * P);
        const size_t go = (size_t)v * P;
        const float4* gco = a.g.co + go;
        const float4* gxyd = a.g.xyd + go;
        const uint4* grect = a.g.rect + go;
        const uint4 rg = grect[g];
        const int wt = (int)(rg.z - rg.x), ht = (int)(rg.w - rg.y);
        const int ntiles = wt * ht;
        if (slot >= ntiles) return;
        const int cap = (P + 15) & ~15;
        DynList<CG> L(s_dyn, cap);
        if (tid < P) L.rect[tid] = grect[tid];
        __syncthreads();
        float* outc = a.out_color + (size_t)v * C * HW;
        float* outi = a.out_invdepth + (size_t)v * HW;
        const bool do_clamp = a.flags & SKS_CLAMP01;
        for (int t = slot; t < ntiles; t += T_SLOTS) {
            const int ty = (int)rg.y + t / wt, tx = (int)rg.x + t % wt;
            if (tid == 0) { s_n = 0; s_min = 0x7fffffff; }
            __syncthreads();
            if (tid < P) {
                const uint4 r = L.rect[tid];
                if ((int)r.x <= tx && tx < (int)r.z && (int)r.y <= ty && ty < (int)r.w) {
                    atomicMin(&s_min, tid);
                    const int sl = atomicAdd(&s_n, 1);
                    L.key[sl] = ((unsigned long long)__float_as_uint(gxyd[tid].z) << 32) | (unsigned)tid;
                }
            }
            __syncthreads();
            const int n = s_n;
            if (s_min == g) {  // uniform: this block owns the tile
                if (tid < n) {
                    const unsigned long long key = L.key[tid];
                    int rank = 0;
                    for (int j = 0; j < n; j++) rank += (L.key[j] < key) ? 1 : 0;
                    const int id = (int)(unsigned)key;
                    const float4 xyd = gxyd[id];
                    L.xy[rank] = make_float2(xyd.x, xyd.y);
                    L.co[rank] = gco[id];
                    L.invd[rank] = xyd.w;
#pragma unroll
                    for (int ch = 0; ch < CG; ch++) L.feat[rank * CG + ch] = ch < C ? a.features[id * C + ch] : 0.0f;
                }
                __syncthreads();
                const int x = tx * TILE + (tid & 15), y = ty * TILE + (tid >> 4);
                if (x < W && y < H) {
                    const float pxf = (float)x, pyf = (float)y;
                    float T = 1.0f, inv = 0.0f, acc[CG];
#pragma unroll
                    for (int ch = 0; ch < CG; ch++) acc[ch] = 0.0f;
                    uint32_t contributor = 0, last = 0;
                    for (int k = 0; k < n; k++) {  // forward.cu:346-386
                        contributor++;
                        const float2 xy = L.xy[k];
                        const float4 co = L.co[k];
                        const float dx = xy.x - pxf, dy = xy.y - pyf;
                        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = fminf(0.99f, co.w * expf_fixed(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break;
#pragma unroll
                        for (int ch = 0; ch < CG; ch++) acc[ch] += L.feat[k * CG + ch] * alpha * T;
                        inv += L.invd[k] * alpha * T;
                        T = test_T;
                        last = contributor;
                    }
                    const size_t pix = (size_t)y * W + x;
#pragma unroll
                    for (int ch = 0; ch < CG; ch++)
                        if (ch < C) outc[(size_t)ch * HW + pix] = do_clamp ? clamp01(acc[ch]) : acc[ch];
                    outi[pix] = inv;
                    if (a.final_T) a.final_T[(size_t)v * HW + pix] = T;
                    if (a.n_contrib) a.n_contrib[(size_t)v * HW + pix] = last;
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---------------- fill role ----------------
    fwd_fill_role<PPT, NT>(a, xq, band_id, zid, gy, pb, cover);
}

// ------------------------------------------------------------------------------------------------------------
// backward compositing core (backward.cu:531-637) for ONE pixel over one LDS batch, back to front.
// All lanes run the k loop together; per-entry partial sums are reduced over the wavefront and added to the
// workgroup's LDS accumulators s_acc[k][NVL].
// ------------------------------------------------------------------------------------------------------------
template <int CG>
struct BwdPix {
    float T, T_final, last_alpha, accum_inv, last_inv, dLi, bgdot;
    float dL[CG], accum_rec[CG], last_color[CG];
};

template <int CG, bool XF, bool DFEAT>
__device__ __forceinline__ void bwd_sweep(const List<CG>& L, int n, int klast /* last accepted LDS index or -1 */,
                                          float pxf, float pyf, int tx, float ddelx_dx, float ddely_dy,
                                          BwdPix<CG>& s, float* s_acc, int nc = CG /* staged (compacted) channels */)
{
    constexpr int NVL = NACC + (DFEAT ? CG : 0);
    const int lane = threadIdx.x & 63;
    for (int k = n - 1; k >= 0; k--) {
        bool act = k <= klast;
        float dx = 0, dy = 0, G = 0, alpha = 0;
        float4 co = make_float4(0, 0, 0, 0);
        if (act) {
            if (XF) {
                const int xr = L.xr[k];
                act = !(tx < (xr & 0xffff) || tx >= (xr >> 16));
            }
            if (act) {
                const float2 xy = L.xy[k];
                co = L.co[k];
                dx = xy.x - pxf;
                dy = xy.y - pyf;
                const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                act = !(power > 0.0f);
                if (act) {
                    G = expf_fixed(power);
                    alpha = fminf(0.99f, co.w * G);
                    act = !(alpha < 1.0f / 255.0f);
                }
            }
        }
        if (!__any(act)) continue;  // wave-uniform
        float vals[NVL];
#pragma unroll
        for (int j = 0; j < NVL; j++) vals[j] = 0.0f;
        if (act) {
            s.T = s.T / (1.f - alpha);
            const float dchannel_dcolor = alpha * s.T;
            float dL_dalpha = 0.0f;
#pragma unroll
            for (int ch = 0; ch < CG; ch++) {
                if (ch >= nc) break;  // wave-uniform
                const float c = L.feat[k * CG + ch];
                s.accum_rec[ch] = s.last_alpha * s.last_color[ch] + (1.f - s.last_alpha) * s.accum_rec[ch];
                s.last_color[ch] = c;
                const float dL_dchannel = s.dL[ch];
                dL_dalpha += (c - s.accum_rec[ch]) * dL_dchannel;
                if (DFEAT) vals[NACC + ch] = dchannel_dcolor * dL_dchannel;
            }
            const float invd = L.invd[k];
            s.accum_inv = s.last_alpha * s.last_inv + (1.f - s.last_alpha) * s.accum_inv;
            s.last_inv = invd;
            dL_dalpha += (invd - s.accum_inv) * s.dLi;
            vals[6] = dchannel_dcolor * s.dLi;
            dL_dalpha *= s.T;
            s.last_alpha = alpha;
            dL_dalpha += (-s.T_final / (1.f - alpha)) * s.bgdot;
            const float dL_dG = co.w * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            const float dG_ddelx = -gdx * co.x - gdy * co.y;
            const float dG_ddely = -gdy * co.z - gdx * co.y;
            vals[0] = dL_dG * dG_ddelx * ddelx_dx;
            vals[1] = dL_dG * dG_ddely * ddely_dy;
            vals[2] = -0.5f * gdx * dx * dL_dG;
            vals[3] = -0.5f * gdx * dy * dL_dG;
            vals[4] = -0.5f * gdy * dy * dL_dG;
            vals[5] = G * dL_dalpha;
        }
        static_assert(NVL % 8 == 0 && NACC == 8, "wave_sum8 works on groups of eight values");
#pragma unroll
        for (int j0 = 0; j0 < NVL; j0 += 8) {   // eight sums at a time; lane j of 0..7 ends up with the total of value j0 + j
            float grp[8];
#pragma unroll
            for (int j = 0; j < 8; j++) grp[j] = vals[j0 + j];
            const float r = wave_sum8(grp);
            if (lane < 8 && !(j0 == 0 && lane == 7)) atomicAdd(&s_acc[k * NVL + j0 + lane], r);
        }
    }
}

struct BwdArgs {
    int P, C, W, H;
    unsigned flags;
    Geom g;
    const float* features;
    const float* bg;
    const float* dL_color;     // fused-loss mode: the pseudo-GT heat-maps (V,C,H,W) instead of dL/d(render)
    const float* dL_invdepth;
    float* accum;  // (V,P,NACC+C)
    const float* tile_S;       // (unused since the loss corrections are taken where the render is positive; kept
    const float* tile_N;       //  for the ABI's argument list)
};

// prepass of one pixel over an LDS batch: same walk as the forward; records the LDS index of the last accepted
// entry (backward.cu:505-506 reads it from n_contrib instead).
template <int CG, bool XF>
__device__ __forceinline__ void bwd_prepass(const List<CG>& L, int n, float pxf, float pyf, int tx, float& T,
                                            float (&acc)[CG], bool want_color, int& klast, bool& done)
{
    for (int k = 0; k < n && !done; k++) {
        if (XF) {
            const int xr = L.xr[k];
            if (tx < (xr & 0xffff) || tx >= (xr >> 16)) continue;
        }
        const float2 xy = L.xy[k];
        const float4 co = L.co[k];
        const float dx = xy.x - pxf, dy = xy.y - pyf;
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co.w * expf_fixed(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) {
            done = true;
            continue;
        }
        if (want_color) {
#pragma unroll
            for (int ch = 0; ch < CG; ch++) acc[ch] += L.feat[k * CG + ch] * alpha * T;
        }
        T = test_T;
        klast = k;
    }
}

// chan == nullptr: identity channel map (position j is channel j), nc == C
template <int CG>
__device__ __forceinline__ void bwd_load_pixel(BwdPix<CG>& s, const BwdArgs& a, int v, size_t pix, size_t HW,
                                               const float (&col)[CG], bool do_clamp, float T_final,
                                               const int* chan = nullptr, int nc = -1)
{
    if (nc < 0) nc = a.C;
    s.T = T_final;
    s.T_final = T_final;
    s.last_alpha = 0;
    s.accum_inv = 0;
    s.last_inv = 0;
    s.bgdot = 0;
    const float* dLc = a.dL_color + (size_t)v * a.C * HW + pix;
#pragma unroll
    for (int j = 0; j < CG; j++) {
        float d = 0.0f;
        if (j < nc) {
            const int ch = chan ? chan[j] : j;
            d = dLc[(size_t)ch * HW];
            // torch.clamp backward passes the gradient where min <= x <= max
            if (do_clamp && !(col[j] >= 0.0f && col[j] <= 1.0f)) d = 0.0f;
            if (a.bg) s.bgdot += a.bg[ch] * d;
        }
        s.dL[j] = d;
        s.accum_rec[j] = 0;
        s.last_color[j] = 0;
    }
    s.dLi = a.dL_invdepth ? a.dL_invdepth[(size_t)v * HW + pix] : 0.0f;
}

// ------------------------------------------------------------------------------------------------------------
// small path backward, gathered per Gaussian: grid (BWD_SPLITS, P, V).  Workgroup (s, g, v) walks the pixels of
// Gaussian g's tile rect (256-pixel chunks s, s+BWD_SPLITS, ...), re-composites each pixel over the LDS list of
// every Gaussian whose rect meets g's rect, walks back to front like backward.cu:531-637 and keeps ONLY g's terms.
// Sums stay in registers, are reduced once per workgroup and stored to the (v, g, s) partial slot: no atomics,
// no zero-initialised scratch, bitwise reproducible gradients.
// ------------------------------------------------------------------------------------------------------------
// dynamic-LDS working set of the gather kernel, capacity `cap` = P rounded up to 16
template <int CG>
struct GatherLds {
    // raw records of ALL Gaussians of the view, indexed by Gaussian id (one global round trip fills them)
    float4* r_co;
    float4* r_xyd;
    uint4* r_rect;
    float* r_feat;  // cap * C
    unsigned long long* key;  // cap
    // the local list, ordered by (depth bits, index)
    float2* xy;
    float4* co;
    float* invd;
    int* xr;
    int* yr;
    int* id;
    float* feat;  // cap * CG, compacted channels
    __device__ GatherLds(char* p, int cap, int C)
    {
        r_co = (float4*)p; p += (size_t)cap * 16;
        r_xyd = (float4*)p; p += (size_t)cap * 16;
        r_rect = (uint4*)p; p += (size_t)cap * 16;
        co = (float4*)p; p += (size_t)cap * 16;
        key = (unsigned long long*)p; p += (size_t)cap * 8;
        xy = (float2*)p; p += (size_t)cap * 8;
        invd = (float*)p; p += (size_t)cap * 4;
        xr = (int*)p; p += (size_t)cap * 4;
        yr = (int*)p; p += (size_t)cap * 4;
        id = (int*)p; p += (size_t)cap * 4;
        feat = (float*)p; p += (size_t)cap * CG * 4;
        r_feat = (float*)p;
    }
    static size_t bytes(int cap, int cg, int C) { return (size_t)cap * (16 * 4 + 8 * 2 + 4 * 4 + 4 * (size_t)cg + 4 * (size_t)C); }
};

template <int CG, bool DFEAT>
__global__ __launch_bounds__(256) void k_render_bwd_gather(BwdArgs a)
{
    constexpr int NV = NACC + (DFEAT ? CG : 0);
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ float s_red[4][NV];
    __shared__ int s_chan[CG], s_act[CG];
    const int sp = blockIdx.x, g = blockIdx.y, v = blockIdx.z, tid = threadIdx.x;
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const int NVS = NACC + C;
    const size_t HW = (size_t)H * W;
    const size_t go = (size_t)v * P;
    const float4* gco = a.g.co + go;
    const float4* gxyd = a.g.xyd + go;
    const uint4* grect = a.g.rect + go;
    // partial-sum slots, layout (v, g, value, split): the BWD_SPLITS partials of one value are one 64-byte line
    float* out = a.accum + ((size_t)v * P + g) * NVS * BWD_SPLITS + sp;
    const uint4 rg = grect[g];
    const int x0 = (int)rg.x * TILE, y0 = (int)rg.y * TILE;
    const int wpx = min(W, (int)rg.z * TILE) - x0, hpx = min(H, (int)rg.w * TILE) - y0;
    const int npx = wpx * hpx;  // 0 when culled (rect all zero)
    const int nchunks = (npx + 255) / 256;
    if (sp >= nchunks) {  // nothing for this slot (also every slot of an invisible Gaussian)
        if (tid < NVS) out[tid * BWD_SPLITS] = 0.0f;
        return;
    }
    // one global round trip: every Gaussian's record + features into LDS
    const int cap = (P + 15) & ~15;
    GatherLds<CG> L(s_dyn, cap, C);
    if (tid < P) {
        L.r_co[tid] = gco[tid];
        L.r_xyd[tid] = gxyd[tid];
        L.r_rect[tid] = grect[tid];
    }
    for (int i = tid; i < P * C; i += 256) L.r_feat[i] = a.features[i];
    __syncthreads();
    // local list: Gaussians whose tile rect intersects g's, ordered by (depth bits, index)
    unsigned long long mykey = ~0ull;
    if (tid < P) {
        const uint4 r = L.r_rect[tid];
        if (r.x < rg.z && r.z > rg.x && r.y < rg.w && r.w > rg.y)
            mykey = ((unsigned long long)__float_as_uint(L.r_xyd[tid].z) << 32) | (unsigned)tid;
        L.key[tid] = mykey;
    }
    __syncthreads();
    int n = 0, rank = 0;
    for (int j = 0; j < P; j++) {
        const unsigned long long kj = L.key[j];
        n += kj != ~0ull ? 1 : 0;
        rank += kj < mykey ? 1 : 0;
    }
    if (mykey != ~0ull) {
        const float4 xyd = L.r_xyd[tid];
        const uint4 r = L.r_rect[tid];
        L.xy[rank] = make_float2(xyd.x, xyd.y);
        L.co[rank] = L.r_co[tid];
        L.invd[rank] = xyd.w;
        L.xr[rank] = (int)(r.x | (r.z << 16));
        L.yr[rank] = (int)(r.y | (r.w << 16));
        L.id[rank] = tid;
    }
    __syncthreads();
    // channel compaction: a channel whose feature is zero for every listed Gaussian multiplies dL/dpixel by exact
    // zeros everywhere in backward.cu:581-594, so its (dense, far-apart) gradient plane is never read.
    // All channels stay active when the feature gradient or the background term needs every dL/dpixel.
    if (tid < CG) {
        bool act = false;
        if (tid < C) {
            act = DFEAT || a.bg != nullptr;
            for (int k = 0; k < n && !act; k++) act = L.r_feat[L.id[k] * C + tid] != 0.0f;
        }
        s_act[tid] = act ? 1 : 0;
    }
    __syncthreads();
    if (tid < CG && s_act[tid]) {
        int pos = 0;
        for (int c = 0; c < tid; c++) pos += s_act[c];
        s_chan[pos] = tid;  // ascending channel order: summation order over channels is fixed
    }
    int nc = 0;
    for (int c = 0; c < CG; c++) nc += s_act[c];
    __syncthreads();
    if (tid < n) {
        const int id = L.id[tid];
#pragma unroll
        for (int j = 0; j < CG; j++) L.feat[tid * CG + j] = j < nc ? L.r_feat[id * C + s_chan[j]] : 0.0f;
    }
    __syncthreads();
    int kg = 0;
    for (int k = 0; k < n; k++) kg = L.id[k] == g ? k : kg;
    const bool do_clamp = a.flags & SKS_CLAMP01;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);  // backward.cu:527-528
    float sum[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) sum[j] = 0.0f;

    for (int c = sp; c < nchunks; c += BWD_SPLITS) {
        const int i = c * 256 + tid;
        if (i >= npx) continue;
        const int yy = i / wpx;
        const int x = x0 + (i - yy * wpx), y = y0 + yy;
        const int tx = x >> 4, ty = y >> 4;
        const float pxf = (float)x, pyf = (float)y;
        // re-composite front to back (forward.cu:346-386): T_final, last accepted entry, colours if clamping
        float T = 1.0f, col[CG];
#pragma unroll
        for (int ch = 0; ch < CG; ch++) col[ch] = 0.0f;
        int klast = -1;
        for (int k = 0; k < n; k++) {
            const int xr = L.xr[k], yr = L.yr[k];
            if (tx < (xr & 0xffff) || tx >= (xr >> 16) || ty < (yr & 0xffff) || ty >= (yr >> 16)) continue;
            const float2 xy = L.xy[k];
            const float4 co = L.co[k];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, co.w * expf_fixed(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) break;
            if (do_clamp) {
#pragma unroll
                for (int ch = 0; ch < CG; ch++) col[ch] += L.feat[k * CG + ch] * alpha * T;
            }
            T = test_T;
            klast = k;
        }
        if (klast < kg) continue;  // g is behind the last contributor (or nothing contributes) at this pixel
        // upstream gradient of this pixel: only now, only the active channels (most rect pixels never get here)
        const size_t pix = (size_t)y * W + x;
        float dLraw[CG];
        {
            const float* dLc = a.dL_color + (size_t)v * C * HW + pix;
#pragma unroll
            for (int j = 0; j < CG; j++) dLraw[j] = j < nc ? dLc[(size_t)s_chan[j] * HW] : 0.0f;
        }
        const float dLi_raw = a.dL_invdepth ? a.dL_invdepth[(size_t)v * HW + pix] : 0.0f;
        BwdPix<CG> s;
        s.T = T;
        s.T_final = T;
        s.last_alpha = 0;
        s.accum_inv = 0;
        s.last_inv = 0;
        s.bgdot = 0;
        s.dLi = dLi_raw;
#pragma unroll
        for (int j = 0; j < CG; j++) {
            float d = dLraw[j];
            // torch.clamp backward passes the gradient where min <= x <= max
            if (do_clamp && !(col[j] >= 0.0f && col[j] <= 1.0f)) d = 0.0f;
            if (a.bg && j < nc) s.bgdot += a.bg[s_chan[j]] * d;
            s.dL[j] = d;
            s.accum_rec[j] = 0;
            s.last_color[j] = 0;
        }
        // back to front down to g (backward.cu:552-636); only g's own terms are kept
        for (int k = klast; k >= kg; k--) {
            const int xr = L.xr[k], yr = L.yr[k];
            if (tx < (xr & 0xffff) || tx >= (xr >> 16) || ty < (yr & 0xffff) || ty >= (yr >> 16)) continue;
            const float2 xy = L.xy[k];
            const float4 co = L.co[k];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            if (power > 0.0f) continue;
            const float G = expf_fixed(power);
            const float alpha = fminf(0.99f, co.w * G);
            if (alpha < 1.0f / 255.0f) continue;
            s.T = s.T / (1.f - alpha);
            const float dchannel_dcolor = alpha * s.T;
            float dL_dalpha = 0.0f;
            const bool mine = k == kg;
#pragma unroll
            for (int ch = 0; ch < CG; ch++) {
                const float cc = L.feat[k * CG + ch];
                s.accum_rec[ch] = s.last_alpha * s.last_color[ch] + (1.f - s.last_alpha) * s.accum_rec[ch];
                s.last_color[ch] = cc;
                const float dL_dchannel = s.dL[ch];
                dL_dalpha += (cc - s.accum_rec[ch]) * dL_dchannel;
                if (DFEAT && mine) sum[NACC + ch] += dchannel_dcolor * dL_dchannel;
            }
            const float invd = L.invd[k];
            s.accum_inv = s.last_alpha * s.last_inv + (1.f - s.last_alpha) * s.accum_inv;
            s.last_inv = invd;
            dL_dalpha += (invd - s.accum_inv) * s.dLi;
            dL_dalpha *= s.T;
            s.last_alpha = alpha;
            dL_dalpha += (-s.T_final / (1.f - alpha)) * s.bgdot;
            if (mine) {
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                sum[0] += dL_dG * dG_ddelx * ddelx_dx;
                sum[1] += dL_dG * dG_ddely * ddely_dy;
                sum[2] += -0.5f * gdx * dx * dL_dG;
                sum[3] += -0.5f * gdx * dy * dL_dG;
                sum[4] += -0.5f * gdy * dy * dL_dG;
                sum[5] += G * dL_dalpha;
                sum[6] += dchannel_dcolor * s.dLi;
            }
        }
    }
    // workgroup reduction in a fixed order
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const float r = wave_sum(sum[j]);
        if (lane == 0) s_red[wv][j] = r;
    }
    __syncthreads();
    if (tid < NVS)
        out[tid * BWD_SPLITS] = tid < NV ? ((s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid])) : 0.0f;
}

// ------------------------------------------------------------------------------------------------------------
// small path backward for P <= 64, wave-resident: same gather-by-Gaussian scheme as k_render_bwd_gather, but the
// whole working set lives in registers.  Lane i of every wavefront holds Gaussian i's record; the local list
// (Gaussians whose rect meets g's, ordered by (depth bits, index)) is a ballot mask + a per-lane rank, and the entry
// being composited is broadcast with v_readlane (wave-uniform values in SGPRs): no LDS and no barrier until the final
// cross-wave reduction, which removes the latency chain that dominates a 17-Gaussian backward.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rl(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ unsigned rlu(unsigned x, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)x, lane); }

// LOSS = true fuses the loop's masked-L2 loss (utils/loss_utils.py:86-100 on the clamped render, train.py:150) into
// this kernel: a.dL_color holds the pseudo-GT heat-maps; dL/d(render) = 2 (r - gt) on the mask {gt > 0 or r > 0} is
// formed per pixel from the re-composited colours, so neither the rendered image nor a dense gradient ever exists.
// The loss sums over the WHOLE image are (per-view totals of the constant heat-maps: sum gt^2 and count over gt > 0,
// i.e. the loss of an all-zero render) + (corrections where the render is positive), the latter accumulated here by
// each tile's owner (lowest-index covering Gaussian) into slots 7 (S) and 8 (N).
template <int CG, bool DFEAT, bool LOSS>
__global__ __launch_bounds__(256) void k_render_bwd_wave(BwdArgs a)
{
    static_assert(!(DFEAT && LOSS), "the fused-loss variant has no feature gradient");
    constexpr int NV = LOSS ? NACC + 1 : NACC + (DFEAT ? CG : 0);
    __shared__ float s_red[4][NV];
    const int sp = blockIdx.x, g = blockIdx.y, v = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const int NVS = NACC + C;
    const size_t HW = (size_t)H * W;
    const size_t go = (size_t)v * P;
    float* out = a.accum + ((size_t)v * P + g) * NVS * BWD_SPLITS + sp;
    // lane i <- Gaussian i (one global round trip, every load independent)
    const bool has = lane < P;
    const int li = has ? lane : 0;
    const float4 m_co = a.g.co[go + li];
    const float4 m_xyd = a.g.xyd[go + li];
    uint4 m_rect = a.g.rect[go + li];
    if (!has) m_rect = make_uint4(0, 0, 0, 0);
    float m_feat[CG];
#pragma unroll
    for (int ch = 0; ch < CG; ch++) m_feat[ch] = (ch < C && has) ? a.features[li * C + ch] : 0.0f;

    const unsigned gx0 = rlu(m_rect.x, g), gy0 = rlu(m_rect.y, g), gx1 = rlu(m_rect.z, g), gy1 = rlu(m_rect.w, g);
    const int x0 = (int)gx0 * TILE, y0 = (int)gy0 * TILE;
    const int wpx = min(W, (int)gx1 * TILE) - x0, hpx = min(H, (int)gy1 * TILE) - y0;
    const int npx = wpx * hpx;  // 0 when culled (rect all zero)
    const int nchunks = (npx + 255) / 256;
    if (sp >= nchunks) {  // nothing for this slot (also every slot of an invisible Gaussian)
        if (tid < NVS) out[tid * BWD_SPLITS] = 0.0f;
        return;
    }
    // local list = lanes whose rect intersects g's; order = rank of (depth bits, index) among them
    const bool hit = m_rect.x < gx1 && m_rect.z > gx0 && m_rect.y < gy1 && m_rect.w > gy0;
    const unsigned long long hm = __ballot(hit);
    const int n = __popcll(hm);
    const unsigned mydepth = __float_as_uint(m_xyd.z);
    int rank = 0;
    for (unsigned long long m = hm; m; m &= m - 1) {
        const int j = __builtin_ctzll(m);
        const unsigned dj = rlu(mydepth, j);
        rank += (dj < mydepth || (dj == mydepth && j < lane)) ? 1 : 0;
    }
    const int kg = __builtin_amdgcn_readlane(rank, g);
    // channels that can matter (see k_render_bwd_gather): a bit mask, uniform across the wavefront
    unsigned chmask = 0;
#pragma unroll
    for (int ch = 0; ch < CG; ch++) {
        if (ch < C) {
            const bool on = (DFEAT || a.bg != nullptr) ? true : (__ballot(hit && m_feat[ch] != 0.0f) != 0ull);
            chmask |= on ? (1u << ch) : 0u;
        }
    }
    const bool do_clamp = a.flags & SKS_CLAMP01;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);  // backward.cu:527-528
    float sum[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) sum[j] = 0.0f;

    for (int c = sp; c < nchunks; c += BWD_SPLITS) {
        const int i = c * 256 + tid;
        const bool in = i < npx;
        const int ii = in ? i : 0;
        const int yy = ii / wpx;
        const int x = x0 + (ii - yy * wpx), y = y0 + yy;
        const int tx = x >> 4, ty = y >> 4;
        const float pxf = (float)x, pyf = (float)y;
        // re-composite front to back (forward.cu:346-386): T_final, last accepted entry, colours if clamping.
        // The k loop is wave-uniform (entries are broadcast); per-lane state is predicated.
        float T = 1.0f, col[CG];
#pragma unroll
        for (int ch = 0; ch < CG; ch++) col[ch] = 0.0f;
        int klast = -1;
        int minid = 0x7fffffff;  // lowest Gaussian index whose rect covers this pixel's tile (the tile's owner)
        bool alive = in;
        for (int k = 0; k < n; k++) {
            const int lk = __builtin_ctzll(__ballot(hit && rank == k));
            const int ex0 = (int)rlu(m_rect.x, lk), ey0 = (int)rlu(m_rect.y, lk), ex1 = (int)rlu(m_rect.z, lk), ey1 = (int)rlu(m_rect.w, lk);
            const float ex = rl(m_xyd.x, lk), ey = rl(m_xyd.y, lk);
            const float cx = rl(m_co.x, lk), cy = rl(m_co.y, lk), cz = rl(m_co.z, lk), cw = rl(m_co.w, lk);
            const bool covers = !(tx < ex0 || tx >= ex1 || ty < ey0 || ty >= ey1);
            if (LOSS && covers) minid = min(minid, lk);
            if (!LOSS && !__any(alive)) break;
            const bool cov = alive && covers;
            const float dx = ex - pxf, dy = ey - pyf;
            const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
            const float alpha = fminf(0.99f, cw * expf_fixed(power));
            const float test_T = T * (1 - alpha);
            const bool pass = cov && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            const bool stop = pass && test_T < 0.0001f;
            const bool acc = pass && !stop;
            if (do_clamp || LOSS) {
#pragma unroll
                for (int ch = 0; ch < CG; ch++) {
                    if (chmask & (1u << ch)) {
                        const float f = rl(m_feat[ch], lk);
                        if (acc) col[ch] += f * alpha * T;
                    }
                }
            }
            if (acc) { T = test_T; klast = k; }
            if (stop) alive = false;
        }
        const bool need = in && klast >= kg;  // g is at or in front of the last contributor at this pixel
        // LOSS: the pixel's loss terms are accounted by the tile's owner, and only where the render is positive:
        // everywhere else the masked-L2 sees render = 0, which the per-view heat-map totals already contain
        bool own = false;
        if (LOSS && in && minid == g) {
#pragma unroll
            for (int ch = 0; ch < CG; ch++)
                if (chmask & (1u << ch)) own = own || col[ch] > 0.0f;
        }
        if (!__any(need || own)) continue;
        // upstream gradient of this pixel: only the active channels, only the lanes that need it
        const size_t pix = (size_t)y * W + x;
        BwdPix<CG> s;
        s.T = T; s.T_final = T; s.last_alpha = 0; s.accum_inv = 0; s.last_inv = 0; s.bgdot = 0;
        s.dLi = (!LOSS && need && a.dL_invdepth) ? a.dL_invdepth[(size_t)v * HW + pix] : 0.0f;
        {
            const float* dLc = a.dL_color + (size_t)v * C * HW + pix;
#pragma unroll
            for (int ch = 0; ch < CG; ch++) {
                float d = 0.0f;
                if (chmask & (1u << ch)) {
                    if (LOSS) {
                        const bool rpos = col[ch] > 0.0f;          // clamp01(col) > 0
                        const float gtv = (need || (own && rpos)) ? dLc[(size_t)ch * HW] : 0.0f;
                        const float r = clamp01(col[ch]);          // gaussian_renderer/__init__.py:129
                        const bool msk = gtv > 0.0f || r > 0.0f;   // loss_utils.py:88-91
                        const float e = r - gtv;
                        // correction to the heat-map-only totals: (r - gt)^2 replaces gt^2 [gt > 0], 1 replaces [gt > 0]
                        if (own && rpos) { sum[7] += e * e - (gtv > 0.0f ? gtv * gtv : 0.0f); sum[8] += gtv > 0.0f ? 0.0f : 1.0f; }
                        d = msk ? 2.0f * e : 0.0f;
                        if (!(col[ch] >= 0.0f && col[ch] <= 1.0f)) d = 0.0f;   // clamp backward
                        if (!need) d = 0.0f;
                    } else {
                        if (need) d = dLc[(size_t)ch * HW];
                        // torch.clamp backward passes the gradient where min <= x <= max
                        if (do_clamp && !(col[ch] >= 0.0f && col[ch] <= 1.0f)) d = 0.0f;
                    }
                    if (a.bg) s.bgdot += a.bg[ch] * d;
                }
                s.dL[ch] = d;
                s.accum_rec[ch] = 0;
                s.last_color[ch] = 0;
            }
        }
        if (!__any(need)) continue;
        // back to front down to g (backward.cu:552-636); only g's own terms are kept
        for (int k = n - 1; k >= kg; k--) {
            const int lk = __builtin_ctzll(__ballot(hit && rank == k));
            const int ex0 = (int)rlu(m_rect.x, lk), ey0 = (int)rlu(m_rect.y, lk), ex1 = (int)rlu(m_rect.z, lk), ey1 = (int)rlu(m_rect.w, lk);
            const float ex = rl(m_xyd.x, lk), ey = rl(m_xyd.y, lk), einvd = rl(m_xyd.w, lk);
            const float cx = rl(m_co.x, lk), cy = rl(m_co.y, lk), cz = rl(m_co.z, lk), cw = rl(m_co.w, lk);
            const float dx = ex - pxf, dy = ey - pyf;
            const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
            const float G = expf_fixed(power);
            const float alpha = fminf(0.99f, cw * G);
            const bool act = need && k <= klast && !(tx < ex0 || tx >= ex1 || ty < ey0 || ty >= ey1) && !(power > 0.0f) &&
                             !(alpha < 1.0f / 255.0f);
            if (!__any(act)) continue;
            const bool mine = k == kg;
            float Tn = s.T, dchannel_dcolor = 0.0f, dL_dalpha = 0.0f;
            if (act) {
                Tn = s.T / (1.f - alpha);
                dchannel_dcolor = alpha * Tn;
            }
#pragma unroll
            for (int ch = 0; ch < CG; ch++) {
                if (chmask & (1u << ch)) {
                    const float cc = rl(m_feat[ch], lk);
                    if (act) {
                        s.accum_rec[ch] = s.last_alpha * s.last_color[ch] + (1.f - s.last_alpha) * s.accum_rec[ch];
                        s.last_color[ch] = cc;
                        const float dL_dchannel = s.dL[ch];
                        dL_dalpha += (cc - s.accum_rec[ch]) * dL_dchannel;
                        if (DFEAT && mine) sum[NACC + ch] += dchannel_dcolor * dL_dchannel;
                    }
                }
            }
            if (act) {
                s.T = Tn;
                s.accum_inv = s.last_alpha * s.last_inv + (1.f - s.last_alpha) * s.accum_inv;
                s.last_inv = einvd;
                dL_dalpha += (einvd - s.accum_inv) * s.dLi;
                dL_dalpha *= s.T;
                s.last_alpha = alpha;
                dL_dalpha += (-s.T_final / (1.f - alpha)) * s.bgdot;
                if (mine) {
                    const float dL_dG = cw * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * cx - gdy * cy;
                    const float dG_ddely = -gdy * cz - gdx * cy;
                    sum[0] += dL_dG * dG_ddelx * ddelx_dx;
                    sum[1] += dL_dG * dG_ddely * ddely_dy;
                    sum[2] += -0.5f * gdx * dx * dL_dG;
                    sum[3] += -0.5f * gdx * dy * dL_dG;
                    sum[4] += -0.5f * gdy * dy * dL_dG;
                    sum[5] += G * dL_dalpha;
                    sum[6] += dchannel_dcolor * s.dLi;
                }
            }
        }
    }
    // workgroup reduction in a fixed order
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const float r = wave_sum(sum[j]);
        if (lane == 0) s_red[wv][j] = r;
    }
    __syncthreads();
    if (tid < NVS)
        out[tid * BWD_SPLITS] = tid < NV ? ((s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid])) : 0.0f;
}

// per (view, tile, channel) statistics of the constant pseudo-GT heat-maps, once per scene: sum of gt^2 and count of
// gt > 0 (what the masked-L2 loss sees wherever the render is zero), and their per-view totals.
// Streaming layout: block (band, channel, view); a thread owns one 4-pixel column of a tile and walks the tile's 16
// rows (16 independent 16-byte loads; a wavefront reads 1 KB contiguous per row), the four threads of a tile are
// combined with DPP quad permutes in a fixed order -- one pass over the heat-maps at streaming rate, reproducible sums.
__global__ __launch_bounds__(256) void k_gt_tile_stats(int C, int W, int H, const float* __restrict__ gt, float* __restrict__ tile_S,
                                                        float* __restrict__ tile_N, double* __restrict__ totals)
{
    __shared__ double s_t[2][4];
    const int band = blockIdx.x, ch = blockIdx.y, v = blockIdx.z, tid = threadIdx.x;
    const int gx = (W + TILE - 1) / TILE, gy = gridDim.x;
    const size_t HW = (size_t)H * W;
    const float* plane = gt + ((size_t)v * C + ch) * HW;
    const int y0 = band * TILE, rows = min(TILE, H - y0);
    const bool vec = (W & 3) == 0;
    double tS = 0.0, tN = 0.0;
    for (int t0 = 0; t0 < gx; t0 += 64) {
        const int tx = t0 + (tid >> 2), x = tx * TILE + (tid & 3) * 4;
        float S = 0.0f, N = 0.0f;
        if (tx < gx && x < W) {
            if (vec) {
                float4 g[TILE];
#pragma unroll
                for (int r = 0; r < TILE; r++)   // all loads first: 16 independent requests in flight per lane
                    g[r] = r < rows ? *reinterpret_cast<const float4*>(plane + (size_t)(y0 + r) * W + x) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < TILE; r++) {
                    S += (g[r].x * g[r].x + g[r].y * g[r].y) + (g[r].z * g[r].z + g[r].w * g[r].w);
                    N += ((g[r].x > 0.0f ? 1.0f : 0.0f) + (g[r].y > 0.0f ? 1.0f : 0.0f)) +
                         ((g[r].z > 0.0f ? 1.0f : 0.0f) + (g[r].w > 0.0f ? 1.0f : 0.0f));
                }
            } else {
                for (int r = 0; r < rows; r++)
                    for (int k = 0; k < 4; k++)
                        if (x + k < W) {
                            const float g = plane[(size_t)(y0 + r) * W + x + k];
                            S += g * g;
                            N += g > 0.0f ? 1.0f : 0.0f;
                        }
            }
        }
        // the four 4-pixel columns of the tile
        S += dpp_move<0xB1>(S); N += dpp_move<0xB1>(N);
        S += dpp_move<0x4E>(S); N += dpp_move<0x4E>(N);
        if ((tid & 3) == 0 && tx < gx) {
            const size_t tb = (((size_t)v * gy + band) * gx + tx) * C + ch;
            if (tile_S) tile_S[tb] = S;
            if (tile_N) tile_N[tb] = N;
            tS += (double)S;
            tN += (double)N;
        }
    }
    tS = wave_sum_d(tS);
    tN = wave_sum_d(tN);
    if ((tid & 63) == 0) { s_t[0][tid >> 6] = tS; s_t[1][tid >> 6] = tN; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&totals[2 * v], (s_t[0][0] + s_t[0][1]) + (s_t[0][2] + s_t[0][3]));
        atomicAdd(&totals[2 * v + 1], (s_t[1][0] + s_t[1][1]) + (s_t[1][2] + s_t[1][3]));
    }
}

// ------------------------------------------------------------------------------------------------------------
// geometry backward: computeCov2DCUDA (backward.cu:147-326) + preprocessCUDA (:398-449) + computeCov3D
// (:330-393) fused; consumes and clears the accumulators.  SH backward (:443-444) intentionally not reproduced
// (SURVEY quirk Q5): dL_dfeatures is the true dL_dcolors of backward.cu:593.
// ------------------------------------------------------------------------------------------------------------
struct GeomBwdArgs {
    int P, C, W, H;
    unsigned flags;
    const float* vms;
    const float* pms;
    const float* means;
    const float* opac;
    const float* scales;
    const float* rots;
    const float* cov3Dp;
    float smod;
    const int* radii;
    const float* accum;
    int nsplit;
    const double* gt_totals;   // fused-loss mode: per-view {sum gt^2, count gt > 0} over the whole image, else nullptr
    double* loss_sums;         // fused-loss mode: out, per-view {S, N} of the masked-L2 loss
    float* packed;             // optional (V,P,11) raw-parameter gradients (scaled by 1/N_v when loss_sums is set)
    float* dmeans3D;
    float* dmeans2D;
    float* dopacity;
    float* dscales;
    float* drots;
    float* dcov3D;
    float* dfeat;
};

__device__ __forceinline__ float sq(float x) { return x * x; }

// part 1 of one (view, Gaussian) pair: the per-split partial sums -> g[0..7], dL/dfeatures, and this Gaussian's share of
// the fused loss sums (slots 7 / 8)
__device__ __forceinline__ void geom_bwd_load(const GeomBwdArgs& a, int v, int idx, float (&g)[NACC], double& pS, double& pN)
{
    const size_t o = (size_t)v * a.P + (idx < a.P ? idx : 0);
    const int NVS = NACC + a.C;
    const float* acc = a.accum + o * a.nsplit * NVS;  // layout (value, split)
    if (a.nsplit == BWD_SPLITS) {
        static_assert(BWD_SPLITS == 16, "4 x float4 per value");
        float4 q[NACC][4];
#pragma unroll
        for (int j = 0; j < NACC - 1; j++)   // all loads in flight before the first add
#pragma unroll
            for (int i = 0; i < 4; i++) q[j][i] = reinterpret_cast<const float4*>(acc + j * BWD_SPLITS)[i];
#pragma unroll
        for (int j = 0; j < NACC - 1; j++) {  // fixed order: reproducible
            float t = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) t += (q[j][i].x + q[j][i].y) + (q[j][i].z + q[j][i].w);
            g[j] = t;
        }
        g[NACC - 1] = 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            float t = 0.0f;
            for (int sp = 0; sp < a.nsplit; sp++) t += acc[j * a.nsplit + sp];
            g[j] = t;
        }
    }
    if (a.dfeat) {
        for (int ch = 0; ch < a.C; ch++) {
            float t = 0.0f;
            for (int sp = 0; sp < a.nsplit; sp++) t += acc[(NACC + ch) * a.nsplit + sp];
            a.dfeat[o * a.C + ch] = t;
        }
    }
    pS = 0.0;
    pN = 0.0;
    if (a.loss_sums && idx < a.P) {
        for (int sp = 0; sp < a.nsplit; sp++) {
            pS += (double)acc[7 * a.nsplit + sp];
            pN += (double)acc[8 * a.nsplit + sp];
        }
    }
}

// part 2: the reference's formulas; n_view = the view's mask count N_v (fused-loss mode, else unused)
__device__ __forceinline__ void geom_bwd_finish(const GeomBwdArgs& a, const ViewTan& vt, int v, int idx, const float (&g)[NACC],
                                                double n_view)
{
    const size_t o = (size_t)v * a.P + idx;
    float dmean[3] = { 0, 0, 0 }, dcov[6] = { 0, 0, 0, 0, 0, 0 }, dscale[3] = { 0, 0, 0 }, dq[4] = { 0, 0, 0, 0 };
    float sc[3] = { 0, 0, 0 }, q[4] = { 1, 0, 0, 0 }, qnorm = 1.0f;   // activated scale / rotation (raw mode: see activate())
    float dop = g[5];
    const float dm2x = g[0], dm2y = g[1];

    if (a.radii[o] > 0) {
        const float* V = a.vms + 16 * v;
        const float* proj = a.pms + 16 * v;
        const float tan_fovx = vt.x[v], tan_fovy = vt.y[v];
        const float h_y = a.H / (2.0f * tan_fovy);
        const float h_x = a.W / (2.0f * tan_fovx);
        const float mean[3] = { a.means[3 * idx], a.means[3 * idx + 1], a.means[3 * idx + 2] };
        float cov3D[6];
        if (a.cov3Dp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = a.cov3Dp[6 * idx + i];
        } else {
            const float rs[3] = { a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2] };
            const float rq[4] = { a.rots[4 * idx], a.rots[4 * idx + 1], a.rots[4 * idx + 2], a.rots[4 * idx + 3] };
            const Activated ac = activate(a.flags & SKS_RAW_PARAMS, 0.0f, rs, rq);
            sc[0] = ac.s[0]; sc[1] = ac.s[1]; sc[2] = ac.s[2];
            q[0] = ac.q[0]; q[1] = ac.q[1]; q[2] = ac.q[2]; q[3] = ac.q[3];
            qnorm = ac.qnorm;
            computeCov3D(sc, a.smod, q, cov3D);
        }
        const float dL_dconic[3] = { g[2], g[3], g[4] };
        Cov2D c;
        cov2d(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, V, c);
        const float* t = c.t;
        const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
        const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
        const M3& T = c.T;
        const M3& Wm = c.W;
        const M3& Vrk = c.Vrk;
        float c_xx = c.cov.m[0][0], c_xy = c.cov.m[0][1], c_yy = c.cov.m[1][1];
        constexpr float h_var = 0.3f;
        float d_inside_root = 0.f;
        const bool aa = a.flags & SKS_ANTIALIASING;
        if (aa) {
            const float det_cov = c_xx * c_yy - c_xy * c_xy;
            c_xx += h_var;
            c_yy += h_var;
            const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
            const float h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
            const float dL_dopacity_v = dop;
            const float op_raw = a.opac[idx];
            const float d_h_convolution_scaling = dL_dopacity_v * ((a.flags & SKS_RAW_PARAMS) ? 1.0f / (1.0f + expf_fixed(-op_raw)) : op_raw);
            dop = dL_dopacity_v * h_convolution_scaling;
            d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
        } else {
            c_xx += h_var;
            c_yy += h_var;
        }
        float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
        if (aa) {
            const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
            const float denom_f = d_inside_root / sq(w * w + w * (x + y) + x * y - z * z);
            dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
            dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
            dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
        }
        const float denom = c_xx * c_yy - c_xy * c_xy;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_dc_xx += denom2inv * (-c_yy * c_yy * dL_dconic[0] + 2 * c_xy * c_yy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[2]);
            dL_dc_yy += denom2inv * (-c_xx * c_xx * dL_dconic[2] + 2 * c_xx * c_xy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[0]);
            dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dL_dconic[0] - (denom + 2 * c_xy * c_xy) * dL_dconic[1] + c_xx * c_xy * dL_dconic[2]);
            dcov[0] = (T.m[0][0] * T.m[0][0] * dL_dc_xx + T.m[0][0] * T.m[1][0] * dL_dc_xy + T.m[1][0] * T.m[1][0] * dL_dc_yy);
            dcov[3] = (T.m[0][1] * T.m[0][1] * dL_dc_xx + T.m[0][1] * T.m[1][1] * dL_dc_xy + T.m[1][1] * T.m[1][1] * dL_dc_yy);
            dcov[5] = (T.m[0][2] * T.m[0][2] * dL_dc_xx + T.m[0][2] * T.m[1][2] * dL_dc_xy + T.m[1][2] * T.m[1][2] * dL_dc_yy);
            dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_dc_xx + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][1] * dL_dc_yy;
            dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_dc_xx + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][2] * dL_dc_yy;
            dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_dc_xx + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_dc_xy + 2 * T.m[1][1] * T.m[1][2] * dL_dc_yy;
        }
        const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_xy;
        const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_xy;
        const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_xy;
        const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xy;
        const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xy;
        const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xy;
        const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
        const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
        const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
        const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
        const float tz = 1.f / t[2];
        const float tz2 = tz * tz;
        const float tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
        dL_dtz -= g[6] / (t[2] * t[2]);  // inverse-depth term, backward.cu:314-315 (always taken)
        // transformVec4x3Transpose, auxiliary.h:101-109
        dmean[0] = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
        dmean[1] = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
        dmean[2] = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;
        // preprocessCUDA backward, backward.cu:423-440
        float m_hom[4];
        transformPoint4x4(mean, proj, m_hom);
        const float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        const float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * dm2x + (proj[1] * m_w - proj[3] * mul2) * dm2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * dm2x + (proj[5] * m_w - proj[7] * mul2) * dm2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * dm2x + (proj[9] * m_w - proj[11] * mul2) * dm2y;
        // computeCov3D backward, backward.cu:330-393
        if (!a.cov3Dp) {
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            const M3 R = quatR(q);
            M3 S = {};
            const float s[3] = { a.smod * sc[0], a.smod * sc[1], a.smod * sc[2] };
            S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
            const M3 M = m3mul(S, R);
            M3 dS;
            dS.m[0][0] = dcov[0]; dS.m[0][1] = 0.5f * dcov[1]; dS.m[0][2] = 0.5f * dcov[2];
            dS.m[1][0] = 0.5f * dcov[1]; dS.m[1][1] = dcov[3]; dS.m[1][2] = 0.5f * dcov[4];
            dS.m[2][0] = 0.5f * dcov[2]; dS.m[2][1] = 0.5f * dcov[4]; dS.m[2][2] = dcov[5];
            M3 M2;
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) M2.m[cc][rr] = 2.0f * M.m[cc][rr];
            const M3 dL_dM = m3mul(M2, dS);
            const M3 Rt = m3T(R);
            M3 dMt = m3T(dL_dM);
#pragma unroll
            for (int k = 0; k < 3; k++)
                dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) dMt.m[k][rr] *= s[k];
            dq[0] = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
            dq[1] = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
            dq[2] = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
            dq[3] = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
        }
    }
    if (a.dmeans3D) { a.dmeans3D[3 * o] = dmean[0]; a.dmeans3D[3 * o + 1] = dmean[1]; a.dmeans3D[3 * o + 2] = dmean[2]; }
    if (a.dmeans2D) { a.dmeans2D[3 * o] = dm2x; a.dmeans2D[3 * o + 1] = dm2y; a.dmeans2D[3 * o + 2] = 0.0f; }
    if (a.dopacity) a.dopacity[o] = dop;
    if (a.dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) a.dcov3D[6 * o + i] = dcov[i];
    }
    if (a.dscales) { a.dscales[3 * o] = dscale[0]; a.dscales[3 * o + 1] = dscale[1]; a.dscales[3 * o + 2] = dscale[2]; }
    if (a.drots) { a.drots[4 * o] = dq[0]; a.drots[4 * o + 1] = dq[1]; a.drots[4 * o + 2] = dq[2]; a.drots[4 * o + 3] = dq[3]; }
    if (a.packed) {
        // raw-parameter gradients [xyz 3 | _scaling 3 | _rotation 4 | _opacity 1] through the activation Jacobians
        // (what autograd does through exp / normalize / sigmoid), times 1/N_v of the fused masked-L2 loss
        float scl = 1.0f;
        if (a.loss_sums) scl = (float)(1.0 / (n_view < 1.0 ? 1.0 : n_view));
        float* pk = a.packed + o * 11;
        const bool raw = a.flags & SKS_RAW_PARAMS;
        pk[0] = dmean[0] * scl; pk[1] = dmean[1] * scl; pk[2] = dmean[2] * scl;
#pragma unroll
        for (int k = 0; k < 3; k++) pk[3 + k] = dscale[k] * (raw ? sc[k] : 1.0f) * scl;
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) dot += q[k] * dq[k];
#pragma unroll
        for (int k = 0; k < 4; k++) pk[6 + k] = (raw ? (dq[k] - q[k] * dot) / qnorm : dq[k]) * scl;
        float oj = 1.0f;
        if (raw) {
            const float so = 1.0f / (1.0f + expf_fixed(-a.opac[idx]));
            oj = so * (1.0f - so);
        }
        pk[10] = dop * oj * scl;
    }
}

__global__ __launch_bounds__(256) void k_geom_bwd(GeomBwdArgs a, ViewTan vt)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y;
    float g[NACC];
    double pS, pN;
    geom_bwd_load(a, v, idx, g, pS, pN);
    double n_view = 1.0;
    if (a.loss_sums) {  // fused-loss mode (single block per view): image-wide S and N from the owners' partial sums
        __shared__ double s_l[2][4];
        pS = wave_sum_d(pS);
        pN = wave_sum_d(pN);
        if ((threadIdx.x & 63) == 0) { s_l[0][threadIdx.x >> 6] = pS; s_l[1][threadIdx.x >> 6] = pN; }
        __syncthreads();
        const double S = a.gt_totals[2 * v] + ((s_l[0][0] + s_l[0][1]) + (s_l[0][2] + s_l[0][3]));
        n_view = a.gt_totals[2 * v + 1] + ((s_l[1][0] + s_l[1][1]) + (s_l[1][2] + s_l[1][3]));
        if (threadIdx.x == 0) {
            a.loss_sums[2 * v] = S;
            a.loss_sums[2 * v + 1] = n_view;
        }
    }
    if (idx >= a.P) return;
    geom_bwd_finish(a, vt, v, idx, g, n_view);
}

// ------------------------------------------------------------------------------------------------------------
// Fused tail of one accumulation group of the sparse loop on ONE GPU (sks_loop_fused_step): everything that follows the
// compositing backward and precedes the next one is tiny (V*P <= a few hundred work items) and used to be three
// launches (k_geom_bwd, k_loop_adam, k_geom_fwd of the next group) whose fixed cost and the gaps between them were a
// third of the group.  One 256-thread workgroup runs them back to back:
//   A  geometry backward of every view: wavefront w takes views w, w+4, ...; lane = Gaussian (P <= 64), so a view's
//      loss sums are one wave reduction; writes the packed raw-parameter gradients;
//   B  the optimiser step (slots, mean over views, limb gradient, LR schedule, Adam) -- sks_loop_dev.h;
//   C  geometry forward of the UPDATED parameters, i.e. the geom / radii the next group's compositor reads.
// Same arithmetic in the same order as the separate kernels (bit-identical results).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_step_tail(GeomBwdArgs ga, ViewTan vt, sksloop::AdamArgs aa, int V, Geom g, int* radii)
{
    __shared__ float s_xyz[256 * 3];
    __shared__ float s_hyp[6];
    __shared__ double s_d[4];
    __shared__ int s_it[2];
    __shared__ float s_np[64 * 11];   // the updated parameters, handed to phase C through LDS (P <= 64)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the optimiser's scalar work (LR schedule, bias corrections: double transcendentals) needs no gradient: started
    // first, it is finished long before phase A's loads are back
    sksloop::adam_block_begin(aa, s_xyz, s_d, s_it);
    for (int v = wv; v < V; v += 4) {
        float gs[NACC];
        double pS, pN;
        geom_bwd_load(ga, v, lane, gs, pS, pN);
        pS = wave_sum_d(pS);
        pN = wave_sum_d(pN);
        const double S = ga.gt_totals[2 * v] + pS, n_view = ga.gt_totals[2 * v + 1] + pN;
        if (lane == 0) { ga.loss_sums[2 * v] = S; ga.loss_sums[2 * v + 1] = n_view; }
        if (lane < ga.P) geom_bwd_finish(ga, vt, v, lane, gs, n_view);
    }
    __syncthreads();   // every view's packed gradients are written (same workgroup, same L1)
    const int P = ga.P;
    const sksloop::AdamLdsParams mirror{ s_np, s_np + 3 * P, s_np + 6 * P, s_np + 10 * P };
    sksloop::adam_block_finish(aa, s_xyz, s_hyp, s_d, s_it, &mirror);
    __syncthreads();   // the parameters are updated
    for (int v = wv; v < V; v += 4)
        geom_fwd_one(P, ga.W, ga.H, vt, ga.vms, ga.pms, mirror.xyz, mirror.opacity, mirror.scaling, mirror.rotation, nullptr,
                     ga.smod, ga.flags, g, radii, v, lane, lane < P);
}

// ------------------------------------------------------------------------------------------------------------
// binned path: tile-centric replacement of InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs +
// identifyTileRanges (rasterizer_impl.cu:70-138, 280-320).  Per tile the reference's sorted order is
// (depth bits, Gaussian index) ascending (stable LSD sort, index-major emission), reproduced exactly.
// ------------------------------------------------------------------------------------------------------------
// BIN_SUB lanes share one Gaussian and stride over the tiles of its rect, so that the atomics of a rect are in flight
// together instead of one after the other (the scatter's atomics return a value: ~1-2 us each when serialised).
constexpr int BIN_SUB = 8;

__global__ void k_bin_count(int P, int NT, int gx, const uint4* __restrict__ rect, uint32_t* __restrict__ count)
{
    const int t = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    const int idx = t / BIN_SUB, sub = t % BIN_SUB;
    if (idx >= P) return;
    const uint4 r = rect[(size_t)v * P + idx];
    const int wt = (int)(r.z - r.x), n = wt * (int)(r.w - r.y);
    for (int i = sub; i < n; i += BIN_SUB) {
        const int yy = i / wt;
        atomicAdd(&count[(size_t)v * NT + (r.y + yy) * gx + r.x + (i - yy * wt)], 1u);
    }
}

// one 1024-thread workgroup per view: exclusive scan over tiles in tile-id order.  Thread t owns the `per`
// consecutive tiles starting at t * per (per a multiple of 4: 16-byte loads and stores).
__global__ __launch_bounds__(1024) void k_bin_scan(int NT, const uint32_t* __restrict__ count, uint32_t* __restrict__ cursor,
                                                    uint2* __restrict__ ranges, int* __restrict__ nrend, int V,
                                                    int* __restrict__ nrend_user)
{
    __shared__ uint32_t s_part[1024];
    const int v = blockIdx.x, tid = threadIdx.x;
    const uint32_t* cnt = count + (size_t)v * NT;
    const int per = (((NT + 1023) / 1024) + 3) & ~3;
    const int b = min(NT, tid * per), e = min(NT, b + per);
    const bool vec = (NT & 3) == 0;   // rows stay 16-byte aligned
    uint32_t sum = 0;
    if (vec) {
        for (int i = b; i < e; i += 4) {
            const uint4 c = *reinterpret_cast<const uint4*>(cnt + i);
            sum += c.x + c.y + c.z + c.w;
        }
    } else {
        for (int i = b; i < e; i++) sum += cnt[i];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t t = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    uint32_t run = tid ? s_part[tid - 1] : 0;
    uint32_t* cur = cursor + (size_t)v * NT;
    uint2* rng = ranges + (size_t)v * NT;
    if (vec) {
        for (int i = b; i < e; i += 4) {
            const uint4 c = *reinterpret_cast<const uint4*>(cnt + i);
            const uint32_t r0 = run, r1 = r0 + c.x, r2 = r1 + c.y, r3 = r2 + c.z;
            run = r3 + c.w;
            *reinterpret_cast<uint4*>(cur + i) = make_uint4(r0, r1, r2, r3);
            // (0, 0) for empty tiles, like the reference's zero-initialised ranges
            *reinterpret_cast<uint4*>(rng + i) = make_uint4(c.x ? r0 : 0u, c.x ? r1 : 0u, c.y ? r1 : 0u, c.y ? r2 : 0u);
            *reinterpret_cast<uint4*>(rng + i + 2) = make_uint4(c.z ? r2 : 0u, c.z ? r3 : 0u, c.w ? r3 : 0u, c.w ? run : 0u);
        }
    } else {
        for (int i = b; i < e; i++) {
            const uint32_t c = cnt[i];
            cur[i] = run;
            rng[i] = c ? make_uint2(run, run + c) : make_uint2(0, 0);
            run += c;
        }
    }
    if (tid == 1023) {
        nrend[v] = (int)s_part[1023];
        if (nrend_user) nrend_user[v] = (int)s_part[1023];
    }
}

__global__ void k_bin_scatter(int P, int NT, int gx, size_t cap, const uint4* __restrict__ rect,
                              const float4* __restrict__ xyd, uint32_t* __restrict__ cursor,
                              unsigned long long* __restrict__ keys, int* __restrict__ overflow)
{
    const int t = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    const int idx = t / BIN_SUB, sub = t % BIN_SUB;
    if (idx >= P) return;
    const size_t o = (size_t)v * P + idx;
    const uint4 r = rect[o];
    const unsigned long long key = ((unsigned long long)__float_as_uint(xyd[o].z) << 32) | (unsigned)idx;
    const int wt = (int)(r.z - r.x), n = wt * (int)(r.w - r.y);
    for (int i = sub; i < n; i += BIN_SUB) {
        const int yy = i / wt;
        const uint32_t slot = atomicAdd(&cursor[(size_t)v * NT + (r.y + yy) * gx + r.x + (i - yy * wt)], 1u);
        if (slot < cap) keys[(size_t)v * cap + slot] = key;
        else *overflow = 1;
    }
}

// per-tile ascending sort of the 64-bit keys.  One workgroup takes 4 consecutive tiles: a wavefront rank-sorts a list
// of up to 64 keys in registers (one key per lane; the rank of a key is the number of smaller ones -- keys are
// distinct, the low word is the Gaussian index); longer lists go through the whole workgroup's all-ascending
// bitonic network with virtual +inf padding (in LDS up to SORT_LDS keys, else in place).
constexpr int SORT_LDS = 2048;
__global__ __launch_bounds__(256) void k_bin_sort(int NT, size_t cap, const uint2* __restrict__ ranges,
                                                   unsigned long long* __restrict__ keys)
{
    __shared__ unsigned long long s[SORT_LDS];
    const int v = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t_w = blockIdx.x * 4 + wv;
    if (t_w < NT) {
        const uint2 r = ranges[(size_t)v * NT + t_w];
        const int n = (int)(r.y - r.x);
        if (r.y > r.x + 1 && r.y <= cap && n <= 64) {
            unsigned long long* g = keys + (size_t)v * cap + r.x;
            const unsigned long long key = lane < n ? g[lane] : ~0ull;
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, j);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), j);
                rank += ((((unsigned long long)hi) << 32) | lo) < key ? 1 : 0;
            }
            if (lane < n) g[rank] = key;
        }
    }
    for (int q = 0; q < 4; q++) {   // the long lists among this workgroup's 4 tiles (uniform control flow)
        const int t = blockIdx.x * 4 + q;
        if (t >= NT) break;
        const uint2 r = ranges[(size_t)v * NT + t];
        if (r.y > cap || r.y <= r.x + 64) continue;
        const int n = (int)(r.y - r.x);
        unsigned long long* g = keys + (size_t)v * cap + r.x;
        unsigned long long* a = g;
        const bool in_lds = n <= SORT_LDS;
        __syncthreads();
        if (in_lds) {
            for (int i = tid; i < n; i += 256) s[i] = g[i];
            a = s;
            __syncthreads();
        }
        int N = 1;
        while (N < n) N <<= 1;
        for (int k = 2; k <= N; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < N; i += 256) {
                    const int p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);  // flip on the first sub-step, then disperse
                    if (p > i && p < n) {
                        const unsigned long long x = a[i], y = a[p];
                        if (x > y) { a[i] = y; a[p] = x; }
                    }
                }
                __syncthreads();
            }
        }
        if (in_lds)
            for (int i = tid; i < n; i += 256) g[i] = s[i];
    }
}

__global__ void k_export_lists(int NT, size_t cap, const uint2* __restrict__ ranges, const unsigned long long* __restrict__ keys,
                               const int* __restrict__ nrend, uint32_t* __restrict__ point_list, uint32_t* __restrict__ out_ranges)
{
    const int v = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)NT) {
        const uint2 r = ranges[(size_t)v * NT + i];
        out_ranges[((size_t)v * NT + i) * 2] = r.x;
        out_ranges[((size_t)v * NT + i) * 2 + 1] = r.y;
    }
    if (i < cap) point_list[(size_t)v * cap + i] = i < (size_t)nrend[v] ? (uint32_t)keys[(size_t)v * cap + i] : 0u;
}

// stage one batch of a tile's sorted entries into LDS (forward.cu:335-343 / backward.cu:536-548).
// chan != nullptr: only the nc listed channels are staged, compacted to positions 0..nc-1 (rest zero).
template <int CG>
__device__ __forceinline__ void stage_batch(List<CG>& L, int cnt, const unsigned long long* __restrict__ keys, int P, int C,
                                            const float4* __restrict__ gco, const float4* __restrict__ gxyd,
                                            const float* __restrict__ features, const int* chan = nullptr, int nc = 0)
{
    const int tid = threadIdx.x;
    if (tid < cnt) {
        const int id = (int)(unsigned)keys[tid];
        const float4 xyd = gxyd[id];
        L.xy[tid] = make_float2(xyd.x, xyd.y);
        L.co[tid] = gco[id];
        L.invd[tid] = xyd.w;
        L.id[tid] = id;
        if (chan) {
#pragma unroll
            for (int j = 0; j < CG; j++) L.feat[tid * CG + j] = j < nc ? features[id * C + chan[j]] : 0.0f;
        } else {
#pragma unroll
            for (int ch = 0; ch < CG; ch++) L.feat[tid * CG + ch] = ch < C ? features[id * C + ch] : 0.0f;
        }
    }
}

struct BinView {
    const uint2* ranges;
    const unsigned long long* keys;
    size_t cap;
    int NT;
};

// cover rows of the binned path: bit (view, band, tile column) = "the tile's list is not empty" (same layout as the
// rows k_geom_fwd writes for the small path: word 0 = any, then one bit per tile column).  grid (gy, V), one wavefront.
__global__ __launch_bounds__(64) void k_bin_cover(int NT, int gx, int cw, size_t cap, const uint2* __restrict__ ranges,
                                                  uint32_t* __restrict__ cover)
{
    const int band = blockIdx.x, v = blockIdx.y, lane = threadIdx.x;
    uint32_t* row = cover + ((size_t)v * gridDim.x + band) * cw;
    unsigned long long any = 0ull;
    for (int w = 0; w * 64 < gx; w++) {
        const int tx = w * 64 + lane;
        bool ne = false;
        if (tx < gx) {
            const uint2 r = ranges[(size_t)v * NT + (size_t)band * gx + tx];
            ne = min((size_t)r.y, cap) > min((size_t)r.x, cap);
        }
        const unsigned long long m = __ballot(ne);
        any |= m;
        if (lane == 0) {
            if (1 + 2 * w < cw) row[1 + 2 * w] = (uint32_t)m;
            if (2 + 2 * w < cw) row[2 + 2 * w] = (uint32_t)(m >> 32);
        }
    }
    if (lane == 0) row[0] = any ? 1u : 0u;
}

// binned forward, "fill + sparse composite" like k_render_fwd_sparse: grid (fsplit + xc, gy, (C+1) * V).
//   * fill role (x < fsplit): fwd_fill_role over the cover rows of k_bin_cover -- empty tiles (the vast majority
//     of a skeleton scene) are zero-filled in long contiguous rows instead of 64-byte tile rows;
//   * composite role (the xc extra blocks of every row): block -> (view, tile); a non-empty tile is composited
//     thread-per-pixel over its sorted list exactly like forward.cu:278-401 and writes all C+1 planes.
template <int CG, int PPT, bool NT>
__global__ __launch_bounds__(256) void k_render_fwd_binned(FwdArgs a, BinView b, int gx, int gy, int fsplit, int pb,
                                                           const uint32_t* __restrict__ cover)
{
    __shared__ List<CG> L;
    const int tid = threadIdx.x;
    const int xq = blockIdx.x, band_id = blockIdx.y, zid = blockIdx.z;
    if (xq < fsplit) {
        fwd_fill_role<PPT, NT>(a, xq, band_id, zid, gy, pb, cover);
        return;
    }
    const int xc = gridDim.x - fsplit;
    const int cb = (zid * gy + band_id) * xc + (xq - fsplit);
    const int v = cb / b.NT;
    if (v >= (int)(gridDim.z / (a.C + 1))) return;
    const int tile = cb - v * b.NT;
    const int ty = tile / gx, tx = tile - ty * gx;
    const uint2 range = b.ranges[(size_t)v * b.NT + tile];
    const int total = (int)(min((size_t)range.y, b.cap) - min((size_t)range.x, b.cap));
    if (total == 0) return;  // a fill block zeroes this tile
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const size_t HW = (size_t)H * W;
    const size_t go = (size_t)v * P;
    const int x = tx * TILE + (tid & 15), y = ty * TILE + (tid >> 4);
    const bool inside = x < W && y < H;
    const unsigned long long* keys = b.keys + (size_t)v * b.cap;
    float T = 1.0f, inv = 0.0f, acc[CG];
#pragma unroll
    for (int ch = 0; ch < CG; ch++) acc[ch] = 0.0f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;
    for (int off = 0; off < total; off += LCAP) {
        if (__syncthreads_count(done) == 256) break;
        const int cnt = min(LCAP, total - off);
        stage_batch<CG>(L, cnt, keys + range.x + off, P, C, a.g.co + go, a.g.xyd + go, a.features);
        __syncthreads();
        composite_px<CG, false>(L, cnt, (float)x, (float)y, 0, T, acc, inv, contributor, last, done);
    }
    if (inside) {
        const size_t pix = (size_t)y * W + x;
        const bool do_clamp = a.flags & SKS_CLAMP01;
        float* outc = a.out_color + (size_t)v * C * HW + pix;
#pragma unroll
        for (int ch = 0; ch < CG; ch++)
            if (ch < C) outc[(size_t)ch * HW] = do_clamp ? clamp01(acc[ch]) : acc[ch];
        a.out_invdepth[(size_t)v * HW + pix] = inv;
        if (a.final_T) a.final_T[(size_t)v * HW + pix] = T;
        if (a.n_contrib) a.n_contrib[(size_t)v * HW + pix] = last;
    }
}

// binned backward: grid (Tx, Ty, V), thread = pixel (backward.cu:452-638).
// Only the channels some entry of the tile's list has a non-zero feature for can contribute to the per-Gaussian sums
// when neither dL/dfeatures nor a background term is wanted (see k_render_bwd_gather): they are found first
// (one light pass over the list) and processed BWD_NB at a time with the features staged compacted -- with one-hot
// skeleton features that is one group of 1-3 planes of dL/d(render) instead of 17-19.  Every per-Gaussian sum is
// linear in dL/dalpha, which is a sum over channels (+ the inverse-depth and background terms, kept in group 0 / split
// like the channels), so groups simply accumulate; T, alpha and the last contributor do not depend on the channel.
// The per-pixel register arrays have BWD_NB entries instead of C, which is what the occupancy of this kernel hinges on.
constexpr int BWD_NB = 8;

template <bool DFEAT>
__global__ __launch_bounds__(256) void k_render_bwd_binned(BwdArgs a, BinView b)
{
    constexpr int NB = BWD_NB;
    constexpr int NVL = NACC + (DFEAT ? NB : 0);
    __shared__ List<NB> L;
    __shared__ float s_acc[LCAP * NVL];
    __shared__ unsigned s_chm;
    __shared__ int s_chan[SKS_MAX_CHANNELS];
    __shared__ float s_full[LCAP * SKS_MAX_CHANNELS];   // full feature rows of a single-batch list
    const int v = blockIdx.z, tid = threadIdx.x;
    const int P = a.P, C = a.C, W = a.W, H = a.H;
    const size_t HW = (size_t)H * W;
    const size_t go = (size_t)v * P;
    const int gx = gridDim.x;
    const uint2 range = b.ranges[(size_t)v * b.NT + blockIdx.y * gx + blockIdx.x];
    const int total = (int)(min((size_t)range.y, b.cap) - min((size_t)range.x, b.cap));
    if (total == 0) return;
    const unsigned long long* keys = b.keys + (size_t)v * b.cap + range.x;
    const int x = blockIdx.x * TILE + (tid & 15), y = blockIdx.y * TILE + (tid >> 4);
    const bool inside = x < W && y < H;
    const size_t pix = inside ? (size_t)y * W + x : 0;
    const bool do_clamp = a.flags & SKS_CLAMP01;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    const int NVS = NACC + C;

    // pass 0: the active channels of this tile's list -> s_chan[0..nc).  A list that fits one LDS batch (the norm) is
    // fetched from global memory exactly once: records into L, full feature rows into s_full, from which every
    // channel group below is compacted; longer lists take a light extra pass over the features and are staged per
    // batch and group.
    const bool all_ch = DFEAT || a.bg != nullptr;
    const bool single = total <= LCAP;
    if (tid == 0) s_chm = 0u;
    __syncthreads();
    if (single) {
        if (tid < total) {
            const int id = (int)(unsigned)keys[tid];
            const float4 xyd = a.g.xyd[go + id];
            L.xy[tid] = make_float2(xyd.x, xyd.y);
            L.co[tid] = a.g.co[go + id];
            L.invd[tid] = xyd.w;
            L.id[tid] = id;
            const float* f = a.features + (size_t)id * C;
            unsigned m = 0u;
            for (int ch = 0; ch < C; ch++) {
                const float fv = f[ch];
                s_full[tid * C + ch] = fv;
                m |= fv != 0.0f ? (1u << ch) : 0u;
            }
            if (m && !all_ch) atomicOr(&s_chm, m);
        }
        __syncthreads();
    } else if (!all_ch) {
        unsigned m = 0u;
        for (int i = tid; i < total; i += 256) {
            const float* f = a.features + (size_t)(unsigned)keys[i] * C;
            for (int ch = 0; ch < C; ch++) m |= f[ch] != 0.0f ? (1u << ch) : 0u;
        }
        if (m) atomicOr(&s_chm, m);
        __syncthreads();
    }
    const unsigned chm = all_ch ? (C >= 32 ? 0xffffffffu : (1u << C) - 1u) : s_chm;
    const int nc = __popc(chm);
    if (tid < SKS_MAX_CHANNELS) {
        unsigned r = chm;
        for (int j = 0; j < tid && r; j++) r &= r - 1;
        s_chan[tid] = r ? __builtin_ctz(r) : 0;
    }
    __syncthreads();

    const int nb = (total + LCAP - 1) / LCAP;
    const int ngroups = nc > 0 ? (nc + NB - 1) / NB : 1;  // all-zero features: the inverse-depth terms still flow
    float T_final = 1.0f;
    int glast = -1;   // global (tile-list) index of the last accepted entry
    int staged = -1;  // (group, batch) currently in LDS
    for (int grp = 0; grp < ngroups; grp++) {
        const int* chan = s_chan + grp * NB;
        const int ncg = min(NB, nc - grp * NB);
        // pass 1: re-composite front to back: T_final and the last contributor (group 0), this group's colours (clamp)
        float col[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) col[j] = 0.0f;
        if (grp == 0 || do_clamp) {
            float T = 1.0f;
            int gl = -1;
            bool done = !inside;
            for (int off = 0; off < total; off += LCAP) {
                if (__syncthreads_count(done) == 256) break;
                const int cnt = min(LCAP, total - off);
                if (single) {
                    if (staged != grp) {   // (the group's features are still in place when pass 2 of it follows)
                        if (tid < cnt) {
#pragma unroll
                            for (int j = 0; j < NB; j++) L.feat[tid * NB + j] = j < ncg ? s_full[tid * C + chan[j]] : 0.0f;
                        }
                    }
                } else {
                    stage_batch<NB>(L, cnt, keys + off, P, C, a.g.co + go, a.g.xyd + go, a.features, chan, ncg);
                }
                staged = grp * nb + off / LCAP;
                __syncthreads();
                int klast = -1;
                bwd_prepass<NB, false>(L, cnt, (float)x, (float)y, 0, T, col, do_clamp, klast, done);
                if (klast >= 0) gl = off + klast;
            }
            __syncthreads();
            if (grp == 0) {
                T_final = T;
                glast = gl;
                if (__syncthreads_count(glast >= 0) == 0) return;   // nothing was accepted anywhere in the tile
            }
        }
        BwdPix<NB> s;
        if (glast >= 0) {
            bwd_load_pixel<NB>(s, a, v, pix, HW, col, do_clamp, T_final, chan, ncg);
            if (grp > 0) s.dLi = 0.0f;   // the inverse-depth terms belong to group 0
        }
        // pass 2: back to front over the batches
        for (int bi = nb - 1; bi >= 0; bi--) {
            const int off = bi * LCAP;
            const int cnt = min(LCAP, total - off);
            if (staged != grp * nb + bi) {
                if (single) {
                    if (tid < cnt) {
#pragma unroll
                        for (int j = 0; j < NB; j++) L.feat[tid * NB + j] = j < ncg ? s_full[tid * C + chan[j]] : 0.0f;
                    }
                } else {
                    stage_batch<NB>(L, cnt, keys + off, P, C, a.g.co + go, a.g.xyd + go, a.features, chan, ncg);
                }
                staged = grp * nb + bi;
            }
            for (int i = tid; i < cnt * NVL; i += 256) s_acc[i] = 0.0f;
            __syncthreads();
            bwd_sweep<NB, false, DFEAT>(L, cnt, glast - off, (float)x, (float)y, 0, ddelx_dx, ddely_dy, s, s_acc, ncg);
            __syncthreads();
            for (int i = tid; i < cnt * NVL; i += 256) {
                const int k = i / NVL, j = i - k * NVL;
                int slot = j;
                if (j >= NACC) {
                    if (j - NACC >= ncg) continue;
                    slot = NACC + chan[j - NACC];
                }
                const float val = s_acc[i];
                if (val != 0.0f) atomicAdd(&a.accum[((size_t)v * P + L.id[k]) * NVS + slot], val);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
inline int pick_cg(int C) { return C <= 4 ? 4 : C <= 16 ? 16 : C <= 20 ? 20 : 32; }

int check_common(int V, int P, int C, int W, int H)
{
    if (V < 1 || V > SKS_MAX_VIEWS) return fail(-1, "V=%d out of range [1,%d]", V, SKS_MAX_VIEWS);
    if (P < 0) return fail(-1, "P=%d negative", P);
    if (C < 1 || C > SKS_MAX_CHANNELS) return fail(-1, "C=%d out of range [1,%d]", C, SKS_MAX_CHANNELS);
    if (W < 1 || H < 1 || W > 4096 * TILE || H > 65535 * TILE) return fail(-1, "image %dx%d out of range", W, H);
    return 0;
}

// Fill-block geometry shared by the two forward launchers.  Row-aligned mode when the image width keeps >= 90% of
// the lanes of a 1024-pixel chunk busy AND a row is a whole number of 128-byte lines (1920, 2048 ... wide images;
// measured +2% at 1920, +12% on the 2048-wide stress config): blocks per band = chunks * row blocks (~8 per band),
// `pb` is returned NEGATIVE = -(rows per block).  Otherwise linear mode, pb passes of 4 KB per block: every pass is
// 32 whole lines whatever the width (at W = 1000 a row is 31.25 lines and the row-aligned mode loses 7% to the
// partial lines at both ends of every row).
inline void fill_geometry(const FwdArgs& a, bool have_cover, int& fsplit, int& pb)
{
    const int ppt = a.W % 4 == 0 ? 4 : 1;
    const int passes = (TILE * a.W + 256 * ppt - 1) / (256 * ppt);
    const int tune = (int)((a.flags >> 8) & 0xff);                    // tuning knob: passes (or rows) per fill block
    const int chunks = (a.W + 1023) / 1024;
    const bool rowmode = ((ppt == 4 && have_cover && a.W % 32 == 0 && (long long)a.W * 10 >= (long long)chunks * 1024 * 9) ||
                          (ppt == 4 && have_cover && (a.flags & SKS_FILL_ROWS))) && !(a.flags & SKS_FILL_LINEAR);
    if (rowmode) {
        int pbr = tune;
        if (pbr <= 0) {
            const int rblocks = chunks >= 8 ? 1 : 8 / chunks;
            pbr = (TILE + rblocks - 1) / rblocks;
        }
        if (pbr > TILE) pbr = TILE;
        fsplit = chunks * ((TILE + pbr - 1) / pbr);
        pb = -pbr;
        return;
    }
    pb = tune;
    if (pb <= 0) pb = passes / 8 > 2 ? (passes + 4) / 8 : 2;          // ~8 fill blocks per (plane, band) row measured best
    fsplit = (passes + pb - 1) / pb;                                  // fill blocks per (plane, band) row
}

template <int CG>
void launch_fwd_small(const FwdArgs& a, int V, int gy, hipStream_t st)
{
    const int ncomp = T_SLOTS * a.P * V;
    int fsplit, pb;
    fill_geometry(a, a.g.cover != nullptr, fsplit, pb);
    const int rows_zy = (a.C + 1) * V * gy;
    const int xc = (ncomp + rows_zy - 1) / rows_zy;                   // composite blocks appended to every row
    const int cap = (a.P + 15) & ~15;
    const size_t lds = DynList<CG>::bytes(cap, CG);
    dim3 grid(fsplit + xc, gy, (a.C + 1) * V);
    const bool nt = !(a.flags & SKS_NO_NT_STORES);
    if (a.W % 4 == 0) {
        if (nt) hipLaunchKernelGGL((k_render_fwd_sparse<CG, 4, true>), grid, dim3(256), lds, st, a, ncomp, gy, fsplit, pb, (const uint32_t*)a.g.cover);
        else hipLaunchKernelGGL((k_render_fwd_sparse<CG, 4, false>), grid, dim3(256), lds, st, a, ncomp, gy, fsplit, pb, (const uint32_t*)a.g.cover);
    } else {
        hipLaunchKernelGGL((k_render_fwd_sparse<CG, 1, false>), grid, dim3(256), lds, st, a, ncomp, gy, fsplit, pb, (const uint32_t*)a.g.cover);
    }
}

template <int CG>
void launch_fwd_binned(const FwdArgs& a, const BinView& bv, int V, int gx, int gy, const uint32_t* cover, hipStream_t st)
{
    int fsplit, pb;
    fill_geometry(a, true, fsplit, pb);
    const int xc = (gx + a.C) / (a.C + 1);                            // (view, tile) composite blocks spread over the rows
    dim3 grid(fsplit + xc, gy, (a.C + 1) * V);
    const bool nt = !(a.flags & SKS_NO_NT_STORES);
    if (a.W % 4 == 0) {
        if (nt) hipLaunchKernelGGL((k_render_fwd_binned<CG, 4, true>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
        else hipLaunchKernelGGL((k_render_fwd_binned<CG, 4, false>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
    } else {
        hipLaunchKernelGGL((k_render_fwd_binned<CG, 1, false>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
    }
}

template <int CG>
void launch_bwd_small(const BwdArgs& a, int V, int gy, bool dfeat, hipStream_t st)
{
    (void)gy;
    dim3 grid(BWD_SPLITS, a.P, V);
    if (a.P <= 64 && !(a.flags & (1u << 20))) {  // wave-resident variant (bit 20: force the LDS variant, tests)
        if (dfeat) hipLaunchKernelGGL((k_render_bwd_wave<CG, true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_render_bwd_wave<CG, false, false>), grid, dim3(256), 0, st, a);
        return;
    }
    const size_t lds = GatherLds<CG>::bytes((a.P + 15) & ~15, CG, a.C);
    if (lds > 48 * 1024) {  // gfx950 has 160 KB of LDS per CU; raise the per-kernel dynamic limit when P is large
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_bwd_gather<CG, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_bwd_gather<CG, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (dfeat) hipLaunchKernelGGL((k_render_bwd_gather<CG, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_render_bwd_gather<CG, false>), grid, dim3(256), lds, st, a);
}

}  // namespace

extern "C" {

const char* sks_last_error(void) { return g_err; }
void sks_set_error_(const char* msg)  // used by the other translation units of the library
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
}
int sks_version(void) { return 1; }

int sks_scratch_bytes(int V, int P, int C, int W, int H, size_t bin_capacity, size_t* geom, size_t* binning, size_t* accum)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    const int NT = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    if (geom) *geom = geom_bytes(V, P > 0 ? P : 1, W, H);
    if (binning) *binning = bin_bytes(V, NT, bin_capacity);
    if (accum) *accum = (size_t)V * (P > 0 ? P : 1) * BWD_SPLITS * (NACC + C) * sizeof(float);
    return 0;
}

int sks_forward(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                const float* tanfovx, const float* tanfovy, const float* means3D, const float* features,
                const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                float scale_modifier, unsigned flags, float* out_color, float* out_invdepth, int* radii, void* geom,
                void* binning, size_t bin_capacity, int* num_rendered_dev, float* final_T, uint32_t* n_contrib,
                void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!out_color || !out_invdepth) return fail(-2, "out_color / out_invdepth must be provided");
    const size_t HW = (size_t)H * W;
    if (P == 0) {  // rasterize_points.cu:88: nothing rendered, outputs stay zero
        HIP_TRY(hipMemsetAsync(out_color, 0, (size_t)V * C * HW * 4, st));
        HIP_TRY(hipMemsetAsync(out_invdepth, 0, (size_t)V * HW * 4, st));
        if (final_T) return fail(-2, "final_T not available for P == 0");
        return 0;
    }
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii || !geom)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    for (int v = 0; v < V; v++) { vt.x[v] = tanfovx[v]; vt.y[v] = tanfovy[v]; }
    Geom g = geom_from(geom, V, P, W, H);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, NT = gx * gy;

    hipLaunchKernelGGL(k_geom_fwd, dim3((P + 255) / 256, V), dim3(256), 0, st, P, W, H, vt, viewmatrix, projmatrix,
                       means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier, flags, g, radii);
    STAGE_CHECK("geometry");

    FwdArgs a{ P, C, W, H, flags, g, features, out_color, out_invdepth, final_T, n_contrib };
    const int cg = pick_cg(C);
    const bool small = P <= SKS_SMALL_P && !(flags & SKS_FORCE_BINNED);
    if (small) {
        {
            ProfScope prof(0, st);
            switch (cg) {
                case 4: launch_fwd_small<4>(a, V, gy, st); break;
                case 16: launch_fwd_small<16>(a, V, gy, st); break;
                case 20: launch_fwd_small<20>(a, V, gy, st); break;
                default: launch_fwd_small<32>(a, V, gy, st); break;
            }
        }
        STAGE_CHECK("render(small)");
        return 0;
    }
    if (!binning) return fail(-2, "binned path needs a binning buffer");
    Bin b = bin_from(binning, V, NT, bin_capacity);
    HIP_TRY(hipMemsetAsync(b.count, 0, (size_t)V * NT * 4, st));
    HIP_TRY(hipMemsetAsync(b.nrend + V, 0, 4, st));
    hipLaunchKernelGGL(k_bin_count, dim3((P * BIN_SUB + 255) / 256, V), dim3(256), 0, st, P, NT, gx, g.rect, b.count);
    hipLaunchKernelGGL(k_bin_scan, dim3(V), dim3(1024), 0, st, NT, b.count, b.cursor, b.ranges, b.nrend, V, num_rendered_dev);
    hipLaunchKernelGGL(k_bin_scatter, dim3((P * BIN_SUB + 255) / 256, V), dim3(256), 0, st, P, NT, gx, bin_capacity, g.rect, g.xyd,
                       b.cursor, b.keys, b.nrend + V);
    hipLaunchKernelGGL(k_bin_sort, dim3((NT + 3) / 4, V), dim3(256), 0, st, NT, bin_capacity, b.ranges, b.keys);
    STAGE_CHECK("binning");
    BinView bv{ b.ranges, b.keys, bin_capacity, NT };
    uint32_t* cover = geom_cover_ptr(geom, V, P);
    const int cw = cover_cw(W);
    hipLaunchKernelGGL(k_bin_cover, dim3(gy, V), dim3(64), 0, st, NT, gx, cw, bin_capacity, b.ranges, cover);
    {
        ProfScope prof(0, st);
        switch (cg) {
            case 4: launch_fwd_binned<4>(a, bv, V, gx, gy, cover, st); break;
            case 16: launch_fwd_binned<16>(a, bv, V, gx, gy, cover, st); break;
            case 20: launch_fwd_binned<20>(a, bv, V, gx, gy, cover, st); break;
            default: launch_fwd_binned<32>(a, bv, V, gx, gy, cover, st); break;
        }
    }
    STAGE_CHECK("render(binned)");
    return 0;
}

int sks_backward(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                 const float* tanfovx, const float* tanfovy, const float* bg, const float* means3D,
                 const float* features, const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, float scale_modifier, unsigned flags, const int* radii, const void* geom,
                 const void* binning, size_t bin_capacity, const float* dL_dout_color, const float* dL_dout_invdepth,
                 void* accum, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dscales,
                 float* dL_drotations, float* dL_dcov3D, float* dL_dfeatures, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii ||
        !geom || !dL_dout_color || !accum || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    for (int v = 0; v < V; v++) { vt.x[v] = tanfovx[v]; vt.y[v] = tanfovy[v]; }
    Geom g = geom_from(const_cast<void*>(geom), V, P, W, H);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, NT = gx * gy;
    BwdArgs a{ P, C, W, H, flags, g, features, bg, dL_dout_color, dL_dout_invdepth, (float*)accum, nullptr, nullptr };
    const int cg = pick_cg(C);
    const bool dfeat = dL_dfeatures != nullptr;
    const bool small = P <= SKS_SMALL_P && !(flags & SKS_FORCE_BINNED);
    if (small) {
        ProfScope prof(1, st);
        switch (cg) {
            case 4: launch_bwd_small<4>(a, V, gy, dfeat, st); break;
            case 16: launch_bwd_small<16>(a, V, gy, dfeat, st); break;
            case 20: launch_bwd_small<20>(a, V, gy, dfeat, st); break;
            default: launch_bwd_small<32>(a, V, gy, dfeat, st); break;
        }
        STAGE_CHECK("render-backward(small)");
    } else {
        if (!binning) return fail(-2, "binned path needs the forward's binning buffer");
        HIP_TRY(hipMemsetAsync(accum, 0, (size_t)V * P * (NACC + C) * sizeof(float), st));  // atomics target
        Bin b = bin_from(const_cast<void*>(binning), V, NT, bin_capacity);
        BinView bv{ b.ranges, b.keys, bin_capacity, NT };
        dim3 grid(gx, gy, V);
        ProfScope prof(1, st);
        if (dfeat) hipLaunchKernelGGL((k_render_bwd_binned<true>), grid, dim3(256), 0, st, a, bv);
        else hipLaunchKernelGGL((k_render_bwd_binned<false>), grid, dim3(256), 0, st, a, bv);
        STAGE_CHECK("render-backward(binned)");
    }
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, means3D, opacities, scales, rotations, cov3D_precomp,
                    scale_modifier, radii, (const float*)accum, small ? BWD_SPLITS : 1, nullptr, nullptr, nullptr, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dscales,
                    dL_drotations, dL_dcov3D, dL_dfeatures };
    hipLaunchKernelGGL(k_geom_bwd, dim3((P + 255) / 256, V), dim3(256), 0, st, ga, vt);
    STAGE_CHECK("geometry-backward");
    return 0;
}

int sks_gt_tile_stats(int V, int C, int W, int H, const float* gt, float* tile_S, float* tile_N, double* totals, void* stream)
{
    if (int rc = check_common(V, 1, C, W, H)) return rc;
    if (!gt || !totals) return fail(-2, "gt_tile_stats: missing pointer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(totals, 0, (size_t)V * 2 * sizeof(double), st));
    dim3 grid((H + TILE - 1) / TILE, C, V);
    hipLaunchKernelGGL(k_gt_tile_stats, grid, dim3(256), 0, st, C, W, H, gt, tile_S, tile_N, totals);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_backward_fused_loss(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                            const float* tanfovx, const float* tanfovy, const float* bg, const float* means3D,
                            const float* features, const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, float scale_modifier, unsigned flags, const int* radii,
                            const void* geom, const float* gt, const float* tile_S, const float* tile_N,
                            const double* gt_totals, void* accum, float* dL_dmeans3D, float* dL_dmeans2D,
                            float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            double* loss_sums, float* packed_raw_grads, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1 || P > 64) return fail(-1, "fused-loss backward needs 1 <= P <= 64 (got %d)", P);
    hipStream_t st = (hipStream_t)stream;
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii || !geom || !gt ||
        !gt_totals || !accum || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity || !loss_sums)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    for (int v = 0; v < V; v++) { vt.x[v] = tanfovx[v]; vt.y[v] = tanfovy[v]; }
    Geom g = geom_from(const_cast<void*>(geom), V, P, W, H);
    BwdArgs a{ P, C, W, H, flags | SKS_CLAMP01, g, features, bg, gt, nullptr, (float*)accum, tile_S, tile_N };
    dim3 grid(BWD_SPLITS, P, V);
    {
        ProfScope prof(1, st);
        switch (pick_cg(C)) {
            case 4: hipLaunchKernelGGL((k_render_bwd_wave<4, false, true>), grid, dim3(256), 0, st, a); break;
            case 16: hipLaunchKernelGGL((k_render_bwd_wave<16, false, true>), grid, dim3(256), 0, st, a); break;
            case 20: hipLaunchKernelGGL((k_render_bwd_wave<20, false, true>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((k_render_bwd_wave<32, false, true>), grid, dim3(256), 0, st, a); break;
        }
    }
    STAGE_CHECK("fused loss + render-backward");
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, means3D, opacities, scales, rotations, cov3D_precomp,
                    scale_modifier, radii, (const float*)accum, BWD_SPLITS, gt_totals, loss_sums, packed_raw_grads, dL_dmeans3D, dL_dmeans2D,
                    dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, nullptr };
    hipLaunchKernelGGL(k_geom_bwd, dim3((P + 255) / 256, V), dim3(256), 0, st, ga, vt);
    STAGE_CHECK("geometry-backward");
    return 0;
}

int sks_geometry(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix, const float* tanfovx,
                 const float* tanfovy, const float* means3D, const float* opacities, const float* scales,
                 const float* rotations, const float* cov3D_precomp, float scale_modifier, unsigned flags, int* radii,
                 void* geom, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1) return fail(-1, "P must be positive");
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !opacities || !radii || !geom)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    hipStream_t st = (hipStream_t)stream;
    ViewTan vt;
    for (int v = 0; v < V; v++) { vt.x[v] = tanfovx[v]; vt.y[v] = tanfovy[v]; }
    Geom g = geom_from(geom, V, P, W, H);
    hipLaunchKernelGGL(k_geom_fwd, dim3((P + 255) / 256, V), dim3(256), 0, st, P, W, H, vt, viewmatrix, projmatrix,
                       means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier, flags, g, radii);
    STAGE_CHECK("geometry");
    return 0;
}

int sks_loop_fused_step(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                        const float* tanfovx, const float* tanfovy, const float* features, float scale_modifier,
                        unsigned flags, int* radii, void* geom, const float* gt, const double* gt_totals, void* accum,
                        double* loss_sums, float* packed, float* slots, unsigned long long group_mask, int last_view,
                        float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                        int* counters, int acc_steps, const double* lr_sched, const double* lrs, const double* adam,
                        float lambda_consistency, const int* limb, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1 || P > 64) return fail(-1, "fused step needs 1 <= P <= 64 (got %d)", P);
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !features || !radii || !geom || !gt || !gt_totals || !accum ||
        !loss_sums || !packed)
        return fail(-2, "missing required pointer");
    hipStream_t st = (hipStream_t)stream;
    ViewTan vt;
    for (int v = 0; v < V; v++) { vt.x[v] = tanfovx[v]; vt.y[v] = tanfovy[v]; }
    flags |= SKS_RAW_PARAMS | SKS_CLAMP01;
    sksloop::AdamArgs aa;
    if (const char* err = sksloop::fill_adam_args(aa, V, P, packed, slots, group_mask, last_view, xyz, scaling, rotation, opacity,
                                                  exp_avg, exp_avg_sq, counters, acc_steps, lr_sched, lrs, adam,
                                                  lambda_consistency, limb))
        return fail(-2, "%s", err);
    Geom g = geom_from(geom, V, P, W, H);
    g.cover = nullptr;   // no forward render on this path
    BwdArgs a{ P, C, W, H, flags, g, features, nullptr, gt, nullptr, (float*)accum, nullptr, nullptr };
    dim3 grid(BWD_SPLITS, P, V);
    {
        ProfScope prof(1, st);
        switch (pick_cg(C)) {
            case 4: hipLaunchKernelGGL((k_render_bwd_wave<4, false, true>), grid, dim3(256), 0, st, a); break;
            case 16: hipLaunchKernelGGL((k_render_bwd_wave<16, false, true>), grid, dim3(256), 0, st, a); break;
            case 20: hipLaunchKernelGGL((k_render_bwd_wave<20, false, true>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((k_render_bwd_wave<32, false, true>), grid, dim3(256), 0, st, a); break;
        }
    }
    STAGE_CHECK("render-backward(fused loss)");
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, xyz, opacity, scaling, rotation, nullptr, scale_modifier, radii,
                    (const float*)accum, BWD_SPLITS, gt_totals, loss_sums, packed, nullptr, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr };
    hipLaunchKernelGGL(k_step_tail, dim3(1), dim3(256), 0, st, ga, vt, aa, V, g, radii);
    STAGE_CHECK("step tail");
    return 0;
}

int sks_prof_enable(int on)
{
    g_prof_on = on != 0;
    return 0;
}

int sks_prof_read(int kind, double* total_ms, long long* launches)
{
    if (kind < 0 || kind > 1 || !total_ms || !launches) return fail(-2, "bad profile query");
    ProfKind& p = g_prof[kind];
    double tot = 0;
    for (int i = 0; i < p.n; i++) {
        HIP_TRY(hipEventSynchronize(p.e[i]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.b[i], p.e[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.n;
    p.n = 0;
    return 0;
}

int sks_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    (void)projmatrix;  // in_frustum only uses the view-space depth (auxiliary.h:166)
    if (P < 0) return fail(-1, "P negative");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(-2, "missing required pointer");
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, means3D, viewmatrix, present);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_export_lists(int V, int W, int H, const void* binning, size_t bin_capacity, uint32_t* point_list,
                     uint32_t* ranges, void* stream)
{
    if (!binning || !point_list || !ranges) return fail(-2, "missing required pointer");
    const int NT = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    Bin b = bin_from(const_cast<void*>(binning), V, NT, bin_capacity);
    const size_t m = bin_capacity > (size_t)NT ? bin_capacity : (size_t)NT;
    hipLaunchKernelGGL(k_export_lists, dim3((unsigned)((m + 255) / 256), V), dim3(256), 0, (hipStream_t)stream, NT,
                       bin_capacity, b.ranges, b.keys, b.nrend, point_list, ranges);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
