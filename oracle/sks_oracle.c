/*
 * sks_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or imported by the product).
 *
 * Scalar CPU restatement of the reference's skeletal-Gaussian rasterizer hot path
 * (laurabragagnolo/SkelSplat, submodules/diff-gaussian-rasterization-{h36m,panoptic,op}).
 * "DGR/" below abbreviates submodules/diff-gaussian-rasterization-h36m/.
 *
 * PARITY STATUS: the reference ships NO golden vectors / tests for the rasterizer and its CUDA
 * sources cannot be built in this image (no nvcc, CUB, cuda_runtime.h), so this restatement is
 * "parity unpinned" against reference-produced numbers.  It is cross-pinned instead by
 * (i) an independent differentiable PyTorch restatement + autograd (oracle/torch_ref.py) and
 * (ii) the reference's own Python (losses, camera matrices, LR schedule, SSIM) imported in the
 * build container to produce tests/golden/ fixtures.
 *
 * Floating-point contract (shared by the HIP kernels so that integer artefacts are bit-exact):
 *   - every expression is evaluated in fp32 in the reference's written order, WITHOUT fused
 *     multiply-add contraction (compile with -ffp-contract=off);
 *   - division and sqrt are IEEE correctly rounded;
 *   - ndc2Pix is evaluated in double exactly as written in DGR/cuda_rasterizer/auxiliary.h:40-43;
 *   - exp() of the compositor (forward.cu:364, backward.cu:568) is a fixed sequence of IEEE ops
 *     (orc_expf below, <=1 ulp like CUDA's expf) so CPU and GPU agree bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:16 */
#define BLOCK_Y 16 /* DGR/cuda_rasterizer/config.h:17 */

/* ------------------------------------------------------------------------------------------ */
/* small fp32 helpers                                                                          */
/* ------------------------------------------------------------------------------------------ */
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

/* exp(x): fixed IEEE op sequence (Cody-Waite reduction + degree-7 Horner with fmaf). */
float orc_expf(float x)
{
    if (x < -80.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float t = x * 1.44269504088896341f;
    float k = rintf(t);
    float r = fmaf(k, -0.693145751953125f, x);
    r = fmaf(k, -1.42860682030941723e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = fmaf(p, r, 1.38888888888888894e-3f);
    p = fmaf(p, r, 8.33333333333333322e-3f);
    p = fmaf(p, r, 4.16666666666666644e-2f);
    p = fmaf(p, r, 1.66666666666666657e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t ki = (int32_t)k;
    union { float f; int32_t i; } u;
    u.f = p;
    u.i += ki * (1 << 23);
    return u.f;
}

/* glm-style column-major 3x3: m[c][r]; product evaluated left-to-right like glm's operator*. */
typedef struct { float m[3][3]; } mat3;

static mat3 mat3_mul(const mat3* a, const mat3* b)
{
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a->m[0][r] * b->m[c][0] + a->m[1][r] * b->m[c][1] + a->m[2][r] * b->m[c][2];
    return o;
}
static mat3 mat3_T(const mat3* a)
{
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a->m[r][c];
    return o;
}

/* DGR/cuda_rasterizer/auxiliary.h:40-43 (double arithmetic, then narrowed) */
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* DGR/cuda_rasterizer/auxiliary.h:45-55 */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t rmin[2], uint32_t rmax[2])
{
    rmin[0] = (uint32_t)imin_(gx, imax_(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin_(gy, imax_(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin_(gx, imax_(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin_(gy, imax_(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* DGR/cuda_rasterizer/auxiliary.h:70-89 */
static void transformPoint4x3(const float p[3], const float* M, float o[3])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
static void transformPoint4x4(const float p[3], const float* M, float o[4])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

/* DGR/cuda_rasterizer/forward.cu:114-150 (quaternion NOT normalised, :123) */
static void computeCov3D(const float s[3], float mod, const float q[4], float cov3D[6], mat3* M_out)
{
    mat3 S;
    memset(&S, 0, sizeof(S));
    S.m[0][0] = mod * s[0];
    S.m[1][1] = mod * s[1];
    S.m[2][2] = mod * s[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    mat3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    mat3 M = mat3_mul(&S, &R);
    mat3 Mt = mat3_T(&M);
    mat3 Sigma = mat3_mul(&Mt, &M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
    if (M_out) *M_out = M;
}

/* shared by forward.cu:74-109 and backward.cu:173-201: t (clamped), J, W, T, cov2D */
typedef struct {
    float t[3]; float txtz, tytz, limx, limy;
    mat3 J, W, T, Vrk, cov;
} Cov2DCtx;

static void cov2d_ctx(const float mean[3], float fx, float fy, float tanfovx, float tanfovy,
                      const float* cov3D, const float* V, Cov2DCtx* c)
{
    transformPoint4x3(mean, V, c->t);
    c->limx = 1.3f * tanfovx;
    c->limy = 1.3f * tanfovy;
    c->txtz = c->t[0] / c->t[2];
    c->tytz = c->t[1] / c->t[2];
    c->t[0] = fminf_(c->limx, fmaxf_(-c->limx, c->txtz)) * c->t[2];
    c->t[1] = fminf_(c->limy, fmaxf_(-c->limy, c->tytz)) * c->t[2];
    const float* t = c->t;
    memset(&c->J, 0, sizeof(mat3));
    c->J.m[0][0] = fx / t[2]; c->J.m[0][1] = 0.0f; c->J.m[0][2] = -(fx * t[0]) / (t[2] * t[2]);
    c->J.m[1][0] = 0.0f; c->J.m[1][1] = fy / t[2]; c->J.m[1][2] = -(fy * t[1]) / (t[2] * t[2]);
    c->W.m[0][0] = V[0]; c->W.m[0][1] = V[4]; c->W.m[0][2] = V[8];
    c->W.m[1][0] = V[1]; c->W.m[1][1] = V[5]; c->W.m[1][2] = V[9];
    c->W.m[2][0] = V[2]; c->W.m[2][1] = V[6]; c->W.m[2][2] = V[10];
    c->T = mat3_mul(&c->W, &c->J);
    c->Vrk.m[0][0] = cov3D[0]; c->Vrk.m[0][1] = cov3D[1]; c->Vrk.m[0][2] = cov3D[2];
    c->Vrk.m[1][0] = cov3D[1]; c->Vrk.m[1][1] = cov3D[3]; c->Vrk.m[1][2] = cov3D[4];
    c->Vrk.m[2][0] = cov3D[2]; c->Vrk.m[2][1] = cov3D[4]; c->Vrk.m[2][2] = cov3D[5];
    mat3 Tt = mat3_T(&c->T), Vt = mat3_T(&c->Vrk);
    mat3 a = mat3_mul(&Tt, &Vt);
    c->cov = mat3_mul(&a, &c->T);
}

/* ------------------------------------------------------------------------------------------ */
/* forward: preprocessCUDA  (DGR/cuda_rasterizer/forward.cu:153-273)                           */
/* ------------------------------------------------------------------------------------------ */
void orc_preprocess(int P, const float* means3D, const float* scales, float scale_modifier,
                    const float* rotations, const float* opacities, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, int W, int H,
                    float tan_fovx, float tan_fovy, int antialiasing,
                    int* radii, float* xy, float* depths, float* cov3Ds, float* conic_opacity,
                    uint32_t* tiles_touched)
{
    /* rasterizer_impl.cu:224-225,236 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;

    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float* p_orig = means3D + 3 * idx;
        /* in_frustum, auxiliary.h:151-176 */
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue;

        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };

        const float* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + idx * 6;
        else {
            computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + idx * 6, NULL);
            cov3D = cov3Ds + idx * 6;
        }
        Cov2DCtx c;
        cov2d_ctx(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c);
        float cov_x = c.cov.m[0][0], cov_y = c.cov.m[0][1], cov_z = c.cov.m[1][1];

        const float h_var = 0.3f;
        const float det_cov = cov_x * cov_z - cov_y * cov_y;
        cov_x += h_var;
        cov_z += h_var;
        const float det_cov_plus_h_cov = cov_x * cov_z - cov_y * cov_y;
        float h_convolution_scaling = 1.0f;
        if (antialiasing) h_convolution_scaling = sqrtf(fmaxf_(0.000025f, det_cov / det_cov_plus_h_cov));

        const float det = det_cov_plus_h_cov;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { cov_z * det_inv, -cov_y * det_inv, cov_x * det_inv };

        float mid = 0.5f * (cov_x + cov_z);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj[0], W), piy = ndc2Pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        getRect(pix, piy, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        xy[2 * idx] = pix;
        xy[2 * idx + 1] = piy;
        float opacity = opacities[idx];
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacity * h_convolution_scaling;
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
}

/* markVisible / checkFrustum (rasterizer_impl.cu:54-66) */
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
    for (int i = 0; i < P; i++) {
        float pv[3];
        transformPoint4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* binning: InclusiveSum + duplicateWithKeys + stable SortPairs + identifyTileRanges            */
/* (rasterizer_impl.cu:70-138, 280-320)                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; uint32_t val; uint32_t seq; } KV;
static int kv_cmp(const void* a, const void* b)
{
    const KV* x = (const KV*)a; const KV* y = (const KV*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable */
}

/* returns R; outputs sized by caller: point_offsets[P], keys/point_list[cap], ranges[gx*gy*2] */
int orc_bin(int P, int W, int H, const float* xy, const float* depths, const int* radii,
            const uint32_t* tiles_touched, uint32_t* point_offsets, uint64_t* keys_sorted,
            uint32_t* point_list, uint32_t* ranges, int cap)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += tiles_touched[i]; point_offsets[i] = run; }
    int R = P ? (int)point_offsets[P - 1] : 0;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    if (R > cap) return -R;
    KV* kv = (KV*)malloc(sizeof(KV) * (size_t)(R ? R : 1));
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(xy[2 * idx], xy[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            for (int y = (int)rmin[1]; y < (int)rmax[1]; y++)
                for (int x = (int)rmin[0]; x < (int)rmax[0]; x++) {
                    uint64_t key = (uint32_t)(y * gx + x);
                    key <<= 32;
                    uint32_t db;
                    memcpy(&db, &depths[idx], 4);
                    key |= db;
                    kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off;
                    off++;
                }
        }
    }
    qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
    for (int i = 0; i < R; i++) { keys_sorted[i] = kv[i].key; point_list[i] = kv[i].val; }
    for (int idx = 0; idx < R; idx++) { /* identifyTileRanges */
        uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) { ranges[2 * prevtile + 1] = idx; ranges[2 * currtile] = idx; }
        }
        if (idx == R - 1) ranges[2 * currtile + 1] = R;
    }
    free(kv);
    return R;
}

/* ------------------------------------------------------------------------------------------ */
/* forward compositor: renderCUDA  (forward.cu:278-401), one pixel at a time                   */
/* ------------------------------------------------------------------------------------------ */
void orc_render_fwd(int W, int H, int C, const uint32_t* ranges, const uint32_t* point_list,
                    const float* xy, const float* features, const float* conic_opacity,
                    const float* depths, float* out_color, float* final_T, uint32_t* n_contrib,
                    float* invdepth)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    float* Cacc = (float*)malloc(sizeof(float) * (size_t)C);
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const uint32_t pix_id = (uint32_t)(W * py + px);
            const float pixf_x = (float)px, pixf_y = (float)py;
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float expected_invdepth = 0.0f;
            for (int ch = 0; ch < C; ch++) Cacc[ch] = 0.0f;
            for (uint32_t e = r0; e < r1; e++) {
                contributor++;
                const uint32_t id = point_list[e];
                float dx = xy[2 * id] - pixf_x, dy = xy[2 * id + 1] - pixf_y;
                const float* co = conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf_(0.99f, co[3] * orc_expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break; /* done = true */
                for (int ch = 0; ch < C; ch++) Cacc[ch] += features[id * C + ch] * alpha * T;
                expected_invdepth += (1 / depths[id]) * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            if (final_T) final_T[pix_id] = T;
            if (n_contrib) n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix_id] = Cacc[ch];
            if (invdepth) invdepth[pix_id] = expected_invdepth;
        }
    free(Cacc);
}

/* ------------------------------------------------------------------------------------------ */
/* backward compositor: renderCUDA (backward.cu:452-638).  The reference accumulates with       */
/* fp32 atomicAdd in arbitrary order; here per-Gaussian sums are accumulated in double so the   */
/* oracle is the order-free "exact" value the fp32 sums approximate.                            */
/* bg: C floats (the reference reads C floats from a 3-float tensor, backward.cu:613-614 --     */
/* an out-of-bounds read; callers pad with zeros).                                              */
/* ------------------------------------------------------------------------------------------ */
void orc_render_bwd(int P, int W, int H, int C, const uint32_t* ranges, const uint32_t* point_list,
                    const float* bg, const float* xy, const float* conic_opacity, const float* colors,
                    const float* depths, const float* final_Ts, const uint32_t* n_contrib,
                    const float* dL_dpixels, const float* dL_invdepths,
                    float* dL_dmean2D /*P*3*/, float* dL_dconic2D /*P*4*/, float* dL_dopacity /*P*/,
                    float* dL_dcolors /*P*C*/, float* dL_dinvdepths /*P or NULL*/,
                    double* abs_sums /* NULL, or P*(3+4+1+C+1) doubles [mean2D 3P | conic 4P | opacity P | colors P*C |
                    invdepth P]: beside every sum, the sum of the ABSOLUTE values of what goes into it, every difference
                    inside a pixel's term taken as a sum too (c - accum_rec -> |c| + |accum_rec| ...): the scale on which
                    two fp32 evaluations of the same sum may differ by rounding and order alone (tests/fuzz_cases.py) */)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    const size_t nacc = (size_t)P * (size_t)(3 + 4 + 1 + C + 1);
    double* acc = (double*)calloc(nacc, sizeof(double));
    double* a_m2d = acc, *a_con = a_m2d + 3 * (size_t)P, *a_op = a_con + 4 * (size_t)P,
           *a_col = a_op + P, *a_inv = a_col + (size_t)P * C;
    double* b_m2d = NULL, *b_con = NULL, *b_op = NULL, *b_col = NULL, *b_inv = NULL;
    float* abs_rec = NULL;    /* the blend of |colour| behind the entry in hand (accum_rec's counterpart) */
    if (abs_sums) {
        memset(abs_sums, 0, nacc * sizeof(double));
        b_m2d = abs_sums; b_con = b_m2d + 3 * (size_t)P; b_op = b_con + 4 * (size_t)P; b_col = b_op + P;
        b_inv = b_col + (size_t)P * C;
        abs_rec = (float*)malloc(sizeof(float) * (C + 1));
    }
    float* accum_rec = (float*)malloc(sizeof(float) * C);
    float* last_color = (float*)malloc(sizeof(float) * C);
    float* dL_dpixel = (float*)malloc(sizeof(float) * C);
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            if (r1 == r0) continue;
            const uint32_t pix_id = (uint32_t)(W * py + px);
            const float pixf_x = (float)px, pixf_y = (float)py;
            const float T_final = final_Ts[pix_id];
            float T = T_final;
            uint32_t contributor = r1 - r0;
            const uint32_t last_contributor = n_contrib[pix_id];
            float accum_invdepth_rec = 0, dL_invdepth = 0, last_alpha = 0, last_invdepth = 0;
            for (int i = 0; i < C; i++) {
                accum_rec[i] = 0; last_color[i] = 0;
                dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
            }
            if (dL_invdepths) dL_invdepth = dL_invdepths[pix_id];
            if (abs_sums) for (int i = 0; i <= C; i++) abs_rec[i] = 0;
            float abs_last_invdepth = 0;

            for (uint32_t e = r1; e-- > r0;) {
                contributor--;
                if (contributor >= last_contributor) continue;
                const uint32_t gid = point_list[e];
                const float dx = xy[2 * gid] - pixf_x, dy = xy[2 * gid + 1] - pixf_y;
                const float* co = conic_opacity + 4 * gid;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = orc_expf(power);
                const float alpha = fminf_(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;

                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                double abs_dalpha = 0.0;
                if (abs_sums) {   /* (before last_color / last_alpha move on to this entry) */
                    for (int ch = 0; ch < C; ch++) {
                        abs_rec[ch] = last_alpha * fabsf(last_color[ch]) + (1.f - last_alpha) * abs_rec[ch];
                        const double ad = fabs((double)dL_dpixel[ch]);
                        abs_dalpha += ((double)fabsf(colors[gid * C + ch]) + abs_rec[ch]) * ad;
                        b_col[(size_t)gid * C + ch] += (double)dchannel_dcolor * ad;
                    }
                    if (dL_dinvdepths) {
                        const float invd = 1.f / depths[gid];
                        abs_rec[C] = last_alpha * abs_last_invdepth + (1.f - last_alpha) * abs_rec[C];
                        abs_last_invdepth = fabsf(invd);
                        abs_dalpha += ((double)fabsf(invd) + abs_rec[C]) * fabs((double)dL_invdepth);
                        b_inv[gid] += (double)dchannel_dcolor * fabs((double)dL_invdepth);
                    }
                    abs_dalpha *= T;
                    double abs_bg = 0;
                    for (int i = 0; i < C; i++) abs_bg += fabs((double)bg[i] * dL_dpixel[i]);
                    abs_dalpha += fabs((double)T_final / (1.0 - alpha)) * abs_bg;
                }
                for (int ch = 0; ch < C; ch++) {
                    const float c = colors[gid * C + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    a_col[(size_t)gid * C + ch] += (double)(dchannel_dcolor * dL_dchannel);
                }
                if (dL_dinvdepths) {
                    const float invd = 1.f / depths[gid];
                    accum_invdepth_rec = last_alpha * last_invdepth + (1.f - last_alpha) * accum_invdepth_rec;
                    last_invdepth = invd;
                    dL_dalpha += (invd - accum_invdepth_rec) * dL_invdepth;
                    a_inv[gid] += (double)(dchannel_dcolor * dL_invdepth);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < C; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                a_m2d[3 * gid + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                a_m2d[3 * gid + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                a_con[4 * gid + 0] += (double)(-0.5f * gdx * dx * dL_dG);
                a_con[4 * gid + 1] += (double)(-0.5f * gdx * dy * dL_dG);
                a_con[4 * gid + 3] += (double)(-0.5f * gdy * dy * dL_dG);
                a_op[gid] += (double)(G * dL_dalpha);
                if (abs_sums) {
                    const double adG = fabs((double)co[3]) * abs_dalpha;
                    const double agx = fabs((double)gdx), agy = fabs((double)gdy);
                    b_m2d[3 * gid + 0] += adG * (agx * fabs((double)co[0]) + agy * fabs((double)co[1])) * ddelx_dx;
                    b_m2d[3 * gid + 1] += adG * (agy * fabs((double)co[2]) + agx * fabs((double)co[1])) * ddely_dy;
                    b_con[4 * gid + 0] += 0.5 * agx * fabs((double)dx) * adG;
                    b_con[4 * gid + 1] += 0.5 * agx * fabs((double)dy) * adG;
                    b_con[4 * gid + 3] += 0.5 * agy * fabs((double)dy) * adG;
                    b_op[gid] += (double)G * abs_dalpha;
                }
            }
        }
    for (size_t i = 0; i < 3 * (size_t)P; i++) dL_dmean2D[i] = (float)a_m2d[i];
    for (size_t i = 0; i < 4 * (size_t)P; i++) dL_dconic2D[i] = (float)a_con[i];
    for (int i = 0; i < P; i++) dL_dopacity[i] = (float)a_op[i];
    for (size_t i = 0; i < (size_t)P * C; i++) dL_dcolors[i] = (float)a_col[i];
    if (dL_dinvdepths) for (int i = 0; i < P; i++) dL_dinvdepths[i] = (float)a_inv[i];
    free(acc); free(accum_rec); free(last_color); free(dL_dpixel); free(abs_rec);
}

/* ------------------------------------------------------------------------------------------ */
/* backward geometry: computeCov2DCUDA (backward.cu:147-326), preprocessCUDA (:398-449),        */
/* computeCov3D (:330-393).  SH backward (:443-444) is NOT reproduced.        */
/* dL_dopacity is in/out (antialiasing rescales it, :217-219).                                  */
/* ------------------------------------------------------------------------------------------ */
static inline float sq(float x) { return x * x; }

void orc_preprocess_bwd(int P, const float* means3D, const int* radii, const float* scales,
                        const float* rotations, float scale_modifier, const float* cov3Ds /*P*6, fwd*/,
                        const float* viewmatrix, const float* projmatrix, int W, int H,
                        float tan_fovx, float tan_fovy, const float* opacities, int antialiasing,
                        const float* dL_dmean2D /*P*3*/, const float* dL_dconics /*P*4*/,
                        const float* dL_dinvdepth /*P or NULL*/, float* dL_dopacity /*P in/out*/,
                        float* dL_dmeans /*P*3 out*/, float* dL_dcov /*P*6 out*/,
                        float* dL_dscales /*P*3 or NULL*/, float* dL_drots /*P*4 or NULL*/)
{
    const float h_y = H / (2.0f * tan_fovy);
    const float h_x = W / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * idx;
        const float* mean = means3D + 3 * idx;
        float dL_dconic[3] = { dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3] };
        Cov2DCtx c;
        cov2d_ctx(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c);
        const float* t = c.t;
        const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
        const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
        const mat3* T = &c.T; const mat3* Wm = &c.W; const mat3* Vrk = &c.Vrk;
        float c_xx = c.cov.m[0][0], c_xy = c.cov.m[0][1], c_yy = c.cov.m[1][1];

        const float h_var = 0.3f;
        float d_inside_root = 0.f;
        if (antialiasing) {
            const float det_cov = c_xx * c_yy - c_xy * c_xy;
            c_xx += h_var; c_yy += h_var;
            const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
            const float h_convolution_scaling = sqrtf(fmaxf_(0.000025f, det_cov / det_cov_plus_h_cov));
            const float dL_dopacity_v = dL_dopacity[idx];
            const float d_h_convolution_scaling = dL_dopacity_v * opacities[idx];
            dL_dopacity[idx] = dL_dopacity_v * h_convolution_scaling;
            d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
        } else { c_xx += h_var; c_yy += h_var; }

        float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
        if (antialiasing) {
            const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
            const float denom_f = d_inside_root / sq(w * w + w * (x + y) + x * y - z * z);
            dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
            dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
            dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
        }
        float denom = c_xx * c_yy - c_xy * c_xy;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_dc_xx += denom2inv * (-c_yy * c_yy * dL_dconic[0] + 2 * c_xy * c_yy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[2]);
            dL_dc_yy += denom2inv * (-c_xx * c_xx * dL_dconic[2] + 2 * c_xx * c_xy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[0]);
            dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dL_dconic[0] - (denom + 2 * c_xy * c_xy) * dL_dconic[1] + c_xx * c_xy * dL_dconic[2]);
            dcov[0] = (T->m[0][0] * T->m[0][0] * dL_dc_xx + T->m[0][0] * T->m[1][0] * dL_dc_xy + T->m[1][0] * T->m[1][0] * dL_dc_yy);
            dcov[3] = (T->m[0][1] * T->m[0][1] * dL_dc_xx + T->m[0][1] * T->m[1][1] * dL_dc_xy + T->m[1][1] * T->m[1][1] * dL_dc_yy);
            dcov[5] = (T->m[0][2] * T->m[0][2] * dL_dc_xx + T->m[0][2] * T->m[1][2] * dL_dc_xy + T->m[1][2] * T->m[1][2] * dL_dc_yy);
            dcov[1] = 2 * T->m[0][0] * T->m[0][1] * dL_dc_xx + (T->m[0][0] * T->m[1][1] + T->m[0][1] * T->m[1][0]) * dL_dc_xy + 2 * T->m[1][0] * T->m[1][1] * dL_dc_yy;
            dcov[2] = 2 * T->m[0][0] * T->m[0][2] * dL_dc_xx + (T->m[0][0] * T->m[1][2] + T->m[0][2] * T->m[1][0]) * dL_dc_xy + 2 * T->m[1][0] * T->m[1][2] * dL_dc_yy;
            dcov[4] = 2 * T->m[0][2] * T->m[0][1] * dL_dc_xx + (T->m[0][1] * T->m[1][2] + T->m[0][2] * T->m[1][1]) * dL_dc_xy + 2 * T->m[1][1] * T->m[1][2] * dL_dc_yy;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        float dL_dT00 = 2 * (T->m[0][0] * Vrk->m[0][0] + T->m[0][1] * Vrk->m[0][1] + T->m[0][2] * Vrk->m[0][2]) * dL_dc_xx +
                        (T->m[1][0] * Vrk->m[0][0] + T->m[1][1] * Vrk->m[0][1] + T->m[1][2] * Vrk->m[0][2]) * dL_dc_xy;
        float dL_dT01 = 2 * (T->m[0][0] * Vrk->m[1][0] + T->m[0][1] * Vrk->m[1][1] + T->m[0][2] * Vrk->m[1][2]) * dL_dc_xx +
                        (T->m[1][0] * Vrk->m[1][0] + T->m[1][1] * Vrk->m[1][1] + T->m[1][2] * Vrk->m[1][2]) * dL_dc_xy;
        float dL_dT02 = 2 * (T->m[0][0] * Vrk->m[2][0] + T->m[0][1] * Vrk->m[2][1] + T->m[0][2] * Vrk->m[2][2]) * dL_dc_xx +
                        (T->m[1][0] * Vrk->m[2][0] + T->m[1][1] * Vrk->m[2][1] + T->m[1][2] * Vrk->m[2][2]) * dL_dc_xy;
        float dL_dT10 = 2 * (T->m[1][0] * Vrk->m[0][0] + T->m[1][1] * Vrk->m[0][1] + T->m[1][2] * Vrk->m[0][2]) * dL_dc_yy +
                        (T->m[0][0] * Vrk->m[0][0] + T->m[0][1] * Vrk->m[0][1] + T->m[0][2] * Vrk->m[0][2]) * dL_dc_xy;
        float dL_dT11 = 2 * (T->m[1][0] * Vrk->m[1][0] + T->m[1][1] * Vrk->m[1][1] + T->m[1][2] * Vrk->m[1][2]) * dL_dc_yy +
                        (T->m[0][0] * Vrk->m[1][0] + T->m[0][1] * Vrk->m[1][1] + T->m[0][2] * Vrk->m[1][2]) * dL_dc_xy;
        float dL_dT12 = 2 * (T->m[1][0] * Vrk->m[2][0] + T->m[1][1] * Vrk->m[2][1] + T->m[1][2] * Vrk->m[2][2]) * dL_dc_yy +
                        (T->m[0][0] * Vrk->m[2][0] + T->m[0][1] * Vrk->m[2][1] + T->m[0][2] * Vrk->m[2][2]) * dL_dc_xy;

        float dL_dJ00 = Wm->m[0][0] * dL_dT00 + Wm->m[0][1] * dL_dT01 + Wm->m[0][2] * dL_dT02;
        float dL_dJ02 = Wm->m[2][0] * dL_dT00 + Wm->m[2][1] * dL_dT01 + Wm->m[2][2] * dL_dT02;
        float dL_dJ11 = Wm->m[1][0] * dL_dT10 + Wm->m[1][1] * dL_dT11 + Wm->m[1][2] * dL_dT12;
        float dL_dJ12 = Wm->m[2][0] * dL_dT10 + Wm->m[2][1] * dL_dT11 + Wm->m[2][2] * dL_dT12;

        float tz = 1.f / t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
        if (dL_dinvdepth) dL_dtz -= dL_dinvdepth[idx] / (t[2] * t[2]);

        /* transformVec4x3Transpose, auxiliary.h:101-109 */
        const float* V = viewmatrix;
        float dL_dmean[3] = {
            V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz,
            V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz,
            V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz };

        /* preprocessCUDA backward, backward.cu:423-440 */
        const float* proj = projmatrix;
        const float* m = mean;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        float d2[3];
        d2[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        d2[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        d2[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmeans[3 * idx + 0] = dL_dmean[0] + d2[0];
        dL_dmeans[3 * idx + 1] = dL_dmean[1] + d2[1];
        dL_dmeans[3 * idx + 2] = dL_dmean[2] + d2[2];

        /* computeCov3D backward, backward.cu:330-393 */
        if (scales && dL_dscales && dL_drots) {
            const float* q = rotations + 4 * idx;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            mat3 R;
            R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
            R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
            R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
            mat3 S; memset(&S, 0, sizeof(S));
            float s[3] = { scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2] };
            S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
            mat3 M = mat3_mul(&S, &R);
            mat3 dS;
            dS.m[0][0] = dcov[0]; dS.m[0][1] = 0.5f * dcov[1]; dS.m[0][2] = 0.5f * dcov[2];
            dS.m[1][0] = 0.5f * dcov[1]; dS.m[1][1] = dcov[3]; dS.m[1][2] = 0.5f * dcov[4];
            dS.m[2][0] = 0.5f * dcov[2]; dS.m[2][1] = 0.5f * dcov[4]; dS.m[2][2] = dcov[5];
            mat3 M2;
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) M2.m[cc][rr] = 2.0f * M.m[cc][rr];
            mat3 dL_dM = mat3_mul(&M2, &dS);
            mat3 Rt = mat3_T(&R);
            mat3 dMt = mat3_T(&dL_dM);
            for (int k = 0; k < 3; k++)
                dL_dscales[3 * idx + k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
            for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dMt.m[k][rr] *= s[k];
            float dq[4];
            dq[0] = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
            dq[1] = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
            dq[2] = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
            dq[3] = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
            for (int k = 0; k < 4; k++) dL_drots[4 * idx + k] = dq[k];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Rounding scale of orc_preprocess_bwd's outputs (tests/fuzz_cases.py): the same chain of     */
/* formulas evaluated on the ABSOLUTE values -- inputs A_* = orc_render_bwd's abs_sums, every  */
/* coefficient |.|, every difference a sum, divisors at their forward values -- so that        */
/* B_out >= sum over all paths of |path product| >= |J| A_in.  Two fp32 evaluations of the     */
/* backward that differ in summation order, contraction or reciprocal may differ by about      */
/* (a few dozen) x 2^-24 x B_out, however small the gradient itself comes out by cancellation. */
/* Not a restatement of reference code: test arithmetic.                                       */
/* ------------------------------------------------------------------------------------------ */
static inline double dsq(double x) { return x * x; }

void orc_preprocess_bwd_bound(int P, const float* means3D, const int* radii, const float* scales,
                              const float* rotations, float scale_modifier, const float* cov3Ds,
                              const float* viewmatrix, const float* projmatrix, int W, int H,
                              float tan_fovx, float tan_fovy, const float* opacities, int antialiasing,
                              const double* A_mean2D /*P*3*/, const double* A_conics /*P*4*/,
                              const double* A_invdepth /*P or NULL*/, const double* A_opacity /*P*/,
                              double* B_opacity /*P*/, double* B_means /*P*3*/, double* B_cov /*P*6*/,
                              double* B_scales /*P*3 or NULL*/, double* B_rots /*P*4 or NULL*/)
{
    const float h_y = H / (2.0f * tan_fovy);
    const float h_x = W / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++) {
        B_opacity[idx] = 0;
        for (int i = 0; i < 3; i++) B_means[3 * idx + i] = 0;
        for (int i = 0; i < 6; i++) B_cov[6 * idx + i] = 0;
        if (B_scales) for (int i = 0; i < 3; i++) B_scales[3 * idx + i] = 0;
        if (B_rots) for (int i = 0; i < 4; i++) B_rots[4 * idx + i] = 0;
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * idx;
        const float* mean = means3D + 3 * idx;
        const double Ac[3] = { A_conics[4 * idx], A_conics[4 * idx + 1], A_conics[4 * idx + 3] };
        Cov2DCtx c;
        cov2d_ctx(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c);
        const float* t = c.t;
        const double x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
        const double y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
        double Ta[3][3], Wa[3][3], Va[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            Ta[i][j] = fabs((double)c.T.m[i][j]); Wa[i][j] = fabs((double)c.W.m[i][j]); Va[i][j] = fabs((double)c.Vrk.m[i][j]);
        }
        double c_xx = c.cov.m[0][0], c_xy = c.cov.m[0][1], c_yy = c.cov.m[1][1];
        const double h_var = 0.3;
        double d_inside_root = 0;
        double Bop = A_opacity[idx];
        if (antialiasing) {
            const double det_cov = c_xx * c_yy - c_xy * c_xy;
            c_xx += h_var; c_yy += h_var;
            const double det2 = c_xx * c_yy - c_xy * c_xy;
            const double hcs = sqrt(fmax(0.000025, det_cov / det2));
            d_inside_root = (det_cov / det2) <= 0.000025 ? 0.0 : A_opacity[idx] * fabs((double)opacities[idx]) / (2 * hcs);
            Bop = A_opacity[idx] * hcs;
        } else { c_xx += h_var; c_yy += h_var; }
        B_opacity[idx] = Bop;
        const double axx = fabs(c_xx), ayy = fabs(c_yy), axy = fabs(c_xy);
        double Bxx = 0, Bxy = 0, Byy = 0;
        if (antialiasing) {
            const double x = c_xx, y = c_yy, z = c_xy, w = h_var;
            const double denom_f = d_inside_root / dsq(w * w + w * (x + y) + x * y - z * z);
            Bxx = w * (w * ayy + ayy * ayy + z * z) * denom_f;
            Byy = w * (w * axx + axx * axx + z * z) * denom_f;
            Bxy = 2. * w * axy * (w + axx + ayy) * denom_f;
        }
        const double denom = c_xx * c_yy - c_xy * c_xy;
        const double denom2inv = 1.0 / ((denom * denom) + 0.0000001);
        const double admx = fabs(denom) + axx * ayy;          /* |denom - c_xx c_yy| as a sum */
        double* dcov = B_cov + 6 * idx;
        if ((float)(1.0f / (((float)denom * (float)denom) + 0.0000001f)) != 0) {
            Bxx += denom2inv * (ayy * ayy * Ac[0] + 2 * axy * ayy * Ac[1] + admx * Ac[2]);
            Byy += denom2inv * (axx * axx * Ac[2] + 2 * axx * axy * Ac[1] + admx * Ac[0]);
            Bxy += denom2inv * 2 * (axy * ayy * Ac[0] + (fabs(denom) + 2 * axy * axy) * Ac[1] + axx * axy * Ac[2]);
            dcov[0] = Ta[0][0] * Ta[0][0] * Bxx + Ta[0][0] * Ta[1][0] * Bxy + Ta[1][0] * Ta[1][0] * Byy;
            dcov[3] = Ta[0][1] * Ta[0][1] * Bxx + Ta[0][1] * Ta[1][1] * Bxy + Ta[1][1] * Ta[1][1] * Byy;
            dcov[5] = Ta[0][2] * Ta[0][2] * Bxx + Ta[0][2] * Ta[1][2] * Bxy + Ta[1][2] * Ta[1][2] * Byy;
            dcov[1] = 2 * Ta[0][0] * Ta[0][1] * Bxx + (Ta[0][0] * Ta[1][1] + Ta[0][1] * Ta[1][0]) * Bxy + 2 * Ta[1][0] * Ta[1][1] * Byy;
            dcov[2] = 2 * Ta[0][0] * Ta[0][2] * Bxx + (Ta[0][0] * Ta[1][2] + Ta[0][2] * Ta[1][0]) * Bxy + 2 * Ta[1][0] * Ta[1][2] * Byy;
            dcov[4] = 2 * Ta[0][2] * Ta[0][1] * Bxx + (Ta[0][1] * Ta[1][2] + Ta[0][2] * Ta[1][1]) * Bxy + 2 * Ta[1][1] * Ta[1][2] * Byy;
        }
        double TV[2][3];   /* sum_k |T[i][k]| |Vrk[j][k]| */
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
            TV[i][j] = Ta[i][0] * Va[j][0] + Ta[i][1] * Va[j][1] + Ta[i][2] * Va[j][2];
        const double BT0[3] = { 2 * TV[0][0] * Bxx + TV[1][0] * Bxy, 2 * TV[0][1] * Bxx + TV[1][1] * Bxy, 2 * TV[0][2] * Bxx + TV[1][2] * Bxy };
        const double BT1[3] = { 2 * TV[1][0] * Byy + TV[0][0] * Bxy, 2 * TV[1][1] * Byy + TV[0][1] * Bxy, 2 * TV[1][2] * Byy + TV[0][2] * Bxy };
        const double BJ00 = Wa[0][0] * BT0[0] + Wa[0][1] * BT0[1] + Wa[0][2] * BT0[2];
        const double BJ02 = Wa[2][0] * BT0[0] + Wa[2][1] * BT0[1] + Wa[2][2] * BT0[2];
        const double BJ11 = Wa[1][0] * BT1[0] + Wa[1][1] * BT1[1] + Wa[1][2] * BT1[2];
        const double BJ12 = Wa[2][0] * BT1[0] + Wa[2][1] * BT1[1] + Wa[2][2] * BT1[2];
        const double tz = fabs(1.0 / t[2]), tz2 = tz * tz, tz3 = tz2 * tz;
        const double Btx = x_grad_mul * h_x * tz2 * BJ02;
        const double Bty = y_grad_mul * h_y * tz2 * BJ12;
        double Btz = h_x * tz2 * BJ00 + h_y * tz2 * BJ11 + fabs(2. * h_x * t[0]) * tz3 * BJ02 + fabs(2. * h_y * t[1]) * tz3 * BJ12;
        if (A_invdepth) Btz += A_invdepth[idx] / ((double)t[2] * t[2]);
        const float* V = viewmatrix;
        double Bm[3] = {
            fabs((double)V[0]) * Btx + fabs((double)V[1]) * Bty + fabs((double)V[2]) * Btz,
            fabs((double)V[4]) * Btx + fabs((double)V[5]) * Bty + fabs((double)V[6]) * Btz,
            fabs((double)V[8]) * Btx + fabs((double)V[9]) * Bty + fabs((double)V[10]) * Btz };
        const float* proj = projmatrix;
        const float* m = mean;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        const double m_w = fabs(1.0 / ((double)m_hom[3] + 0.0000001));
        const double mul1 = (fabs((double)proj[0] * m[0]) + fabs((double)proj[4] * m[1]) + fabs((double)proj[8] * m[2]) + fabs((double)proj[12])) * m_w * m_w;
        const double mul2 = (fabs((double)proj[1] * m[0]) + fabs((double)proj[5] * m[1]) + fabs((double)proj[9] * m[2]) + fabs((double)proj[13])) * m_w * m_w;
        const double g2x = A_mean2D[3 * idx], g2y = A_mean2D[3 * idx + 1];
        for (int k = 0; k < 3; k++)
            B_means[3 * idx + k] = Bm[k] + (fabs((double)proj[4 * k]) * m_w + fabs((double)proj[4 * k + 3]) * mul1) * g2x +
                                   (fabs((double)proj[4 * k + 1]) * m_w + fabs((double)proj[4 * k + 3]) * mul2) * g2y;
        if (scales && B_scales && B_rots) {
            const float* q = rotations + 4 * idx;
            const double r = fabs((double)q[0]), x = fabs((double)q[1]), y = fabs((double)q[2]), z = fabs((double)q[3]);
            double R[3][3];     /* |entries| of the rotation as sums (glm layout m[c][r] as in orc_preprocess_bwd) */
            R[0][0] = 1. + 2. * (y * y + z * z); R[0][1] = 2. * (x * y + r * z); R[0][2] = 2. * (x * z + r * y);
            R[1][0] = 2. * (x * y + r * z); R[1][1] = 1. + 2. * (x * x + z * z); R[1][2] = 2. * (y * z + r * x);
            R[2][0] = 2. * (x * z + r * y); R[2][1] = 2. * (y * z + r * x); R[2][2] = 1. + 2. * (x * x + y * y);
            const double s[3] = { fabs((double)scale_modifier * scales[3 * idx]), fabs((double)scale_modifier * scales[3 * idx + 1]),
                                  fabs((double)scale_modifier * scales[3 * idx + 2]) };
            double M[3][3], dS[3][3], dM[3][3];
            /* M = S * R in mat3_mul's convention: o.m[c][r] = sum_k a.m[k][r] b.m[c][k], S diagonal */
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) M[cc][rr] = 2.0 * s[rr] * R[cc][rr];
            dS[0][0] = dcov[0]; dS[0][1] = 0.5 * dcov[1]; dS[0][2] = 0.5 * dcov[2];
            dS[1][0] = 0.5 * dcov[1]; dS[1][1] = dcov[3]; dS[1][2] = 0.5 * dcov[4];
            dS[2][0] = 0.5 * dcov[2]; dS[2][1] = 0.5 * dcov[4]; dS[2][2] = dcov[5];
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++)
                dM[cc][rr] = M[0][rr] * dS[cc][0] + M[1][rr] * dS[cc][1] + M[2][rr] * dS[cc][2];
            double dMt[3][3], Rt[3][3];
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) { dMt[cc][rr] = dM[rr][cc]; Rt[cc][rr] = R[rr][cc]; }
            for (int k = 0; k < 3; k++)
                B_scales[3 * idx + k] = Rt[k][0] * dMt[k][0] + Rt[k][1] * dMt[k][1] + Rt[k][2] * dMt[k][2];
            for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dMt[k][rr] *= s[k];
            B_rots[4 * idx + 0] = 2 * z * (dMt[0][1] + dMt[1][0]) + 2 * y * (dMt[2][0] + dMt[0][2]) + 2 * x * (dMt[1][2] + dMt[2][1]);
            B_rots[4 * idx + 1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] + dMt[2][1]) + 4 * x * (dMt[2][2] + dMt[1][1]);
            B_rots[4 * idx + 2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] + dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) + 4 * y * (dMt[2][2] + dMt[0][0]);
            B_rots[4 * idx + 3] = 2 * r * (dMt[0][1] + dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) + 4 * z * (dMt[1][1] + dMt[0][0]);
        }
    }
}
