// sks_math.h -- device math shared by the rasterizer kernels (gfx950).
//
// Numeric contract (DESIGN.md "Numerics"): every list-determining quantity (depth, radius, tile rect, pixel
// centre) and the compositor are evaluated in fp32 in the reference's written operation order with FMA
// contraction OFF (the translation unit is built with -ffp-contract=off), IEEE division / sqrt, and a fixed-
// sequence expf, so tile lists, n_contrib and forward images are reproducible bit-for-bit on any IEEE host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sks {

constexpr int TILE = 16;  // BLOCK_X == BLOCK_Y, DGR/cuda_rasterizer/config.h:16-17

// column-major 3x3 like glm::mat3: m[c][r]; products evaluated left-to-right as glm's operator* does.
struct M3 {
    float m[3][3];
};

__device__ __forceinline__ M3 m3mul(const M3& a, const M3& b)
{
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a.m[0][r] * b.m[c][0] + a.m[1][r] * b.m[c][1] + a.m[2][r] * b.m[c][2];
    return o;
}

__device__ __forceinline__ M3 m3T(const M3& a)
{
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a.m[r][c];
    return o;
}

// exp(x), x <= 0 in the compositor (forward.cu:364, backward.cu:568).  Cody-Waite reduction + degree-7 Horner
// in explicit fmaf: the same IEEE operation sequence as the CPU oracle, <= 1 ulp.
__device__ __forceinline__ float expf_fixed(float x)
{
    if (x < -80.0f) return 0.0f;
    if (x > 88.0f) return __builtin_huge_valf();
    float t = x * 1.44269504088896341f;
    float k = __builtin_rintf(t);
    float r = __builtin_fmaf(k, -0.693145751953125f, x);
    r = __builtin_fmaf(k, -1.42860682030941723e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = __builtin_fmaf(p, r, 1.38888888888888894e-3f);
    p = __builtin_fmaf(p, r, 8.33333333333333322e-3f);
    p = __builtin_fmaf(p, r, 4.16666666666666644e-2f);
    p = __builtin_fmaf(p, r, 1.66666666666666657e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int ki = (int)k;
    return __int_as_float(__float_as_int(p) + ki * (1 << 23));
}

// The same polynomial without the two range branches, for compositor loops that only use the result where power <= 0 and
// reject alpha < 1/255: x is clamped at -80 (exp(-80) = 1.8e-35 is rejected like the 0 expf_fixed returns below -80), x > 0
// never reaches a use.  Bit-identical to expf_fixed on [-80, 0].
__device__ __forceinline__ float expf_fixed_neg(float x)
{
    x = fmaxf(x, -80.0f);
    float t = x * 1.44269504088896341f;
    float k = __builtin_rintf(t);
    float r = __builtin_fmaf(k, -0.693145751953125f, x);
    r = __builtin_fmaf(k, -1.42860682030941723e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = __builtin_fmaf(p, r, 1.38888888888888894e-3f);
    p = __builtin_fmaf(p, r, 8.33333333333333322e-3f);
    p = __builtin_fmaf(p, r, 4.16666666666666644e-2f);
    p = __builtin_fmaf(p, r, 1.66666666666666657e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int ki = (int)k;
    return __int_as_float(__float_as_int(p) + ki * (1 << 23));
}

// auxiliary.h:40-43 -- double arithmetic as written in the reference
__device__ __forceinline__ float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// auxiliary.h:45-55
__device__ __forceinline__ void getRect(float px, float py, int max_radius, int gx, int gy, int& xmin, int& ymin,
                                        int& xmax, int& ymax)
{
    xmin = imin(gx, imax(0, (int)((px - max_radius) / TILE)));
    ymin = imin(gy, imax(0, (int)((py - max_radius) / TILE)));
    xmax = imin(gx, imax(0, (int)((px + max_radius + TILE - 1) / TILE)));
    ymax = imin(gy, imax(0, (int)((py + max_radius + TILE - 1) / TILE)));
}

// auxiliary.h:70-89
__device__ __forceinline__ void transformPoint4x3(const float p[3], const float* M, float o[3])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
__device__ __forceinline__ void transformPoint4x4(const float p[3], const float* M, float o[4])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

// quaternion (r,x,y,z) -> glm-style R (columns as the reference's ctor lists them), forward.cu:123-134
__device__ __forceinline__ M3 quatR(const float q[4])
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    M3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// forward.cu:114-150 (quaternion NOT normalised, :123)
__device__ __forceinline__ void computeCov3D(const float s[3], float mod, const float q[4], float cov3D[6])
{
    M3 S = {};
    S.m[0][0] = mod * s[0];
    S.m[1][1] = mod * s[1];
    S.m[2][2] = mod * s[2];
    M3 R = quatR(q);
    M3 M = m3mul(S, R);
    M3 Sigma = m3mul(m3T(M), M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

// shared by forward.cu:74-109 and backward.cu:173-201
struct Cov2D {
    float t[3];
    float txtz, tytz, limx, limy;
    M3 W, T, Vrk, cov;
};

__device__ __forceinline__ void cov2d(const float mean[3], float fx, float fy, float tanfovx, float tanfovy,
                                      const float* cov3D, const float* V, Cov2D& c)
{
    transformPoint4x3(mean, V, c.t);
    c.limx = 1.3f * tanfovx;
    c.limy = 1.3f * tanfovy;
    c.txtz = c.t[0] / c.t[2];
    c.tytz = c.t[1] / c.t[2];
    c.t[0] = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * c.t[2];
    c.t[1] = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * c.t[2];
    M3 J = {};
    J.m[0][0] = fx / c.t[2]; J.m[0][2] = -(fx * c.t[0]) / (c.t[2] * c.t[2]);
    J.m[1][1] = fy / c.t[2]; J.m[1][2] = -(fy * c.t[1]) / (c.t[2] * c.t[2]);
    c.W.m[0][0] = V[0]; c.W.m[0][1] = V[4]; c.W.m[0][2] = V[8];
    c.W.m[1][0] = V[1]; c.W.m[1][1] = V[5]; c.W.m[1][2] = V[9];
    c.W.m[2][0] = V[2]; c.W.m[2][1] = V[6]; c.W.m[2][2] = V[10];
    c.T = m3mul(c.W, J);
    c.Vrk.m[0][0] = cov3D[0]; c.Vrk.m[0][1] = cov3D[1]; c.Vrk.m[0][2] = cov3D[2];
    c.Vrk.m[1][0] = cov3D[1]; c.Vrk.m[1][1] = cov3D[3]; c.Vrk.m[1][2] = cov3D[4];
    c.Vrk.m[2][0] = cov3D[2]; c.Vrk.m[2][1] = cov3D[4]; c.Vrk.m[2][2] = cov3D[5];
    c.cov = m3mul(m3mul(m3T(c.T), m3T(c.Vrk)), c.T);
}

// Sum over the 64 lanes of a wavefront, returned in EVERY lane.  Inside each row of 16 lanes the partial sums move
// with DPP modifiers (VALU speed; a __shfl is a ds_bpermute through the LDS crossbar, ~10x the latency, and a
// reduction is a chain of them); the four row sums are read with v_readlane.  Fixed order -> reproducible.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x124>(v);   // row_ror:4
    v += dpp_move<0x128>(v);   // row_ror:8  -> every lane holds the sum of its row of 16
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// double-precision twin of wave_sum (same scheme, returned in every lane)
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_move_d<0xB1>(v);
    v += dpp_move_d<0x4E>(v);
    v += dpp_move_d<0x124>(v);
    v += dpp_move_d<0x128>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// Eight sums over the wavefront at once: on return lane L (L < 8) -- in fact every lane with L % 8 == idx -- holds the
// total of v[idx] over the 64 lanes.  Each of the first three steps halves the number of live values (a lane keeps
// the value its lane-index bit selects and hands the other to its partner), so the whole thing costs ~25 VALU ops and
// two cross-row shuffles instead of eight full reductions.
__device__ __forceinline__ float wave_sum8(const float (&v)[8])
{
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float w[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {   // partner = lane ^ 1
        const float keep = b0 ? v[2 * p + 1] : v[2 * p], give = b0 ? v[2 * p] : v[2 * p + 1];
        w[p] = keep + dpp_move<0xB1>(give);
    }
    float x[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {   // partner = lane ^ 2
        const float keep = b1 ? w[2 * p + 1] : w[2 * p], give = b1 ? w[2 * p] : w[2 * p + 1];
        x[p] = keep + dpp_move<0x4E>(give);
    }
    // lane i now holds value (i & 3) of x[0] (values 0..3) / x[1] (values 4..7), summed over its quad
    float z = (b2 ? x[1] : x[0]) + dpp_move<0x124>(b2 ? x[0] : x[1]);   // from lane i - 4 (mod 16): the other bit-2 class
    z += dpp_move<0x128>(z);                                             // from lane i - 8 (mod 16): the row's other two quads
    // the other rows: gfx950's row / half swaps (VALU; a __shfl_xor is a ds_bpermute: an LDS round trip in the middle of the
    // caller's loop).  v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
    // v_permlane32_swap the upper half of the first with the lower half of the second: fed z twice, the two results are
    // (z of the even / lower partner, z of the odd / upper partner) in every lane -- their sum is z + z[lane ^ 16 / 32].
    {
        const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
        z = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    }
    {
        const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
        z = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    }
    return z;
}

}  // namespace sks
