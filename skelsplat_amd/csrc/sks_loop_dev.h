// sks_loop_dev.h -- device side of the optimiser step of the multi-view loop (train.py:160-222), shared by
// k_loop_adam (sks_loop.hip) and the fused step tail k_step_tail (sks_raster.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace sksloop {

struct AdamArgs {
    int V, P;
    const float* grads;        // (V,P,11) raw-parameter gradients of every view (this group)
    float* slots;              // (V,P,3) persistent per-view xyz gradient slots (train.py:121,175)
    unsigned long long group_mask;  // views rendered in this group (bit v)
    int last_view;             // view of the group's last iteration: its scaling/rotation/opacity grads win (Q7)
    float* xyz; float* scaling; float* rotation; float* opacity;     // raw leaf parameters, updated in place
    float* m; float* vv;       // Adam moments, (P,11) each, same packing as the gradients
    int* counters;             // [0] iteration (advanced by acc_steps), [1] Adam step count
    int acc_steps;
    double lr_init, lr_final, lr_delay_mult; int lr_delay_steps, lr_max_steps;  // xyz schedule (general_utils.py:38-71)
    double log_lr_init, log_lr_final;   // np.log of the two, taken on the host like the reference does
    double lr_scaling, lr_rotation, lr_opacity;
    double beta1, beta2, eps;
    float lambda_consistency;
    int limb[8];               // l_arm, r_arm, l_leg, r_leg joint index pairs (loss_utils.py:226-250); limb[0] < 0: none
    int n_joints;              // joints per skeleton (limb indices address the first skeleton, like the reference)
    // layout of `grads`: world == 1: view-major (V,P,11).  world > 1: what all_gather_into_tensor leaves when view v is
    // rendered by rank v % world as that rank's local view v / world and every rank contributes vmax = ceil(V / world)
    // rows: (world, vmax, P, 11), so the step reads the gathered buffer in place (no re-ordering pass in between).
    // rank_stride (floats) >= vmax * P * 11: a rank's block may carry a tail behind its rows -- the views' loss sums, so that
    // the early-stopping criterion rides in the SAME collective as the gradients (train.py:155)
    int world, vmax;
    long long rank_stride;
    // device-side opt_early_stopping (utils/general_utils.py:467-491, train.py:155,182-233); es_state == nullptr: off
    int* es_state;             // [0] losses seen so far, [1] iteration the scene stopped at (0 = running),
                               // [2 .. 2 + 2 * es_window): the last 2 * window losses (float bits), a ring
    int es_window;
    float es_tol;
    const double* es_sums;     // world == 1: (V,2) {S, N} of the views; world > 1: nullptr = the tail of each rank's block
    int* es_host_flag;         // pinned host int or nullptr: receives the stopping iteration when the criterion fires
};

// view v's (P,11) gradient rows in AdamArgs::grads
__device__ __forceinline__ const float* grad_rows(const AdamArgs& a, int v)
{
    if (a.world == 1) return a.grads + (size_t)v * a.P * 11;
    const int r = v % a.world;
    return a.grads + (size_t)r * a.rank_stride + (size_t)((v - r) / a.world) * a.P * 11;
}
// where a rank's block keeps its views' loss sums (doubles: the rows padded to an even number of floats)
__host__ __device__ inline long long es_tail_offset(int vmax, int P) { return ((long long)vmax * P * 11 + 1) & ~1ll; }
__device__ __forceinline__ const double* loss_sums_of(const AdamArgs& a, int v)
{
    if (a.world == 1 || a.es_sums) return a.es_sums + 2 * (size_t)(a.world == 1 ? v : (v % a.world) * a.vmax + v / a.world);
    const int r = v % a.world;
    return reinterpret_cast<const double*>(a.grads + (size_t)r * a.rank_stride + es_tail_offset(a.vmax, a.P)) + 2 * ((v - r) / a.world);
}

// The reference's criterion, one thread, BEFORE the step: train.py:155 feeds OptEarlyStopping every iteration's loss
// (masked L2 of that iteration's view + lambda x limb loss, an fp32 number) in order; it keeps the history and fires when the
// last `window` losses repeat the `window` before them to within `tol` (general_utils.py:483-491, compared in fp32).  When it
// fires at the k-th iteration of this group, only the first k views refresh their slots, view k's scaling / rotation / opacity
// gradients win, the optimiser steps at once (train.py:182) and the scene has ended: every later launch of the step does
// nothing.  out: cut[0] = last view, cut[1] = iterations of this step, cut[2] = 1 if the scene had stopped before.
constexpr int ES_MAX_WINDOW = 16;
__device__ inline void early_stop_decide(const AdamArgs& a, const float* s_xyz, unsigned long long* mask_out, int* cut)
{
    int* st = a.es_state;
    cut[0] = a.last_view; cut[1] = a.acc_steps; cut[2] = 0;
    *mask_out = a.group_mask;
    if (st[1] != 0) { cut[2] = 1; return; }
    float cons = 0.0f;
    if (a.lambda_consistency != 0.0f && a.limb[0] >= 0) {
        float len[4];
        for (int k = 0; k < 4; k++) {
            const int i0 = a.limb[2 * k], i1 = a.limb[2 * k + 1];
            const float d0 = s_xyz[3 * i0] - s_xyz[3 * i1], d1 = s_xyz[3 * i0 + 1] - s_xyz[3 * i1 + 1], d2 = s_xyz[3 * i0 + 2] - s_xyz[3 * i1 + 2];
            len[k] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        }
        cons = (fabsf(len[0] - len[1]) + fabsf(len[2] - len[3])) * a.lambda_consistency;
    }
    const int w = a.es_window, ring = 2 * w;
    const int it0 = a.counters[0] + 1;
    int n = st[0];
    unsigned long long mask = 0ull;
    for (int k = 0; k < a.acc_steps; k++) {
        const int v = (it0 + k - 1) % a.V;
        mask |= 1ull << v;
        const double* sn = loss_sums_of(a, v);
        const double cnt = sn[1] < 1.0 ? 1.0 : sn[1];
        const float loss = (float)(sn[0] / cnt) + cons;
        st[2 + n % ring] = __float_as_int(loss);
        n++;
        bool fire = n >= ring;
        for (int i = 0; fire && i < w; i++) {
            const float h1 = __int_as_float(st[2 + (n - ring + i) % ring]), h2 = __int_as_float(st[2 + (n - w + i) % ring]);
            fire = fabsf(h1 - h2) < a.es_tol;
        }
        if (fire) {
            st[1] = it0 + k;
            if (a.es_host_flag) *a.es_host_flag = it0 + k;
            cut[0] = v; cut[1] = k + 1;
            *mask_out = mask;
            break;
        }
    }
    st[0] = n;
}

// torch.optim.Adam single-tensor path (python scalars are doubles, tensor math is fp32):
//   exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2);
//   denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr/bias_correction1)
__device__ __forceinline__ void adam_update(float& param, float g, float& m, float& v, float w1, float b2, float w2,
                                            float eps, float step_size, float bc2_sqrt)
{
    m = m + w1 * (g - m);
    v = v * b2 + w2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    param = param - step_size * (m / denom);
}

// LDS mirror of the updated parameters (k_step_tail hands them to the next geometry pass without a trip to memory)
struct AdamLdsParams {
    float* xyz;       // 3 * P
    float* scaling;   // 3 * P
    float* rotation;  // 4 * P
    float* opacity;   // P
};

// One optimiser step by one 256-thread workgroup (thread p < P owns Gaussian p), in two parts that every thread of the
// workgroup must call, with a __syncthreads() between them:
//   adam_block_begin : reads the step counters and the current xyz, starts the LR-schedule / bias-correction
//                      transcendentals (needs nothing from the gradients, so a caller can overlap it with other work);
//   adam_block_finish: slots, mean over views, limb gradient, Adam update, counters.
// s_xyz: 768 floats, s_hyp: 6 floats, s_d: 4 doubles, s_it: 2 ints of LDS.
__device__ __forceinline__ void adam_block_begin(const AdamArgs& a, float* s_xyz, double* s_d, int* s_it)
{
    const int p = threadIdx.x;
    const int P = a.P;
    const bool live = p < P;
    const int it1 = a.counters[0] + a.acc_steps;   // iteration at which the optimiser steps (train.py:182)
    const int step = a.counters[1] + 1;
    if (p == 0) { s_it[0] = it1; s_it[1] = step; }
    if (live) {
        s_xyz[3 * p] = a.xyz[3 * p]; s_xyz[3 * p + 1] = a.xyz[3 * p + 1]; s_xyz[3 * p + 2] = a.xyz[3 * p + 2];
    }
    // LR schedule + bias corrections, in double like the host code (train.py:134, quirk Q9).  The four double
    // transcendentals are a few hundred dependent instructions each; one lane of each of the four wavefronts takes
    // one of them so that they run side by side (different lanes of ONE wavefront would serialise).
    // s_d: delay factor, exp(interpolated log lr), beta1^step, beta2^step
    if ((p & 63) == 0) {
        const int w = p >> 6;
        const bool sched = !(a.lr_init == 0.0 && a.lr_final == 0.0);
        if (w == 0) {
            double delay = 1.0;
            if (sched && a.lr_delay_steps > 0) {
                const double c = fmin(fmax((double)it1 / (double)a.lr_delay_steps, 0.0), 1.0);
                delay = a.lr_delay_mult + (1.0 - a.lr_delay_mult) * sin(0.5 * 3.14159265358979323846 * c);
            }
            s_d[0] = delay;
        } else if (w == 1) {
            const double t = fmin(fmax((double)it1 / (double)a.lr_max_steps, 0.0), 1.0);
            s_d[1] = sched ? exp(a.log_lr_init * (1.0 - t) + a.log_lr_final * t) : 0.0;
        } else if (w == 2) {
            s_d[2] = pow(a.beta1, (double)step);
        } else {
            s_d[3] = pow(a.beta2, (double)step);
        }
    }
}

// s_hyp: step sizes xyz / scaling / rotation / opacity, sqrt(bias_correction2), spare -- from adam_block_begin's s_d
__device__ __forceinline__ void adam_step_sizes(const AdamArgs& a, const double* s_d, float* s_hyp)
{
    const double lr_xyz = s_d[0] * s_d[1];
    const double bc1 = 1.0 - s_d[2];
    s_hyp[0] = (float)(lr_xyz / bc1);
    s_hyp[1] = (float)(a.lr_scaling / bc1);
    s_hyp[2] = (float)(a.lr_rotation / bc1);
    s_hyp[3] = (float)(a.lr_opacity / bc1);
    s_hyp[4] = (float)sqrt(1.0 - s_d[3]);
}

__device__ __forceinline__ void adam_block_finish(const AdamArgs& a, float* s_xyz, float* s_hyp, double* s_d, int* s_it,
                                                  const AdamLdsParams* mirror = nullptr, bool step_sizes_ready = false)
{
    const int p = threadIdx.x;
    const int P = a.P, V = a.V;
    const bool live = p < P;
    const int it1 = s_it[0], step = s_it[1];
    // s_hyp: step sizes xyz / scaling / rotation / opacity, sqrt(bias_correction2), spare.  Thread 0 forms them (a few
    // double divisions and a square root) while everybody else already gathers gradients, slots, moments and parameters:
    // the barrier that publishes s_hyp comes only right before the update, so those loads and the scalar work overlap.
    if (p == 0 && !step_sizes_ready) adam_step_sizes(a, s_d, s_hyp);
    // limb-symmetry loss gradient: L = lambda * (| |la| - |ra| | + | |ll| - |rl| |)  (loss_utils.py:226-250);
    // every view's loss contains it, so every slot carries it (train.py:150-152,175)
    float gc[3] = { 0, 0, 0 };
    if (live && a.lambda_consistency != 0.0f && a.limb[0] >= 0) {
        float len[4], dir[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i0 = a.limb[2 * k], i1 = a.limb[2 * k + 1];
            float d[3] = { s_xyz[3 * i0] - s_xyz[3 * i1], s_xyz[3 * i0 + 1] - s_xyz[3 * i1 + 1], s_xyz[3 * i0 + 2] - s_xyz[3 * i1 + 2] };
            len[k] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
            for (int c = 0; c < 3; c++) dir[k][c] = len[k] > 0.0f ? d[c] / len[k] : 0.0f;
        }
        // d|x|/dx = sign(x) (torch.norm of a scalar; 0 at 0)
        const float sa = (len[0] - len[1]) > 0.0f ? 1.0f : ((len[0] - len[1]) < 0.0f ? -1.0f : 0.0f);
        const float sl = (len[2] - len[3]) > 0.0f ? 1.0f : ((len[2] - len[3]) < 0.0f ? -1.0f : 0.0f);
        const float coef[4] = { sa, -sa, sl, -sl };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float w = a.lambda_consistency * coef[k];
            if (p == a.limb[2 * k])
                for (int c = 0; c < 3; c++) gc[c] += w * dir[k][c];
            if (p == a.limb[2 * k + 1])
                for (int c = 0; c < 3; c++) gc[c] -= w * dir[k][c];
        }
    }
    // slots of the views rendered in this group get this group's gradient (others keep what they had: quirk Q8)
    float g11[11], m[11], vv[11], np[11];
#pragma unroll
    for (int c = 0; c < 11; c++) { g11[c] = 0.0f; m[c] = 0.0f; vv[c] = 0.0f; np[c] = 0.0f; }
    const int pl = live ? p : 0;
    float* mp = a.m + (size_t)pl * 11;
    float* vp = a.vv + (size_t)pl * 11;
    // With many views (Panoptic: 31) one thread per joint walking its V slots is a chain of V dependent round trips
    // (21 us for 31 views): all threads update the (view, joint) slots side by side and park them in LDS, then the
    // joint's thread adds them up in view order as before (same values, same order: bit-identical).
    constexpr int SLOT_LDS = 8192;
    __shared__ float s_gc[256 * 3];
    __shared__ float s_slot[SLOT_LDS];
    const bool wide = V > 8 && V * P * 3 <= SLOT_LDS;   // (block-uniform)
    if (wide) {
        if (live) { s_gc[3 * p] = gc[0]; s_gc[3 * p + 1] = gc[1]; s_gc[3 * p + 2] = gc[2]; }
        __syncthreads();
        for (int i = p; i < V * P; i += (int)blockDim.x) {
            const int v = i / P, pp = i - v * P;
            float* sl = a.slots + (size_t)i * 3;
            float val[3];
            if ((a.group_mask >> v) & 1ull) {
                const float* gr = grad_rows(a, v) + (size_t)pp * 11;
#pragma unroll
                for (int c = 0; c < 3; c++) { val[c] = gr[c] + s_gc[3 * pp + c]; sl[c] = val[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) val[c] = sl[c];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) s_slot[3 * i + c] = val[c];
        }
        __syncthreads();
    }
    if (live) {
        for (int v = 0; v < V; v++) {
            if (wide) {
#pragma unroll
                for (int c = 0; c < 3; c++) g11[c] += s_slot[3 * (v * P + p) + c];
                continue;
            }
            float* sl = a.slots + ((size_t)v * P + p) * 3;
            if ((a.group_mask >> v) & 1ull) {
                const float* gr = grad_rows(a, v) + (size_t)p * 11;
#pragma unroll
                for (int c = 0; c < 3; c++) sl[c] = gr[c] + gc[c];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) g11[c] += sl[c];   // mean over the V slots, view order (train.py:215-217)
        }
#pragma unroll
        for (int c = 0; c < 3; c++) g11[c] /= (float)V;
        const float* gl = grad_rows(a, a.last_view) + (size_t)p * 11;
#pragma unroll
        for (int c = 3; c < 11; c++) g11[c] = gl[c];
#pragma unroll
        for (int c = 0; c < 11; c++) { m[c] = mp[c]; vv[c] = vp[c]; }
#pragma unroll
        for (int c = 0; c < 3; c++) { np[c] = a.xyz[3 * p + c]; np[3 + c] = a.scaling[3 * p + c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) np[6 + c] = a.rotation[4 * p + c];
        np[10] = a.opacity[p];
    }
    __syncthreads();   // s_hyp is published
    if (!live) return;
    const float bc2s = s_hyp[4];
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
    const float ss[11] = { s_hyp[0], s_hyp[0], s_hyp[0], s_hyp[1], s_hyp[1], s_hyp[1], s_hyp[2], s_hyp[2], s_hyp[2], s_hyp[2], s_hyp[3] };
#pragma unroll
    for (int c = 0; c < 11; c++) {
        adam_update(np[c], g11[c], m[c], vv[c], w1, b2, w2, eps, ss[c], bc2s);
        mp[c] = m[c];
        vp[c] = vv[c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { a.xyz[3 * p + c] = np[c]; a.scaling[3 * p + c] = np[3 + c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) a.rotation[4 * p + c] = np[6 + c];
    a.opacity[p] = np[10];
    if (mirror) {
#pragma unroll
        for (int c = 0; c < 3; c++) { mirror->xyz[3 * p + c] = np[c]; mirror->scaling[3 * p + c] = np[3 + c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) mirror->rotation[4 * p + c] = np[6 + c];
        mirror->opacity[p] = np[10];
    }
    if (p == 0) { a.counters[0] = it1; a.counters[1] = step; }
}


// host: fills AdamArgs from the C ABI's argument lists; returns nullptr or an error text
inline const char* fill_adam_args(AdamArgs& a, int V, int P, const float* grads, float* slots, unsigned long long group_mask,
                                  int last_view, float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg,
                                  float* exp_avg_sq, int* counters, int acc_steps, const double* lr_sched, const double* lrs,
                                  const double* adam, float lambda_consistency, const int* limb, int shard_world = 1)
{
    if (shard_world < 1) return "loop_adam: shard_world must be >= 1";
    a.world = shard_world;
    a.vmax = (V + shard_world - 1) / shard_world;
    a.rank_stride = (long long)a.vmax * P * 11;
    a.es_state = nullptr; a.es_window = 0; a.es_tol = 0.0f; a.es_sums = nullptr; a.es_host_flag = nullptr;
    if (!grads || !slots || !xyz || !scaling || !rotation || !opacity || !exp_avg || !exp_avg_sq || !counters || !lr_sched || !lrs || !adam)
        return "loop_adam: missing pointer";
    if (last_view < 0 || last_view >= V) return "loop_adam: last_view out of range";
    a.V = V; a.P = P; a.grads = grads; a.slots = slots; a.group_mask = group_mask; a.last_view = last_view;
    a.xyz = xyz; a.scaling = scaling; a.rotation = rotation; a.opacity = opacity; a.m = exp_avg; a.vv = exp_avg_sq;
    a.counters = counters; a.acc_steps = acc_steps;
    a.lr_init = lr_sched[0]; a.lr_final = lr_sched[1]; a.lr_delay_mult = lr_sched[2];
    // (log(0) = -inf like np.log in the reference, general_utils.py:66: a schedule with one zero end point is 0 where that end
    // point has weight and NaN where its weight is exactly 0; an all-zero schedule is switched off before it gets here)
    a.log_lr_init = log(lr_sched[0]);
    a.log_lr_final = log(lr_sched[1]);
    a.lr_delay_steps = (int)lr_sched[3]; a.lr_max_steps = (int)lr_sched[4];
    a.lr_scaling = lrs[0]; a.lr_rotation = lrs[1]; a.lr_opacity = lrs[2];
    a.beta1 = adam[0]; a.beta2 = adam[1]; a.eps = adam[2];
    a.lambda_consistency = lambda_consistency;
    for (int i = 0; i < 8; i++) a.limb[i] = limb ? limb[i] : -1;
    if (limb) for (int i = 0; i < 8; i++) if (limb[i] < 0 || limb[i] >= P) return "loop_adam: limb index out of range";
    a.n_joints = P;
    return nullptr;
}

}  // namespace sksloop
