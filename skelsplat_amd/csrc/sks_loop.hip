// sks_loop.hip -- device-side tail of the multi-view loop (reference: train.py:160-222, scene/gaussian_model.py:
// 203-248, utils/general_utils.py:38-71, utils/loss_utils.py:226-250) for MI355X.
//
// After the rasterizer backward the reference runs ~40 tiny tensor ops per iteration (activation Jacobians through
// autograd, the limb-symmetry loss and its gradient, the V-slot mean, torch.optim.Adam, the LR schedule on the host).
// Here that tail is two single-launch kernels with NO host-side state, so a whole accumulation group is a fixed
// launch sequence that can be captured into a hipGraph and replayed:
//   k_loop_pack : (V_local,P) gradients wrt the ACTIVATED tensors  ->  (V_local,P,11) gradients wrt the RAW parameters
//                 (exp / normalize / sigmoid Jacobians), times the per-view 1/N of the masked-L2 loss;
//   k_loop_adam : slot update + mean over the V view slots + limb-symmetry gradient + LR schedule + Adam, in place.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/skelsplat_hip.h"
#include "sks_loop_dev.h"

extern "C" void sks_set_error_(const char* msg);

namespace {

int fail3(int code, const char* msg)
{
    sks_set_error_(msg);
    return code;
}

// (V,P) threads.  g_* are sks_backward's outputs; raw_* the leaf parameters; scale (V) = 1/N_v or NULL.
__global__ void k_loop_pack(int V, int P, const float* __restrict__ g_means, const float* __restrict__ g_scales,
                            const float* __restrict__ g_rots, const float* __restrict__ g_opac,
                            const float* __restrict__ raw_scaling, const float* __restrict__ raw_rotation,
                            const float* __restrict__ raw_opacity, const double* __restrict__ sums /* V x {S,N} or NULL */,
                            float* __restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V * P) return;
    const int v = i / P, p = i - v * P;
    float sc = 1.0f;
    if (sums) {
        const double n = sums[2 * v + 1];
        sc = (float)(1.0 / (n < 1.0 ? 1.0 : n));
    }
    float* o = packed + (size_t)i * 11;
    o[0] = g_means[3 * i] * sc; o[1] = g_means[3 * i + 1] * sc; o[2] = g_means[3 * i + 2] * sc;
    // scaling = exp(raw)  (gaussian_model.py:39,103-104)
#pragma unroll
    for (int k = 0; k < 3; k++) o[3 + k] = g_scales[3 * i + k] * expf(raw_scaling[3 * p + k]) * sc;
    // rotation = raw / max(|raw|, 1e-12)  (torch.nn.functional.normalize, gaussian_model.py:47,107-108)
    float q[4], gq[4];
    float nn = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) { q[k] = raw_rotation[4 * p + k]; gq[k] = g_rots[4 * i + k]; nn += q[k] * q[k]; }
    const float nrm = fmaxf(sqrtf(nn), 1e-12f);
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) { q[k] /= nrm; dot += q[k] * gq[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) o[6 + k] = (gq[k] - q[k] * dot) / nrm * sc;
    // opacity = sigmoid(raw)  (gaussian_model.py:44,126-127)
    const float x = raw_opacity[p];
    const float s = 1.0f / (1.0f + expf(-x));
    o[10] = g_opac[i] * (s * (1.0f - s)) * sc;
}

using sksloop::AdamArgs;

__global__ __launch_bounds__(256) void k_loop_adam(AdamArgs a)
{
    __shared__ float s_xyz[256 * 3];
    __shared__ float s_hyp[6];
    __shared__ double s_d[4];
    __shared__ int s_it[2];
    if (a.es_state) {   // device-side early stopping: the criterion decides what this step is -- or that there is none
        __shared__ unsigned long long s_mask;
        __shared__ int s_cut[3];
        const int p = threadIdx.x;
        if (p < a.P) { s_xyz[3 * p] = a.xyz[3 * p]; s_xyz[3 * p + 1] = a.xyz[3 * p + 1]; s_xyz[3 * p + 2] = a.xyz[3 * p + 2]; }
        __syncthreads();
        if (p == 0) sksloop::early_stop_decide(a, s_xyz, &s_mask, s_cut);
        __syncthreads();
        if (s_cut[2]) return;            // the scene stopped at an earlier launch: parameters, moments, counters stay
        a.group_mask = s_mask; a.last_view = s_cut[0]; a.acc_steps = s_cut[1];
        __syncthreads();
    }
    sksloop::adam_block_begin(a, s_xyz, s_d, s_it);
    __syncthreads();
    sksloop::adam_block_finish(a, s_xyz, s_hyp, s_d, s_it);
}


// torch.optim.Adam's step for up to ADAM_MULTI_MAX parameter tensors in ONE launch (the reference's optimiser has six parameter
// groups, gaussian_model.py:203-218; torch's foreach path is ~10 launches per group).  Block b belongs to the tensor whose block
// range holds it; step sizes and bias corrections come from the host in the precision torch forms them in (Python doubles).
constexpr int ADAM_MULTI_MAX = 8;
struct AdamMultiArgs {
    float* p[ADAM_MULTI_MAX];
    const float* g[ADAM_MULTI_MAX];
    float* m[ADAM_MULTI_MAX];
    float* v[ADAM_MULTI_MAX];
    long long n[ADAM_MULTI_MAX];
    unsigned blk0[ADAM_MULTI_MAX + 1];
    float step_size[ADAM_MULTI_MAX], bc2_sqrt[ADAM_MULTI_MAX];
    float w1, b2, w2, eps;
    int nt;
};
__global__ __launch_bounds__(256) void k_adam_multi(AdamMultiArgs a)
{
    int t = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MULTI_MAX; k++) t = (k < a.nt && blockIdx.x >= a.blk0[k]) ? k : t;
    const long long i = (long long)(blockIdx.x - a.blk0[t]) * 256 + threadIdx.x;
    if (i >= a.n[t]) return;
    float prm = a.p[t][i], m = a.m[t][i], v = a.v[t][i];
    sksloop::adam_update(prm, a.g[t][i], m, v, a.w1, a.b2, a.w2, a.eps, a.step_size[t], a.bc2_sqrt[t]);
    a.p[t][i] = prm; a.m[t][i] = m; a.v[t][i] = v;
}

}  // namespace

extern "C" {

int sks_loop_pack_grads(int V, int P, const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                        const float* dL_dopacity, const float* raw_scaling, const float* raw_rotation,
                        const float* raw_opacity, const double* loss_sums, float* packed, void* stream)
{
    if (V < 1 || P < 1) return fail3(-1, "loop_pack: bad shape");
    if (!dL_dmeans3D || !dL_dscales || !dL_drotations || !dL_dopacity || !raw_scaling || !raw_rotation || !raw_opacity || !packed)
        return fail3(-2, "loop_pack: missing pointer");
    const int n = V * P;
    hipLaunchKernelGGL(k_loop_pack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, V, P, dL_dmeans3D, dL_dscales,
                       dL_drotations, dL_dopacity, raw_scaling, raw_rotation, raw_opacity, loss_sums, packed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail3((int)e, hipGetErrorString(e));
    return 0;
}

int sks_loop_adam_step(int V, int P, const float* grads, float* slots, unsigned long long group_mask, int last_view,
                       float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                       int* counters, int acc_steps, const double* lr_sched /*HOST 5: init, final, delay_mult, delay_steps, max_steps*/,
                       const double* lrs /*HOST 3: scaling, rotation, opacity*/, const double* adam /*HOST 3: beta1, beta2, eps*/,
                       float lambda_consistency, const int* limb /*HOST 8 or NULL*/, int shard_world, void* stream)
{
    if (V < 1 || V > SKS_MAX_VIEWS || P < 1 || P > SKS_SMALL_P) return fail3(-1, "loop_adam: V or P out of range");
    AdamArgs a;
    if (const char* err = sksloop::fill_adam_args(a, V, P, grads, slots, group_mask, last_view, xyz, scaling, rotation, opacity,
                                                  exp_avg, exp_avg_sq, counters, acc_steps, lr_sched, lrs, adam,
                                                  lambda_consistency, limb, shard_world))
        return fail3(-2, err);
    hipLaunchKernelGGL(k_loop_adam, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail3((int)e, hipGetErrorString(e));
    return 0;
}

int sks_loop_adam_step_es(int V, int P, const float* grads, float* slots, unsigned long long group_mask, int last_view,
                          float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                          int* counters, int acc_steps, const double* lr_sched, const double* lrs, const double* adam,
                          float lambda_consistency, const int* limb, int shard_world, const double* loss_sums, int* es_state,
                          int es_window, float es_tolerance, int* es_host_flag, void* stream)
{
    if (V < 1 || V > SKS_MAX_VIEWS || P < 1 || P > SKS_SMALL_P) return fail3(-1, "loop_adam: V or P out of range");
    if (!es_state) return fail3(-2, "loop_adam_es: es_state is required (sks_loop_adam_step is the step without a criterion)");
    if (es_window < 1 || es_window > sksloop::ES_MAX_WINDOW) return fail3(-1, "loop_adam_es: window out of range [1, 16]");
    if (shard_world == 1 && !loss_sums) return fail3(-2, "loop_adam_es: loss_sums is required on one rank");
    AdamArgs a;
    if (const char* err = sksloop::fill_adam_args(a, V, P, grads, slots, group_mask, last_view, xyz, scaling, rotation, opacity,
                                                  exp_avg, exp_avg_sq, counters, acc_steps, lr_sched, lrs, adam,
                                                  lambda_consistency, limb, shard_world))
        return fail3(-2, err);
    a.es_state = es_state; a.es_window = es_window; a.es_tol = es_tolerance; a.es_sums = loss_sums; a.es_host_flag = es_host_flag;
    if (shard_world > 1 && !loss_sums) a.rank_stride = sksloop::es_tail_offset(a.vmax, P) + 4ll * a.vmax;
    hipLaunchKernelGGL(k_loop_adam, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail3((int)e, hipGetErrorString(e));
    return 0;
}

int sks_adam_multi(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long long* numel, const double* lr, const long long* step, double beta1, double beta2, double eps, void* stream)
{
    if (n_tensors < 1 || n_tensors > ADAM_MULTI_MAX) return fail3(-1, "adam_multi: 1 .. 8 tensors per call");
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step) return fail3(-2, "adam_multi: missing pointer");
    AdamMultiArgs a;
    a.nt = n_tensors;
    unsigned blocks = 0;
    for (int t = 0; t < ADAM_MULTI_MAX; t++) {
        const bool on = t < n_tensors;
        if (on && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t])) return fail3(-2, "adam_multi: missing tensor");
        if (on && (numel[t] < 0 || step[t] < 1)) return fail3(-1, "adam_multi: numel >= 0 and step >= 1 (the count after this step)");
        a.p[t] = on ? params[t] : nullptr; a.g[t] = on ? grads[t] : nullptr; a.m[t] = on ? exp_avg[t] : nullptr; a.v[t] = on ? exp_avg_sq[t] : nullptr;
        a.n[t] = on ? numel[t] : 0;
        a.blk0[t] = blocks;
        if (on) {
            const long long nb = (numel[t] + 255) / 256;
            if (nb > 0x7fffffffll - (long long)blocks) return fail3(-1, "adam_multi: too many elements for one launch");
            blocks += (unsigned)nb;
            // torch/optim/adam.py (_single_tensor_adam, not capturable): bias corrections and the step size as Python floats
            const double bc1 = 1.0 - pow(beta1, (double)step[t]), bc2 = 1.0 - pow(beta2, (double)step[t]);
            a.step_size[t] = (float)(lr[t] / bc1);
            a.bc2_sqrt[t] = (float)sqrt(bc2);
        } else {
            a.step_size[t] = 0.0f; a.bc2_sqrt[t] = 1.0f;
        }
    }
    a.blk0[ADAM_MULTI_MAX] = blocks;
    a.w1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.w2 = (float)(1.0 - beta2); a.eps = (float)eps;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail3((int)e, hipGetErrorString(e));
    return 0;
}

size_t sks_loop_shard_floats(int V, int P, int shard_world)
{
    if (V < 1 || P < 1 || shard_world < 1) return 0;
    const int vmax = (V + shard_world - 1) / shard_world;
    return (size_t)(sksloop::es_tail_offset(vmax, P) + 4ll * vmax);
}

}  // extern "C"
