// Fused Adam / AdamW step over ONE flat fp32 parameter buffer.
//
// The reference optimises with torch.optim.Adam(lr=1e-3, weight_decay=0) — AdamW with parameter
// groups when weight_decay > 0 (multi_part_assembly/models/modules/base_model.py:389-406) — which
// PyTorch runs as several elementwise passes per tensor.  All parameters, gradients and both
// moments of the model live in four flat buffers here (multi_part_assembly_amd/optim.py), so one
// step is ONE streaming kernel: 16 B read + 12 B written per element, float4-vectorised.
// `grad_scale` folds the 1/world_size of the data-parallel gradient mean into the same pass.
// mpa_adam_step_dev is the graph-capturable twin: learning rate, bias corrections and grad_scale are read
// from a 4-float DEVICE buffer, so a captured launch stays valid while the host advances the schedule.
//
// Update rule = torch.optim.Adam's (single-tensor path), term by term:
//   g  = grad*grad_scale (+ wd*p for Adam's L2 form);   p *= 1 - lr*wd  for AdamW's decoupled form
//   m  = m + (g - m)*(1-beta1);   v = v*beta2 + g*g*(1-beta2)
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale;
  int decoupled;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs a) {
  g = g * a.grad_scale;
  if (a.weight_decay != 0.0f) {
    if (a.decoupled) p = p * (1.0f - a.lr * a.weight_decay);
    else g = g + a.weight_decay * p;
  }
  m = m + (g - m) * (1.0f - a.beta1);
  v = v * a.beta2 + (g * g) * (1.0f - a.beta2);
  const float denom = __builtin_sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ param,
                                                        const float* __restrict__ grad,
                                                        float* __restrict__ exp_avg,
                                                        float* __restrict__ exp_avg_sq,
                                                        long long n, AdamArgs a,
                                                        const float* __restrict__ dev_hyper) {
  if (dev_hyper != nullptr) {  // {lr, bc1, bc2_sqrt, grad_scale} refreshed by the host between replays
    a.lr = dev_hyper[0];
    a.bc1 = dev_hyper[1];
    a.bc2_sqrt = dev_hyper[2];
    a.grad_scale = dev_hyper[3];
  }
  const long long n4 = n / 4;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(param)[i];
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    float4 m = reinterpret_cast<float4*>(exp_avg)[i];
    float4 v = reinterpret_cast<float4*>(exp_avg_sq)[i];
    adam_one(p.x, g.x, m.x, v.x, a);
    adam_one(p.y, g.y, m.y, v.y, a);
    adam_one(p.z, g.z, m.z, v.z, a);
    adam_one(p.w, g.w, m.w, v.w, a);
    reinterpret_cast<float4*>(param)[i] = p;
    reinterpret_cast<float4*>(exp_avg)[i] = m;
    reinterpret_cast<float4*>(exp_avg_sq)[i] = v;
  }
  // tail (< 4 elements)
  const long long t = n4 * 4 + (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t < n) adam_one(param[t], grad[t], exp_avg[t], exp_avg_sq[t], a);
}

}  // namespace

extern "C" int mpa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             int64_t numel, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int decoupled_weight_decay, int64_t step,
                             float grad_scale, void* stream) {
  MPA_REQUIRE(numel >= 0 && step >= 1, "adam_step: bad numel/step");
  if (numel == 0) return MPA_OK;
  MPA_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
  MPA_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "adam_step: buffers must be 16-byte aligned");
  AdamArgs a;
  a.lr = lr;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.decoupled = decoupled_weight_decay;
  a.grad_scale = grad_scale;
  // bias corrections in double, as torch computes them on the host
  a.bc1 = (float)(1.0 - __builtin_pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)__builtin_sqrt(1.0 - __builtin_pow((double)beta2, (double)step));
  long long blocks = (numel / 4 + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, mpa::as_stream(stream),
                     param, grad, exp_avg, exp_avg_sq, (long long)numel, a, (const float*)nullptr);
  return mpa::check_launch("adam_step");
}

extern "C" int mpa_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                 int64_t numel, const float* hyper, float beta1, float beta2, float eps,
                                 float weight_decay, int decoupled_weight_decay, void* stream) {
  MPA_REQUIRE(numel >= 0, "adam_step_dev: bad numel");
  if (numel == 0) return MPA_OK;
  MPA_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper, "adam_step_dev: null pointer");
  MPA_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "adam_step_dev: buffers must be 16-byte aligned");
  AdamArgs a;
  a.lr = a.bc1 = a.bc2_sqrt = a.grad_scale = 0.0f;  // taken from `hyper` on the device
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.decoupled = decoupled_weight_decay;
  long long blocks = (numel / 4 + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, mpa::as_stream(stream),
                     param, grad, exp_avg, exp_avg_sq, (long long)numel, a, hyper);
  return mpa::check_launch("adam_step_dev");
}
