// Fused Adam / AdamW step over ONE flat fp32 parameter buffer.
//
// The reference optimises with torch.optim.Adam(lr=1e-3, weight_decay=0) — AdamW with parameter
// groups when weight_decay > 0 (multi_part_assembly/models/modules/base_model.py:389-406) — which
// PyTorch runs as several elementwise passes per tensor.  All parameters, gradients and both
// moments of the model live in four flat buffers here (multi_part_assembly_amd/optim.py), so one
// step is ONE streaming kernel: 16 B read + 12 B written per element, float4-vectorised.
// `grad_scale` folds the 1/world_size of the data-parallel gradient mean into the same pass.
// mpa_adam_step_dev is the graph-capturable twin: learning rate, grad_scale, the STEP COUNTER and the bias
// corrections live in an 8-float DEVICE buffer; a one-thread kernel in front of the update advances the counter and
// recomputes the corrections, so a captured launch stays valid across replays and the host never has to upload
// per-step scalars (a re-used pinned staging buffer would be overwritten by a host that runs steps ahead of the GPU).
// `decay_mask` (nullable): 1/0 per element — the reference exempts biases and normalisation weights from weight
// decay (utils/utils.py:90-125 filter_wd_parameters).  mpa_grad_clip_coef: global-norm gradient clipping
// (Lightning's gradient_clip_val, scripts/train.py:90) as a coefficient folded into the same update.
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

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs a, float wd) {
  g = g * a.grad_scale;
  if (wd != 0.0f) {
    if (a.decoupled) p = p * (1.0f - a.lr * wd);
    else g = g + wd * p;
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
                                                        const float* __restrict__ dev_hyper,
                                                        const float* __restrict__ decay_mask) {
  if (dev_hyper != nullptr) {  // {lr, bc1, bc2_sqrt, grad_scale, step bits, clip coefficient}
    a.lr = dev_hyper[0];
    a.bc1 = dev_hyper[1];
    a.bc2_sqrt = dev_hyper[2];
    a.grad_scale = dev_hyper[3] * dev_hyper[5];
  }
  const bool masked = decay_mask != nullptr && a.weight_decay != 0.0f;
  const long long n4 = n / 4;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(param)[i];
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    float4 m = reinterpret_cast<float4*>(exp_avg)[i];
    float4 v = reinterpret_cast<float4*>(exp_avg_sq)[i];
    float4 w = make_float4(a.weight_decay, a.weight_decay, a.weight_decay, a.weight_decay);
    if (masked) {
      const float4 k = reinterpret_cast<const float4*>(decay_mask)[i];
      w = make_float4(w.x * k.x, w.y * k.y, w.z * k.z, w.w * k.w);
    }
    adam_one(p.x, g.x, m.x, v.x, a, w.x);
    adam_one(p.y, g.y, m.y, v.y, a, w.y);
    adam_one(p.z, g.z, m.z, v.z, a, w.z);
    adam_one(p.w, g.w, m.w, v.w, a, w.w);
    reinterpret_cast<float4*>(param)[i] = p;
    reinterpret_cast<float4*>(exp_avg)[i] = m;
    reinterpret_cast<float4*>(exp_avg_sq)[i] = v;
  }
  // tail (< 4 elements)
  const long long t = n4 * 4 + (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t < n) adam_one(param[t], grad[t], exp_avg[t], exp_avg_sq[t], a, masked ? a.weight_decay * decay_mask[t] : a.weight_decay);
}

// hyper[4] holds the step count (int32 bits): advance it and recompute the bias corrections, in double like the host
__global__ void adam_advance_kernel(float* __restrict__ hyper, float beta1, float beta2) {
  const int step = __float_as_int(hyper[4]) + 1;
  hyper[4] = __int_as_float(step);
  hyper[1] = (float)(1.0 - pow((double)beta1, (double)step));
  hyper[2] = (float)sqrt(1.0 - pow((double)beta2, (double)step));
}

// sum of squares of the gradient: per-block partials in double, then ONE block adds them in a fixed order
constexpr int kNormBlocks = 512;
__global__ __launch_bounds__(kThreads) void grad_sqnorm_kernel(const float* __restrict__ grad, long long n,
                                                               double* __restrict__ partial) {
  __shared__ double red[kThreads];
  double acc = 0.0;
  const long long n4 = n / 4, stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    acc += (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z + (double)g.w * g.w;
  }
  const long long t = n4 * 4 + (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t < n) acc += (double)grad[t] * grad[t];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void grad_clip_coef_kernel(const double* __restrict__ partial, int blocks, float max_norm,
                                      const float* __restrict__ scale_dev, float scale, float* __restrict__ coef) {
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += partial[b];
  const double k = scale_dev != nullptr ? (double)scale_dev[0] : (double)scale;
  const double norm = __builtin_sqrt(s) * k;                       // norm of the (already scaled) mean gradient
  const double c = (double)max_norm / (norm + 1e-6);               // torch.nn.utils.clip_grad_norm_
  coef[0] = c < 1.0 ? (float)c : 1.0f;
  coef[1] = (float)norm;
}

}  // namespace

extern "C" int mpa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             int64_t numel, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int decoupled_weight_decay, int64_t step,
                             float grad_scale, const float* decay_mask, void* stream) {
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
                     param, grad, exp_avg, exp_avg_sq, (long long)numel, a, (const float*)nullptr, decay_mask);
  return mpa::check_launch("adam_step");
}

extern "C" int mpa_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                 int64_t numel, float* hyper, float beta1, float beta2, float eps,
                                 float weight_decay, int decoupled_weight_decay, const float* decay_mask,
                                 void* stream) {
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
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, mpa::as_stream(stream), hyper, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, mpa::as_stream(stream),
                     param, grad, exp_avg, exp_avg_sq, (long long)numel, a, (const float*)hyper, decay_mask);
  return mpa::check_launch("adam_step_dev");
}

extern "C" int mpa_grad_clip_workspace(int64_t* bytes) {
  MPA_REQUIRE(bytes != nullptr, "grad_clip_workspace: null pointer");
  *bytes = (int64_t)kNormBlocks * 8;
  return MPA_OK;
}

extern "C" int mpa_grad_clip_coef(const float* grad, int64_t numel, float max_norm, const float* grad_scale_dev,
                                  float grad_scale, void* ws, float* coef, void* stream) {
  MPA_REQUIRE(numel >= 0 && max_norm > 0.0f, "grad_clip_coef: bad numel / max_norm");
  MPA_REQUIRE(grad && ws && coef, "grad_clip_coef: null pointer");
  MPA_REQUIRE((uintptr_t)grad % 16 == 0 && (uintptr_t)ws % 8 == 0, "grad_clip_coef: misaligned buffer");
  hipStream_t s = mpa::as_stream(stream);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(kNormBlocks), dim3(kThreads), 0, s, grad, (long long)numel, partial);
  hipLaunchKernelGGL(grad_clip_coef_kernel, dim3(1), dim3(1), 0, s, (const double*)partial, kNormBlocks, max_norm,
                     grad_scale_dev, grad_scale, coef);
  return mpa::check_launch("grad_clip_coef");
}
