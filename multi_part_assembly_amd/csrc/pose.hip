// Rigid pose application (quaternion rotate [+ translate]) to part point clouds, forward and backward.
//
// Replaces the tensor-op chain behind rot_pc / transform_pc of the reference
// (multi_part_assembly/utils/transforms.py:75-109,199-244), which repeat_interleaves q and t to
// [B,P,N,4]/[B,P,N,3] and calls pytorch3d's quaternion_apply — two Hamilton products over
// materialised [B,P,N,4] tensors.  Here one thread handles one point with the part's quaternion and
// translation held in scalar registers; nothing is materialised but the output.
//
// Arithmetic is pinned to pytorch3d's published definition, operation by operation (this file is
// built with -ffp-contract=off), so that the transformed clouds — and therefore the Chamfer
// arg-mins computed on them — are bit-identical to the reference CPU path:
//     r   = raw_multiply(q, (0, p))          ow = aw*bw - ax*bx - ay*by - az*bz, ... left to right
//     out = raw_multiply(r, q * (1,-1,-1,-1))[1:]  (+ t)
// No normalisation of q (a non-unit quaternion scales by |q|^2, as in the reference).
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct Quat {
  float w, x, y, z;
};

__device__ __forceinline__ Quat raw_mul(const Quat a, const Quat b) {
  Quat o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return o;
}

__device__ __forceinline__ void quat_apply(const Quat q, float px, float py, float pz, float& ox,
                                           float& oy, float& oz) {
  const Quat p{0.0f, px, py, pz};
  const Quat c{q.w * 1.0f, q.x * -1.0f, q.y * -1.0f, q.z * -1.0f};
  const Quat r = raw_mul(raw_mul(q, p), c);
  ox = r.x;
  oy = r.y;
  oz = r.z;
}

// grid = (ceil(N / kThreads), M); one part per blockIdx.y so q/t/mask are wave-uniform.
__global__ __launch_bounds__(kThreads) void pose_apply_kernel(
    const float* __restrict__ pc, const float* __restrict__ quat, const float* __restrict__ trans,
    const float* __restrict__ mask, float fill, int n_points, float* __restrict__ out) {
  const int m = blockIdx.y;
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= n_points) return;
  const Quat q{quat[4 * m + 0], quat[4 * m + 1], quat[4 * m + 2], quat[4 * m + 3]};
  const long long o = 3 * ((long long)m * n_points + n);
  float px, py, pz;
  if (mask != nullptr && mask[m] == 0.0f) {
    px = py = pz = fill;  // masked_fill(valid == 0, fill) BEFORE the transform (loss.py:173-175)
  } else {
    px = pc[o + 0];
    py = pc[o + 1];
    pz = pc[o + 2];
  }
  float ox, oy, oz;
  quat_apply(q, px, py, pz, ox, oy, oz);
  if (trans != nullptr) {
    ox = ox + trans[3 * m + 0];
    oy = oy + trans[3 * m + 1];
    oz = oz + trans[3 * m + 2];
  }
  out[o + 0] = ox;
  out[o + 1] = oy;
  out[o + 2] = oz;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Backward: out = R(q) p + t with R(q) p = (w^2 - |u|^2) p + 2 (u.p) u + 2 w (u x p), q = (w, u).
//   grad_t  = sum_n g_n
//   grad_w  = sum_n g_n . (2 w p_n + 2 u x p_n)
//   grad_u  = sum_n  -2 (g_n.p_n) u + 2 (g_n.u) p_n + 2 (u.p_n) g_n + 2 w (p_n x g_n)
//   grad_pc = R(q)^T g = the same polynomial with u -> -u   (optional; zero where masked)
// One block per part; fixed-shape tree reduction (deterministic).
__global__ __launch_bounds__(kThreads) void pose_grad_kernel(
    const float* __restrict__ gout, const float* __restrict__ pc, const float* __restrict__ quat,
    const float* __restrict__ mask, float fill, int n_points, float* __restrict__ gquat,
    float* __restrict__ gtrans, float* __restrict__ gpc) {
  const int m = blockIdx.x;
  const float w = quat[4 * m + 0], ux = quat[4 * m + 1], uy = quat[4 * m + 2], uz = quat[4 * m + 3];
  const bool masked = mask != nullptr && mask[m] == 0.0f;
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};  // gw, gux, guy, guz, gtx, gty, gtz
  for (int n = threadIdx.x; n < n_points; n += kThreads) {
    const long long o = 3 * ((long long)m * n_points + n);
    const float gx = gout[o + 0], gy = gout[o + 1], gz = gout[o + 2];
    float px, py, pz;
    if (masked) {
      px = py = pz = fill;
    } else {
      px = pc[o + 0];
      py = pc[o + 1];
      pz = pc[o + 2];
    }
    const float cx = uy * pz - uz * py, cy = uz * px - ux * pz, cz = ux * py - uy * px;  // u x p
    const float gp = gx * px + gy * py + gz * pz;
    const float gu = gx * ux + gy * uy + gz * uz;
    const float up = ux * px + uy * py + uz * pz;
    const float dx = py * gz - pz * gy, dy = pz * gx - px * gz, dz = px * gy - py * gx;  // p x g
    acc[0] += 2.0f * (w * gp + (gx * cx + gy * cy + gz * cz));
    acc[1] += 2.0f * (-gp * ux + gu * px + up * gx + w * dx);
    acc[2] += 2.0f * (-gp * uy + gu * py + up * gy + w * dy);
    acc[3] += 2.0f * (-gp * uz + gu * pz + up * gz + w * dz);
    acc[4] += gx;
    acc[5] += gy;
    acc[6] += gz;
    if (gpc != nullptr) {
      float rx = 0.0f, ry = 0.0f, rz = 0.0f;
      if (!masked) {
        const float s = w * w - (ux * ux + uy * uy + uz * uz);
        const float ex = uy * gz - uz * gy, ey = uz * gx - ux * gz, ez = ux * gy - uy * gx;  // u x g
        rx = s * gx + 2.0f * gu * ux - 2.0f * w * ex;
        ry = s * gy + 2.0f * gu * uy - 2.0f * w * ey;
        rz = s * gz + 2.0f * gu * uz - 2.0f * w * ez;
      }
      gpc[o + 0] = rx;
      gpc[o + 1] = ry;
      gpc[o + 2] = rz;
    }
  }
  __shared__ float red[kThreads / 64][7];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < kThreads / 64; ++v) s += red[v][threadIdx.x];
    if (threadIdx.x < 4) gquat[4 * m + threadIdx.x] = s;
    else if (gtrans != nullptr) gtrans[3 * m + (threadIdx.x - 4)] = s;
  }
}

// Rotation3D's constructor rule (utils/rotation.py:115-126 upstream): quaternions of norm <= 0.5 (zero padding) become
// the identity.  One launch instead of norm + compare + where; keep [count] (0/1) gates the gradient.
__global__ void quat_sanitize_kernel(const float* __restrict__ q, long long count, float* __restrict__ out,
                                     float* __restrict__ keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float4 v = reinterpret_cast<const float4*>(q)[i];
  // torch.norm(p=2): sqrt of the sum of squares
  const bool ok = __builtin_sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w) > 0.5f;
  reinterpret_cast<float4*>(out)[i] = ok ? v : make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  keep[i] = ok ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int mpa_quat_sanitize(const float* quat, int64_t count, float* out, float* keep, void* stream) {
  MPA_REQUIRE(count >= 0, "quat_sanitize: negative size");
  if (count == 0) return MPA_OK;
  MPA_REQUIRE(quat && out && keep, "quat_sanitize: null pointer");
  hipLaunchKernelGGL(quat_sanitize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, mpa::as_stream(stream),
                     quat, (long long)count, out, keep);
  return mpa::check_launch("quat_sanitize");
}

// ---- the weighted batch mean of the loss terms (base_model.py:348-387 with one sample: loss = sum_k w_k mean_b t_kb) ----------
// One block; wave w reduces terms k = w, w + 4, ... over the batch (lane-strided partial sums, then the fixed xor tree),
// thread 0 adds the weighted means in term order.
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ terms, const float* __restrict__ w, int K,
                                                          int B, float* __restrict__ means, float* __restrict__ loss) {
  __shared__ float sm[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < K; k += 4) {
    float s = 0.0f;
    for (int b = lane; b < B; b += 64) s += terms[(long long)k * B + b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      const float m = s / (float)B;
      sm[k] = m;
      means[k] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int k = 0; k < K; ++k) t += w[k] * sm[k];
    loss[0] = t;
  }
}

// d terms[k][b] = (g_loss w_k + g_means[k]) / B  (either incoming gradient may be absent)
__global__ __launch_bounds__(256) void loss_reduce_bwd_kernel(const float* __restrict__ g_loss, const float* __restrict__ g_means,
                                                              const float* __restrict__ w, int K, int B,
                                                              float* __restrict__ g_terms) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)K * B) return;
  const int k = (int)(i / B);
  float g = g_loss != nullptr ? g_loss[0] * w[k] : 0.0f;
  if (g_means != nullptr) g += g_means[k];
  g_terms[i] = g / (float)B;
}

extern "C" int mpa_loss_reduce_forward(const float* terms, const float* weights, int64_t K, int64_t B, float* means,
                                       float* loss, void* stream) {
  MPA_REQUIRE(K >= 1 && K <= 64 && B >= 1 && B < (1LL << 31), "loss_reduce_forward: need 1 <= K <= 64 terms and B >= 1");
  MPA_REQUIRE(terms && weights && means && loss, "loss_reduce_forward: null pointer");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, mpa::as_stream(stream), terms, weights, (int)K, (int)B, means,
                     loss);
  return mpa::check_launch("loss_reduce_forward");
}

extern "C" int mpa_loss_reduce_backward(const float* grad_loss, const float* grad_means, const float* weights, int64_t K,
                                        int64_t B, float* grad_terms, void* stream) {
  MPA_REQUIRE(K >= 1 && K <= 64 && B >= 1 && B < (1LL << 31), "loss_reduce_backward: need 1 <= K <= 64 terms and B >= 1");
  MPA_REQUIRE(weights && grad_terms, "loss_reduce_backward: null pointer");
  hipLaunchKernelGGL(loss_reduce_bwd_kernel, dim3((unsigned)((K * B + 255) / 256)), dim3(256), 0, mpa::as_stream(stream),
                     grad_loss, grad_means, weights, (int)K, (int)B, grad_terms);
  return mpa::check_launch("loss_reduce_backward");
}

extern "C" int mpa_pose_apply_forward(const float* pc, const float* quat, const float* trans,
                                      const float* mask, float fill, int64_t num_parts,
                                      int64_t num_points, float* out, void* stream) {
  MPA_REQUIRE(num_parts >= 0 && num_points >= 0, "pose_apply_forward: negative size");
  if (num_parts == 0 || num_points == 0) return MPA_OK;
  MPA_REQUIRE(pc && quat && out, "pose_apply_forward: null pointer");
  MPA_REQUIRE(num_parts <= 65535 && num_points < (1LL << 31), "pose_apply_forward: size too large");
  dim3 grid((unsigned)((num_points + kThreads - 1) / kThreads), (unsigned)num_parts, 1);
  hipLaunchKernelGGL(pose_apply_kernel, grid, dim3(kThreads), 0, mpa::as_stream(stream), pc, quat,
                     trans, mask, fill, (int)num_points, out);
  return mpa::check_launch("pose_apply_forward");
}

extern "C" int mpa_pose_apply_backward(const float* grad_out, const float* pc, const float* quat,
                                       const float* mask, float fill, int64_t num_parts,
                                       int64_t num_points, float* grad_quat, float* grad_trans,
                                       float* grad_pc, void* stream) {
  MPA_REQUIRE(num_parts >= 0 && num_points >= 0, "pose_apply_backward: negative size");
  if (num_parts == 0) return MPA_OK;
  MPA_REQUIRE(grad_out && pc && quat && grad_quat, "pose_apply_backward: null pointer");
  MPA_REQUIRE(num_parts < (1LL << 31) && num_points < (1LL << 31), "pose_apply_backward: size too large");
  hipLaunchKernelGGL(pose_grad_kernel, dim3((unsigned)num_parts), dim3(kThreads), 0,
                     mpa::as_stream(stream), grad_out, pc, quat, mask, fill, (int)num_points,
                     grad_quat, grad_trans, grad_pc);
  return mpa::check_launch("pose_apply_backward");
}
