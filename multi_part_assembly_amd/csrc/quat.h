// Quaternion rotation with pytorch3d's operation order (see pose.hip for the arithmetic contract).
#pragma once

#include <hip/hip_runtime.h>

namespace mpa {

struct Quat {
  float w, x, y, z;
};

__device__ __forceinline__ Quat quat_raw_mul(const Quat a, const Quat b) {
  Quat o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return o;
}

// v' = (q (0,v) q*)[1:], no normalisation of q
__device__ __forceinline__ void quat_rotate(const Quat q, float px, float py, float pz, float& ox, float& oy,
                                            float& oz) {
  const Quat p{0.0f, px, py, pz};
  const Quat c{q.w * 1.0f, q.x * -1.0f, q.y * -1.0f, q.z * -1.0f};
  const Quat r = quat_raw_mul(quat_raw_mul(q, p), c);
  ox = r.x;
  oy = r.y;
  oz = r.z;
}

}  // namespace mpa
