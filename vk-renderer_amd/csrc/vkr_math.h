// Host-side matrix helpers (binary32, column-major m[col*4+row]) for the matrices the caller of the
// two passes builds every frame: projection, view, their product and its inverse
// (/root/reference/src/main.cpp:95,206-209; /root/reference/inc/vkcamera.hpp:51-53 use glm for
// these), plus the clip-plane extraction the cull pass consumes.
#pragma once

#include <cmath>

namespace vkrm {

// Right-handed perspective with clip-space depth in [-1, 1] (the convention the reference gets from
// glm with no GLM_FORCE_* macro defined).
inline void perspective(float fovy, float aspect, float z_near, float z_far, float m[16]) {
  const float t = std::tan(fovy / 2.0f);
  for (int i = 0; i < 16; ++i) m[i] = 0.0f;
  m[0] = 1.0f / (aspect * t);
  m[5] = 1.0f / t;
  m[10] = -(z_far + z_near) / (z_far - z_near);
  m[11] = -1.0f;
  m[14] = -(2.0f * z_far * z_near) / (z_far - z_near);
}

inline void look_at(const float eye[3], const float center[3], const float up[3], float m[16]) {
  float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
  const float fl = 1.0f / std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (float& c : f) c *= fl;
  float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
  const float sl = 1.0f / std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (float& c : s) c *= sl;
  const float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
  m[0] = s[0];  m[4] = s[1];  m[8] = s[2];
  m[1] = u[0];  m[5] = u[1];  m[9] = u[2];
  m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
  m[3] = 0.0f;  m[7] = 0.0f;  m[11] = 0.0f;
  m[12] = -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]);
  m[13] = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
  m[14] = f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2];
  m[15] = 1.0f;
}

inline void mul(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += a[k * 4 + row] * b[c * 4 + k];
      r[c * 4 + row] = acc;
    }
  for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// Cofactor expansion through 2x2 sub-determinants of the top and bottom row pairs.
inline bool inverse(const float m[16], float out[16]) {
  const float a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3];
  const float a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
  const float a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11];
  const float a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
  const float b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
  const float b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
  const float b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
  const float b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
  const float det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
  if (det == 0.0f || !std::isfinite(det)) return false;
  const float id = 1.0f / det;
  float r[16];
  r[0] = (a11 * b11 - a12 * b10 + a13 * b09) * id;
  r[1] = (a02 * b10 - a01 * b11 - a03 * b09) * id;
  r[2] = (a31 * b05 - a32 * b04 + a33 * b03) * id;
  r[3] = (a22 * b04 - a21 * b05 - a23 * b03) * id;
  r[4] = (a12 * b08 - a10 * b11 - a13 * b07) * id;
  r[5] = (a00 * b11 - a02 * b08 + a03 * b07) * id;
  r[6] = (a32 * b02 - a30 * b05 - a33 * b01) * id;
  r[7] = (a20 * b05 - a22 * b02 + a23 * b01) * id;
  r[8] = (a10 * b10 - a11 * b08 + a13 * b06) * id;
  r[9] = (a01 * b08 - a00 * b10 - a03 * b06) * id;
  r[10] = (a30 * b04 - a31 * b02 + a33 * b00) * id;
  r[11] = (a21 * b02 - a20 * b04 - a23 * b00) * id;
  r[12] = (a11 * b07 - a10 * b09 - a12 * b06) * id;
  r[13] = (a00 * b09 - a01 * b07 + a02 * b06) * id;
  r[14] = (a31 * b01 - a30 * b03 - a32 * b00) * id;
  r[15] = (a20 * b03 - a21 * b01 + a22 * b00) * id;
  for (int i = 0; i < 16; ++i) out[i] = r[i];
  return true;
}

// Clip planes of clip = VP * world under Vulkan rules (-w<=x<=w, -w<=y<=w, 0<=z<=w).  Row r of VP is
// (m[r], m[4+r], m[8+r], m[12+r]).  A point is inside plane k iff dot(plane_k.xyz, p) + plane_k.w >= 0.
// Planes are left un-normalised: only additions of matrix rows, so CPU and GPU callers derive
// identical bits.
inline void frustum_planes(const float vp[16], float planes[24]) {
  float row[4][4];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) row[r][c] = vp[c * 4 + r];
  for (int c = 0; c < 4; ++c) {
    planes[0 + c] = row[3][c] + row[0][c];   // left:   x >= -w
    planes[4 + c] = row[3][c] - row[0][c];   // right:  x <=  w
    planes[8 + c] = row[3][c] + row[1][c];   // bottom: y >= -w
    planes[12 + c] = row[3][c] - row[1][c];  // top:    y <=  w
    planes[16 + c] = row[2][c];              // near:   z >=  0
    planes[20 + c] = row[3][c] - row[2][c];  // far:    z <=  w
  }
}

}  // namespace vkrm
