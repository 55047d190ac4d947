// ppo_tile.h -- device pieces shared by the PPO kernels (ppo_kernels.hip) and the fused rollout kernel of the env engine
// (sumo_engine.hip): the flat parameter layout of the MLP(64,64) policy / value nets (reference model.py:153-177 order) and the
// one-wave, one-16-row-tile trunk forward + Gaussian head on v_mfma_f32_16x16x4_f32.  Every kernel that evaluates a net goes
// through these functions, so their outputs agree bit for bit (tests: test_selfplay_forward_equals_separate_evaluations,
// test_rollout_kernel_matches_stepwise_path).
#ifndef PPO_TILE_H
#define PPO_TILE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_H 64           /* hidden units (PPO_HIDDEN) */
#define PT_HS 66          /* LDS row stride of a 16 x 64 activation tile (== 2 mod 32) */
#define PT_MAXA 16        /* action dims are padded to one 16-column MFMA tile */
#define PT_LOG2PI_F 1.8378770664093453f
#define PT_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct ParamLayout {  // offsets into the flat parameter vector (checkpoint order)
  int D, A, P;
  int pi_w0, pi_b0, pi_w1, pi_b1, vf_w0, vf_b0, vf_w1, vf_b1, pi_w, pi_b, logstd, vf_w, vf_b;
};
static ParamLayout make_layout(int D, int A) {
  ParamLayout L;
  L.D = D; L.A = A;
  int o = 0;
  L.pi_w0 = o; o += D * PT_H; L.pi_b0 = o; o += PT_H; L.pi_w1 = o; o += PT_H * PT_H; L.pi_b1 = o; o += PT_H;
  L.vf_w0 = o; o += D * PT_H; L.vf_b0 = o; o += PT_H; L.vf_w1 = o; o += PT_H * PT_H; L.vf_b1 = o; o += PT_H;
  L.pi_w = o; o += PT_H * A; L.pi_b = o; o += A; L.logstd = o; o += A; L.vf_w = o; o += PT_H; L.vf_b = o; o += 1;
  L.P = o;
  return L;
}
extern "C" int ppo_param_count(int ob_dim, int ac_dim) { return make_layout(ob_dim, ac_dim).P; }

static int x_stride(int D) {  // LDS row stride of the staged observation tile: >= 16*ceil(D/16), == 2 mod 32
  int cols = ((D + 15) / 16) * 16;
  int s = cols;
  while (s % 32 != 2) s++;
  return s;
}

// ---------------------------------------------------------------------------------------------------------
// shared device pieces: one wave, one 16-row tile, one trunk
// ---------------------------------------------------------------------------------------------------------
struct Net { const float *w0, *b0, *w1, *b1, *w2, *b2; int nout; };

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// forward of one trunk on the staged tile.  h1buf/h2buf [16][PT_HS] receive the relu activations; returns the head tile
// (D layout: lane (i = lane&15, kq = lane>>4) holds rows 4kq+r, column i; columns >= nout are zero + garbage-free).
// ROWS < 16: only the first ROWS rows of the tile carry data (the fused rollout kernel evaluates its env's two observations);
// the buffers then hold ROWS rows, the other rows of the MFMA operands are zero and their results are dropped.  Rows of a
// tile never mix, so a row's result does not depend on ROWS.
template <bool TANH = false, int ROWS = 16>
__device__ __forceinline__ f32x4 trunk_forward(const Net& net, const float* xbuf, int XS, int D, float* h1buf, float* h2buf, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  const int Dp = (D + 3) & ~3;
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) acc[ct] = (f32x4){0, 0, 0, 0};
  // k-steps are issued eight at a time with all of their operand loads in flight first (32 weight loads per batch): the
  // accumulation order is unchanged, but one memory round trip is exposed per batch instead of one per k-step
  for (int k0 = 0; k0 < Dp; k0 += 32) {
    float a[8], b[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int k = k0 + 4 * u + kq;
      const bool ok = k < D;
      a[u] = (ok && (ROWS == 16 || i < ROWS)) ? xbuf[i * XS + k] : 0.0f;
#pragma unroll
      for (int ct = 0; ct < 4; ct++) b[u][ct] = ok ? net.w0[k * PT_H + ct * 16 + i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (k0 + 4 * u < Dp) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[ct] = PT_MFMA(a[u], b[u][ct], acc[ct]);
      }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    float bias = net.b0[ct * 16 + i];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float z = acc[ct][r] + bias;
      if (ROWS == 16 || 4 * kq + r < ROWS) h1buf[(4 * kq + r) * PT_HS + ct * 16 + i] = TANH ? tanhf(z) : fmaxf(z, 0.0f);
    }
    acc[ct] = (f32x4){0, 0, 0, 0};
  }
  wave_sync();
  {
    float b[PT_H / 4][4];
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) b[u][ct] = net.w1[(4 * u + kq) * PT_H + ct * 16 + i];
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) {
      const float a = (ROWS == 16 || i < ROWS) ? h1buf[i * PT_HS + 4 * u + kq] : 0.0f;
#pragma unroll
      for (int ct = 0; ct < 4; ct++) acc[ct] = PT_MFMA(a, b[u][ct], acc[ct]);
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    float bias = net.b1[ct * 16 + i];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float z = acc[ct][r] + bias;
      if (ROWS == 16 || 4 * kq + r < ROWS) h2buf[(4 * kq + r) * PT_HS + ct * 16 + i] = TANH ? tanhf(z) : fmaxf(z, 0.0f);
    }
  }
  wave_sync();
  f32x4 out = (f32x4){0, 0, 0, 0};
  const bool col_ok = i < net.nout;
  {
    float b[PT_H / 4];
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) b[u] = col_ok ? net.w2[(4 * u + kq) * net.nout + i] : 0.0f;
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) out = PT_MFMA((ROWS == 16 || i < ROWS) ? h2buf[i * PT_HS + 4 * u + kq] : 0.0f, b[u], out);
  }
  float bias = col_ok ? net.b2[i] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; r++) out[r] += bias;
  return out;
}

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes that share lane>>4
  v += __shfl_xor(v, 1, PT_WAVE); v += __shfl_xor(v, 2, PT_WAVE); v += __shfl_xor(v, 4, PT_WAVE); v += __shfl_xor(v, 8, PT_WAVE);
  return v;
}
__device__ __forceinline__ float kq_sum(float v) {  // sum over the 4 lanes that share lane&15
  v += __shfl_xor(v, 16, PT_WAVE); v += __shfl_xor(v, 32, PT_WAVE);
  return v;
}

__device__ __forceinline__ Net pi_net(const float* p, const ParamLayout& L) {
  Net n = {p + L.pi_w0, p + L.pi_b0, p + L.pi_w1, p + L.pi_b1, p + L.pi_w, p + L.pi_b, L.A};
  return n;
}
__device__ __forceinline__ Net vf_net(const float* p, const ParamLayout& L) {
  Net n = {p + L.vf_w0, p + L.vf_b0, p + L.vf_w1, p + L.vf_b1, p + L.vf_w, p + L.vf_b, 1};
  return n;
}

// One row of the diagonal-Gaussian head (baselines distributions.py:227-251) in the tile's D layout: lane i < A of the row's
// 16-lane group holds the mean m of action dimension i.  sample: act = m + std * noise, else `act` is given.
// Returns -log pi(act | obs) of the row (every lane of the group).
__device__ __forceinline__ float gauss_row(float m, float std, float sum_logstd, bool ok, bool sample, float noise, float& act, int A) {
  if (sample) act = ok ? m + std * noise : m;
  const float z = ok ? (act - m) / std : 0.0f;
  const float ss = row16_sum(z * z);
  return 0.5f * ss + 0.5f * PT_LOG2PI_F * (float)A + sum_logstd;
}

// Gaussian head on a policy tile (D layout): samples (noise != nullptr) or scores `act`; returns the row's neglogp in
// the lanes with i == 0 via nlp[r]
__device__ __forceinline__ void gauss_head(const float* params, const ParamLayout& L, const f32x4& mean, const float* noise, int r0, int n,
                                           int lane, float (&act)[4], float (&nlp)[4]) {
  const int i = lane & 15, kq = lane >> 4, A = L.A;
  const bool col = i < A;
  const float logstd = col ? params[L.logstd + i] : 0.0f;
  const float std = expf(logstd);
  const float sum_logstd = row16_sum(logstd);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = r0 + 4 * kq + r;
    const bool ok = col && row < n;
    const float nz = (noise && ok) ? noise[(size_t)row * A + i] : 0.0f;
    nlp[r] = gauss_row(mean[r], std, sum_logstd, ok, noise != nullptr, nz, act[r], A);
  }
}

// reward curriculum of the rollout (runner.py:134): alpha * shaping + (1 - alpha) * main, evaluated in float64, stored as float32
__device__ __forceinline__ float reward_mix(double alpha, double shaping, double main_r) {
  return (float)(alpha * shaping + (1 - alpha) * main_r);
}

#endif
