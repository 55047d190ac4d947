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

#ifndef PT_PROBE          /* development hooks inside trunk_forward (the rollout kernel's probe build defines them) */
#define PT_PROBE_INIT() do { } while (0)
#define PT_PROBE(k) do { } while (0)
#endif

// Global-memory pointers.  A pointer the compiler cannot trace to a kernel argument (one read from a struct in memory: a net
// description, the laundered launch arguments of the rollout kernel) is a GENERIC pointer and its loads are flat_load, which
// occupy the LDS counter as well as the vector-memory counter: every wait for an LDS read then also waits for the weight loads in
// flight, and a loop that interleaves both runs one memory round trip per iteration whatever the prefetch depth.  Loads through
// an address_space(1) pointer are global_load (vmcnt only).
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_GAS __attribute__((address_space(1)))
#define PT_CAS __attribute__((address_space(4)))   /* constant memory (the kernel-argument segment): scalar loads */
#else   /* the host pass parses the device functions too, and has no address spaces */
#define PT_GAS
#define PT_CAS
#endif
template <class T> __device__ __forceinline__ const T PT_GAS* pt_global(const T* p) { return (const T PT_GAS*)p; }
template <class T> __device__ __forceinline__ T PT_GAS* pt_global(T* p) { return (T PT_GAS*)p; }
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
struct Net { const float PT_GAS *w0, *b0, *w1, *b1, *w2, *b2; int nout; };

// Ordering point between LDS writes and reads of ONE wave (lanes exchanging a tile through the wave's own LDS region).  The
// hardware executes a wave's LDS instructions in order, so only the compiler has to be held: a wavefront-scope fence.  (A
// workgroup-scope fence also drains the vector-memory counter -- s_waitcnt vmcnt(0) -- i.e. it waits for every global store and
// prefetch in flight, e.g. the rollout records written just before a policy trunk.)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

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
  // the biases of all three layers are requested first: read where they are used, each sat behind its own memory round trip
  // (load, wait, add -- per column tile and layer: the compiler does not move a load across the row-guard branches around the
  // activation stores), which was most of the phase's time in the rollout kernel
  float bias0[4], bias1[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) { bias0[ct] = net.b0[ct * 16 + i]; bias1[ct] = net.b1[ct * 16 + i]; }
  const bool col_ok = i < net.nout;
  const float bias2 = net.b2[col_ok ? i : 0];
  PT_PROBE_INIT();
  // k-steps are issued eight at a time with all of their operand loads in flight first (32 weight loads per batch): the
  // accumulation order is unchanged, but one memory round trip is exposed per batch instead of one per k-step
  for (int k0 = 0; k0 < Dp; k0 += 32) {
    float a[8], b[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int k = k0 + 4 * u + kq;
      const bool ok = k < D;
      const int kc = ok ? k : D - 1;   // unconditional loads from a clamped row, zeroed by a select (a guarded load is a branch)
      const float av = xbuf[(ROWS == 16 ? i : (i < ROWS ? i : 0)) * XS + kc];
      a[u] = (ok && (ROWS == 16 || i < ROWS)) ? av : 0.0f;
#pragma unroll
      for (int ct = 0; ct < 4; ct++) { const float wv = net.w0[kc * PT_H + ct * 16 + i]; b[u][ct] = ok ? wv : 0.0f; }
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (k0 + 4 * u < Dp) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[ct] = PT_MFMA(a[u], b[u][ct], acc[ct]);
      }
  }
  PT_PROBE(0);   // first layer: operand loads + products
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const float bias = bias0[ct];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float z = acc[ct][r] + bias;
      if (ROWS == 16 || 4 * kq + r < ROWS) h1buf[(4 * kq + r) * PT_HS + ct * 16 + i] = TANH ? tanhf(z) : fmaxf(z, 0.0f);
    }
    acc[ct] = (f32x4){0, 0, 0, 0};
  }
  wave_sync();
  PT_PROBE(1);   // bias + activation + LDS write of the first layer
  {
    float b[PT_H / 4][4];
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) b[u][ct] = net.w1[(4 * u + kq) * PT_H + ct * 16 + i];
    float av[PT_H / 4];   // the tile's A operands in one batch of LDS reads
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) { const float v = h1buf[(ROWS == 16 ? i : (i < ROWS ? i : 0)) * PT_HS + 4 * u + kq]; av[u] = (ROWS == 16 || i < ROWS) ? v : 0.0f; }
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++) acc[ct] = PT_MFMA(av[u], b[u][ct], acc[ct]);
    }
  }
  PT_PROBE(2);   // second layer: loads + products
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const float bias = bias1[ct];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float z = acc[ct][r] + bias;
      if (ROWS == 16 || 4 * kq + r < ROWS) h2buf[(4 * kq + r) * PT_HS + ct * 16 + i] = TANH ? tanhf(z) : fmaxf(z, 0.0f);
    }
  }
  wave_sync();
  PT_PROBE(3);   // bias + activation + LDS write of the second layer
  f32x4 out = (f32x4){0, 0, 0, 0};
  {
    float b[PT_H / 4], av[PT_H / 4];
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) { const float wv = net.w2[(4 * u + kq) * net.nout + (col_ok ? i : 0)]; b[u] = col_ok ? wv : 0.0f; }
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) { const float v = h2buf[(ROWS == 16 ? i : (i < ROWS ? i : 0)) * PT_HS + 4 * u + kq]; av[u] = (ROWS == 16 || i < ROWS) ? v : 0.0f; }
#pragma unroll
    for (int u = 0; u < PT_H / 4; u++) out = PT_MFMA(av[u], b[u], out);
  }
  PT_PROBE(4);   // head: loads + products
  float bias = col_ok ? bias2 : 0.0f;
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

template <class PL>   // ParamLayout in any address space (the rollout kernel reads it in place from its kernel-argument segment)
__device__ __forceinline__ Net pi_net(const float* p_, const PL& L) {
  const float PT_GAS* p = pt_global(p_);
  Net n = {p + L.pi_w0, p + L.pi_b0, p + L.pi_w1, p + L.pi_b1, p + L.pi_w, p + L.pi_b, L.A};
  return n;
}
template <class PL>
__device__ __forceinline__ Net vf_net(const float* p_, const PL& L) {
  const float PT_GAS* p = pt_global(p_);
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
__device__ __forceinline__ void gauss_head(const float* params_, const ParamLayout& L, const f32x4& mean, const float* noise_, int r0, int n,
                                           int lane, float (&act)[4], float (&nlp)[4]) {
  const float PT_GAS* params = pt_global(params_);
  const float PT_GAS* noise = pt_global(noise_);
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

// ---------------------------------------------------------------------------------------------------------
// recurrent nets (baselines lstm(nlstm), a2c/utils.py:82-103): pieces shared by ppo_lstm_step_kernel (16-row MFMA tiles,
// ppo_kernels.hip) and the fused rollout kernel (an env's few rows on the vector ALU, sumo_engine.hip)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pt_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct LstmCell { float ig, fg, og, ug, cn, tcn, hn; };
// gate pre-activations (bias not yet added; bf carries the forget-bias offset) and the previous cell -> gates, new cell, latent
__device__ __forceinline__ LstmCell lstm_cell(float zi, float zf, float zo, float zu, float bi, float bf, float bo, float bu, float cp) {
  LstmCell o;
  o.ig = pt_sigmoid(zi + bi); o.fg = pt_sigmoid(zf + bf); o.og = pt_sigmoid(zo + bo); o.ug = tanhf(zu + bu);
  o.cn = __builtin_fmaf(o.fg, cp, o.ig * o.ug);   // one explicit contraction, so every caller rounds alike
  o.tcn = tanhf(o.cn); o.hn = o.og * o.tcn;
  return o;
}

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));   // two floats at dword alignment (rows of a snapshot pool start at odd offsets)

// Gate pre-activations of R rows on the vector ALU, equal bit for bit to the MFMA tiles of ppo_lstm_step_kernel: an f32 MFMA
// adds its four k-products to C as sequential FMAs in ascending k, so a lane that owns a column and
// runs fmaf over the input block (k = 0 .. D-1) and then the recurrent block (k = 0 .. NH-1) reproduces that column of the
// tile.  One env has 2-3 rows per net: on a 16-row tile 13 of 16 rows would be idle and every B operand a 256-byte load; here a
// lane owns the units 2 lane, 2 lane + 1 -- all four gates of them, columns g NH + 2 lane + {0, 1} -- and a k-step costs four
// 512-byte loads for the whole wave.  xr[r] / hr[r]: LDS rows (16-byte aligned; x rows zero-padded to a multiple of four).
template <int NH, int R>
__device__ __forceinline__ void lstm_gates_valu(const float* wx, const float* wh, int D, const float* const (&xr)[R], const float* const (&hr)[R],
                                                int lane, float (&z)[4][2][R]) {
  static_assert(NH == 128, "two units per lane");
  const int nx = (D + 3) >> 2, nch = nx + NH / 4;
  const int col = 2 * lane;
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int r = 0; r < R; r++) { z[g][0][r] = 0.0f; z[g][1][r] = 0.0f; }
  auto fetch = [&](int ch, f32x2u (&w)[4][4]) {
    const bool in_x = ch < nx;
    const float PT_GAS* base = pt_global(in_x ? wx : wh);
    const int k0 = 4 * (in_x ? ch : ch - nx), kmax = in_x ? D : NH;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      int k = k0 + kk;
      if (k >= kmax) k = kmax - 1;   // the x rows are zero there: fma(0, w, z) == z for any finite weight
      const float PT_GAS* wrow = base + (size_t)k * 4 * NH + col;
#pragma unroll
      for (int g = 0; g < 4; g++) w[kk][g] = *(const f32x2u PT_GAS*)(wrow + g * NH);
    }
  };
  auto consume = [&](int ch, const f32x2u (&w)[4][4]) {
    const bool in_x = ch < nx;
    const int k0 = 4 * (in_x ? ch : ch - nx);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float4 xv = *(const float4*)((in_x ? xr[r] : hr[r]) + k0);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          z[g][0][r] = __builtin_fmaf(xs[kk], w[kk][g].x, z[g][0][r]);
          z[g][1][r] = __builtin_fmaf(xs[kk], w[kk][g].y, z[g][1][r]);
        }
    }
  };
  // a chunk costs one trip to the L2 / MALL (~0.8 us under load) unless enough of them are in flight: ring of four chunk buffers,
  // three chunks (48 loads, 24 KB per wave) ahead of the products
  f32x2u w[4][4][4];
  fetch(0, w[0]); fetch(1, w[1]); fetch(2, w[2]);
  int ch = 0;
  for (; ch + 4 <= nch; ch += 4) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int nxt = ch + b + 3;
      fetch(nxt < nch ? nxt : nch - 1, w[(b + 3) & 3]);   // (no branch around the loads: past the end the last chunk is fetched again)
      consume(ch + b, w[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < 3; b++)   // the last nch % 4 chunks are in flight in buffers 0 .. 2
    if (ch + b < nch) consume(ch + b, w[b]);
}

// Heads on R latent rows in LDS (hn [R][NH]): lane i < A accumulates column i of the Gaussian mean, lane 16 the value, over the
// units in ascending order (the k order of the head tiles of ppo_lstm_step_kernel); biases are added by the caller.
template <int NH, int R>
__device__ __forceinline__ void lstm_heads_valu(const float* head_w_, const float* vf_w_, int A, const float* hn, int lane, float (&acc)[R]) {
  const float PT_GAS* head_w = pt_global(head_w_);
  const float PT_GAS* vf_w = pt_global(vf_w_);
  const bool pi = lane < A, vf = lane == 16;
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.0f;
  for (int j0 = 0; j0 < NH; j0 += 64) {   // 64 weight loads in flight: two trips to the L2 per head instead of one per 16 units
    float w[64];
#pragma unroll
    for (int u = 0; u < 64; u++) w[u] = pi ? head_w[(j0 + u) * A + lane] : (vf ? vf_w[j0 + u] : 0.0f);
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const float4 hv = *(const float4*)(hn + r * NH + j0 + 4 * q);
        acc[r] = __builtin_fmaf(hv.x, w[4 * q], acc[r]); acc[r] = __builtin_fmaf(hv.y, w[4 * q + 1], acc[r]);
        acc[r] = __builtin_fmaf(hv.z, w[4 * q + 2], acc[r]); acc[r] = __builtin_fmaf(hv.w, w[4 * q + 3], acc[r]);
      }
  }
}

#endif
