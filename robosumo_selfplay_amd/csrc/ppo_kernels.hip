// ppo_kernels.hip -- MI355X (gfx950) kernels for the PPO2 self-play rollout / update arithmetic (include/sumo_ppo.h).
//
// Dense layers run on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact f32, so results stay comparable with
// the reference's float32 TF graph).  Inference: one wavefront owns a 16-row tile of the batch, activations of the tile live in
// LDS (row stride == 2 mod 32 floats -> conflict-free A-operand reads), weights are read straight from L2 as B
// operands (24.5 k parameters, resident).  Training (ppo_grad_kernel): four waves share a tile, each owning 16 hidden units with
// its weight slices resident in registers; weight gradients accumulate in MFMA accumulator registers across all tiles of a
// workgroup and are written once as a per-workgroup slab that a second kernel reduces in a fixed order (deterministic, no float
// atomics).
//
// Reference arithmetic reproduced: policies.py:14-128 / baselines models.py:74-103 / distributions.py:227-251 (forward),
// model.py:65-139 (loss, gradient clipping, Adam), model.py:180-185 (advantage normalisation), runner.py:127-143,166-196
// (reward curriculum, IS ratios, V-trace).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sumo_ppo.h"

#define WAVE 64
#include "ppo_tile.h"   /* parameter layout + the one-wave / one-tile trunk and Gaussian-head device functions (shared with sumo_engine.hip) */
#define H PT_H
#define HS PT_HS
#define MAXA PT_MAXA
#define LOG2PI_F PT_LOG2PI_F
#define MFMA PT_MFMA

static thread_local char g_err[256];
extern "C" const char* ppo_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return code; } while (0)
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) FAIL(-100, "%s failed: %s", #expr, hipGetErrorString(_e)); } while (0)

// stage 16 rows of obs (row ids from idx, or consecutive) into xbuf[16][XS], zero padded
// (optional observation filter of the policy-zoo nets: clip((x - mean) * invstd, -clip, clip))
__device__ __forceinline__ void stage_x(float* xbuf, int XS, const float* obs, int obs_stride, int D, const int32_t* idx, int r0,
                                        int n, int lane, const float* f_mean = nullptr, const float* f_invstd = nullptr,
                                        float f_clip = 0.0f, int nthreads = WAVE) {
  // one column chunk of all 16 rows at a time: the 16 row loads are in flight together (no per-element division either);
  // `lane` may be a workgroup-wide thread id with nthreads = the workgroup size
  for (int c0 = 0; c0 < XS; c0 += nthreads) {
    const int c = c0 + lane;
    const bool cok = c < D;
    const float fm = (f_mean && cok) ? f_mean[c] : 0.0f, fi = (f_mean && cok) ? f_invstd[c] : 1.0f;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = r0 + r;
      v[r] = 0.0f;
      if (row < n && cok) {
        const int src = idx ? idx[row] : row;
        v[r] = obs[(size_t)src * obs_stride + c];
      }
    }
    if (f_mean) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (r0 + r < n && cok) v[r] = fminf(fmaxf((v[r] - fm) * fi, -f_clip), f_clip);
    }
    if (c < XS) {
#pragma unroll
      for (int r = 0; r < 16; r++) xbuf[r * XS + c] = v[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// inference
// ---------------------------------------------------------------------------------------------------------
struct FwdArgs {
  const float *params, *obs, *noise, *given;
  float *action, *neglogp, *value, *mean;
  const float *f_mean, *f_invstd;   // optional observation filter (policy-zoo nets)
  float f_clip;
  int n, obs_stride, flags, XS;
  ParamLayout L;
};

extern __shared__ float smem_f[];

__global__ void __launch_bounds__(256) ppo_forward_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tile = blockIdx.x * (blockDim.x >> 6) + wid, r0 = tile * 16;
  if (r0 >= a.n) return;
  const int XS = a.XS, D = a.L.D, A = a.L.A;
  float* base = smem_f + wid * (16 * XS + 2 * 16 * HS);
  float *xbuf = base, *h1 = base + 16 * XS, *h2 = h1 + 16 * HS;
  const int i = lane & 15, kq = lane >> 4;
  stage_x(xbuf, XS, a.obs, a.obs_stride, D, nullptr, r0, a.n, lane, a.f_mean, a.f_invstd, a.f_clip);
  wave_sync();
  const bool use_tanh = (a.flags & PPO_FWD_TANH) != 0;
  if (a.flags & PPO_FWD_PI) {
    Net net = pi_net(a.params, a.L);
    f32x4 mean = use_tanh ? trunk_forward<true>(net, xbuf, XS, D, h1, h2, lane) : trunk_forward<false>(net, xbuf, XS, D, h1, h2, lane);
    const bool col = i < A;
    float logstd = col ? a.params[a.L.logstd + i] : 0.0f;
    float std = expf(logstd);
    float sum_logstd = row16_sum(logstd);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int row = r0 + 4 * kq + r;
      bool ok = col && row < a.n;
      float m = mean[r], act = m;
      if (ok) {
        if (a.given) act = a.given[(size_t)row * A + i];
        else if (a.noise) act = m + std * a.noise[(size_t)row * A + i];
        if (a.action) a.action[(size_t)row * A + i] = act;
        if (a.mean) a.mean[(size_t)row * A + i] = m;
      }
      float z = ok ? (act - m) / std : 0.0f;
      float ss = row16_sum(z * z);
      if (a.neglogp && i == 0 && row < a.n) a.neglogp[row] = 0.5f * ss + 0.5f * LOG2PI_F * (float)A + sum_logstd;
    }
    wave_sync();
  }
  if (a.flags & PPO_FWD_VF) {
    Net net = vf_net(a.params, a.L);
    f32x4 v = use_tanh ? trunk_forward<true>(net, xbuf, XS, D, h1, h2, lane) : trunk_forward<false>(net, xbuf, XS, D, h1, h2, lane);
    if (a.value && i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) { int row = r0 + 4 * kq + r; if (row < a.n) a.value[row] = v[r]; }
    }
  }
}

extern "C" int ppo_forward_filtered(const float* params, const float* obs, int n, int obs_stride, int ob_dim, int ac_dim, int flags,
                                    const float* obs_mean, const float* obs_invstd, float obs_clip, const float* noise,
                                    const float* given_action, float* action_out, float* neglogp_out, float* value_out,
                                    float* mean_out, void* stream);
extern "C" int ppo_forward(const float* params, const float* obs, int n, int obs_stride, int ob_dim, int ac_dim, int flags,
                           const float* noise, const float* given_action, float* action_out, float* neglogp_out,
                           float* value_out, float* mean_out, void* stream) {
  return ppo_forward_filtered(params, obs, n, obs_stride, ob_dim, ac_dim, flags, nullptr, nullptr, 0.0f, noise, given_action,
                              action_out, neglogp_out, value_out, mean_out, stream);
}
extern "C" int ppo_forward_filtered(const float* params, const float* obs, int n, int obs_stride, int ob_dim, int ac_dim, int flags,
                                    const float* obs_mean, const float* obs_invstd, float obs_clip, const float* noise,
                                    const float* given_action, float* action_out, float* neglogp_out, float* value_out,
                                    float* mean_out, void* stream) {
  if (!params || !obs || n <= 0) FAIL(-1, "bad arguments");
  if (ac_dim < 1 || ac_dim > MAXA) FAIL(-2, "ac_dim %d not in [1,%d]", ac_dim, MAXA);
  if (ob_dim < 1 || ob_dim > 512 || obs_stride < ob_dim) FAIL(-3, "bad ob_dim/obs_stride");
  if ((obs_mean == nullptr) != (obs_invstd == nullptr)) FAIL(-4, "obs_mean and obs_invstd must be given together");
  FwdArgs a;
  a.params = params; a.obs = obs; a.noise = noise; a.given = given_action; a.action = action_out; a.neglogp = neglogp_out;
  a.value = value_out; a.mean = mean_out; a.n = n; a.obs_stride = obs_stride; a.flags = flags; a.XS = x_stride(ob_dim);
  a.f_mean = obs_mean; a.f_invstd = obs_invstd; a.f_clip = obs_clip;
  a.L = make_layout(ob_dim, ac_dim);
  // one wave (one 16-row tile) per workgroup: 29 KB of LDS each, so a tile can start on any CU as soon as a little LDS frees
  // up next to the env kernels of another env group (a 4-wave workgroup needed 116 KB at once)
  const int wpb = 1;
  size_t lds = (size_t)wpb * (16 * a.XS + 2 * 16 * HS) * sizeof(float);
  int tiles = (n + 15) / 16;
  static thread_local size_t lds_set = 0;   // raise the dynamic-LDS limit once (per thread / size), not on every launch
  if (lds > 64 * 1024 && lds > lds_set) {
    HIPCHK(hipFuncSetAttribute((const void*)ppo_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(ppo_forward_kernel, dim3((tiles + wpb - 1) / wpb), dim3(64 * wpb), lds, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The five evaluations of one self-play rollout step (runner.py:62-96) in ONE launch: workgroup (tile, side) stages 16
// observations of agent `side` once and runs the acting net's policy trunk (sample + neglogp), the other net's policy
// trunk (neglogp of that action) and the learner's value trunk on them.  Same device functions and the same order of
// operations as ppo_forward_kernel: the outputs are bit-identical to the four separate launches.
// ---------------------------------------------------------------------------------------------------------
struct SelfplayArgs {
  const float *learner, *opponent, *obs, *noise0, *noise1;
  const uint8_t* done_in;
  float *act_env, *obs_out0, *obs_out1, *act0, *act1, *nlp0, *nlp1, *onlp0, *onlp1, *val0, *val1;
  uint8_t *done_out0, *done_out1;
  int n, env_stride, agent_stride, XS;
  ParamLayout L;
};

__global__ void __launch_bounds__(192) ppo_selfplay_kernel(SelfplayArgs a) {
  // three waves per (tile, side): wave 0 the acting net's policy trunk (samples), wave 1 the scoring net's policy trunk,
  // wave 2 the learner's value trunk -- in parallel on the shared observation tile; the sampled action reaches wave 1
  // through LDS.  (One wave running the three trunks back to back took 45 us between two env steps of a group.)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, side = blockIdx.y, r0 = blockIdx.x * 16;
  if (r0 >= a.n) return;
  const int XS = a.XS, D = a.L.D, A = a.L.A;
  float* xbuf = smem_f;
  float* actl = xbuf + 16 * XS;                       // [16][17] sampled actions
  float* h1 = actl + 16 * 17 + wid * (2 * 16 * HS);   // per-wave activation tiles
  float* h2 = h1 + 16 * HS;
  const int i = lane & 15, kq = lane >> 4;
  stage_x(xbuf, XS, a.obs + (size_t)side * a.agent_stride, a.env_stride, D, nullptr, r0, a.n, tid, nullptr, nullptr, 0.0f, 192);
  __syncthreads();
  const float* actor = side == 0 ? a.learner : a.opponent;
  const float* scorer = side == 0 ? a.opponent : a.learner;
  float* nlp_out = side == 0 ? a.nlp0 : a.nlp1;        // the LEARNER's neglogp of the action taken on this side
  float* onlp_out = side == 0 ? a.onlp0 : a.onlp1;     // the OPPONENT's
  float act[4] = {0, 0, 0, 0}, nlp[4];
  f32x4 head = (f32x4){0, 0, 0, 0};
  if (wid == 0) {
    head = trunk_forward<false>(pi_net(actor, a.L), xbuf, XS, D, h1, h2, lane);
    gauss_head(actor, a.L, head, side == 0 ? a.noise0 : a.noise1, r0, a.n, lane, act, nlp);
#pragma unroll
    for (int r = 0; r < 4; r++) actl[(4 * kq + r) * 17 + i] = act[r];
  } else if (wid == 1) {
    head = trunk_forward<false>(pi_net(scorer, a.L), xbuf, XS, D, h1, h2, lane);
  } else {
    head = trunk_forward<false>(vf_net(a.learner, a.L), xbuf, XS, D, h1, h2, lane);
  }
  __syncthreads();
  if (wid == 0) {
    float* act_out = side == 0 ? a.act0 : a.act1;
    float* out = side == 0 ? nlp_out : onlp_out;       // the acting net is the learner on side 0, the opponent on side 1
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      if (row >= a.n) continue;
      if (i < A) {
        act_out[(size_t)row * A + i] = act[r];
        a.act_env[((size_t)row * 2 + side) * A + i] = act[r];
      }
      if (i == 0) out[row] = nlp[r];
    }
  } else if (wid == 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) act[r] = actl[(4 * kq + r) * 17 + i];
    gauss_head(scorer, a.L, head, nullptr, r0, a.n, lane, act, nlp);
    float* out = side == 0 ? onlp_out : nlp_out;
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) { const int row = r0 + 4 * kq + r; if (row < a.n) out[row] = nlp[r]; }
    }
  } else {
    float* val_out = side == 0 ? a.val0 : a.val1;
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) { const int row = r0 + 4 * kq + r; if (row < a.n) val_out[row] = head[r]; }
    }
  }
  // rollout records of the inputs: the staged observations and the done flags the step started from
  float* obs_out = side == 0 ? a.obs_out0 : a.obs_out1;
  if (obs_out) {
    for (int r = 0; r < 16 && r0 + r < a.n; r++)
      for (int c = tid; c < D; c += 192) obs_out[(size_t)(r0 + r) * D + c] = xbuf[r * XS + c];
  }
  uint8_t* done_out = side == 0 ? a.done_out0 : a.done_out1;
  if (done_out && a.done_in && tid < 16 && r0 + tid < a.n) done_out[r0 + tid] = a.done_in[(size_t)(r0 + tid) * 2 + side];
}

extern "C" int ppo_selfplay_forward(const float* learner_params, const float* opponent_params, const float* obs, int n, int env_stride,
                                    int agent_stride, int ob_dim, int ac_dim, const float* noise0, const float* noise1,
                                    const uint8_t* done_in, float* act_env, float* const* out_f32, uint8_t* const* out_done,
                                    void* stream) {
  if (!learner_params || !opponent_params || !obs || !noise0 || !noise1 || !act_env || !out_f32 || n <= 0) FAIL(-1, "bad arguments");
  if (ac_dim < 1 || ac_dim > MAXA) FAIL(-2, "ac_dim %d not in [1,%d]", ac_dim, MAXA);
  if (ob_dim < 1 || ob_dim > 512 || agent_stride < ob_dim || env_stride < agent_stride + ob_dim) FAIL(-3, "bad ob_dim / strides");
  for (int k = 2; k < 10; k++) if (!out_f32[k]) FAIL(-4, "output %d missing", k);
  SelfplayArgs a;
  a.learner = learner_params; a.opponent = opponent_params; a.obs = obs; a.noise0 = noise0; a.noise1 = noise1; a.done_in = done_in;
  a.act_env = act_env;
  a.obs_out0 = out_f32[0]; a.obs_out1 = out_f32[1]; a.act0 = out_f32[2]; a.act1 = out_f32[3]; a.nlp0 = out_f32[4]; a.nlp1 = out_f32[5];
  a.onlp0 = out_f32[6]; a.onlp1 = out_f32[7]; a.val0 = out_f32[8]; a.val1 = out_f32[9];
  a.done_out0 = out_done ? out_done[0] : nullptr; a.done_out1 = out_done ? out_done[1] : nullptr;
  a.n = n; a.env_stride = env_stride; a.agent_stride = agent_stride; a.XS = x_stride(ob_dim);
  a.L = make_layout(ob_dim, ac_dim);
  size_t lds = (size_t)(16 * a.XS + 16 * 17 + 3 * 2 * 16 * HS) * sizeof(float);
  static thread_local size_t lds_set = 0;
  if (lds > 64 * 1024 && lds > lds_set) {
    HIPCHK(hipFuncSetAttribute((const void*)ppo_selfplay_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(ppo_selfplay_kernel, dim3((n + 15) / 16, 2), dim3(192), lds, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// recurrent policies: one LSTM time step (see sumo_ppo.h).  One workgroup of four waves per 16-row tile (a wave per
// quarter of the units).  LDS per workgroup: x [16][XS] | emb [16][HS] | h_prev [16][HP] | h_new [16][HP]
// ---------------------------------------------------------------------------------------------------------
#define LSTM_MAXH 128

struct LstmArgs {
  ppo_lstm_net net;
  const ppo_lstm_net* pool;   // opponent pool: device array of nets (same dimensions as `net`) ...
  const int32_t* tile_net;    // ... and the net each 16-row tile is evaluated with (NULL pool: `net` for every tile)
  const float *obs, *mask, *noise, *given;
  float *c, *h, *action, *neglogp, *value, *mean;
  float *sv_gates, *sv_cprev, *sv_hprev, *sv_tanhc;   // training: per-step records for the backward pass (all or none)
  const float* zinit;         // training: x * wx of this time step [n][4 NH] from ppo_lstm_xproj (the gate tiles start from it; no obs)
  float* sv_latent;           // training: the new latent [n][NH] (a copy of the h rows, in the layout the head / weight gradients read)
  int n, obs_stride, state_stride, XS, HP;
};

// NH hidden units: 64 or 128; gate order (static so the gate tiles are static registers); ZIN: gate sums start from LstmArgs::zinit;
// NW waves per tile: 4, or 8 for NH = 128 (one unit tile per wave: half the products and half the transcendentals on a wave's chain)
template <int NH, int ORDER, bool ZIN = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW) ppo_lstm_step_kernel(LstmArgs a) {
  constexpr int NT = 64 * NW;
  // FOUR waves per 16-row tile: wave w owns the units [w NH/4, (w+1) NH/4) -- all four gates of those units, their cell
  // update and their slice of the new latent -- so the serial chain per wave is a quarter of the tile's (one wave per tile
  // took 150 us for H = 128 whatever the batch: 63 dependent k-steps of 32 products, then 128 units of transcendentals).
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r0 = blockIdx.x * 16;
  const ppo_lstm_net N = a.pool ? a.pool[a.tile_net[blockIdx.x]] : a.net;
  const int D = N.ob_dim, E = N.emb_dim, A = N.ac_dim, XS = a.XS, HP = a.HP;
  float* xbuf = smem_f;
  float* ebuf = xbuf + 16 * XS;
  float* hprev = ebuf + 16 * HS;
  float* hnew = hprev + 16 * HP;
  const int i = lane & 15, kq = lane >> 4;
  constexpr int UT = NH / 16, UTW = UT / NW, NTW = 4 * UTW;   // unit tiles, unit tiles per wave, gate tiles per wave
  static_assert(UTW >= 1 && UTW * NW == UT, "a whole number of unit tiles per wave");
  f32x4 z[NTW];
  if (ZIN) {   // (first thing in the kernel: these loads are in flight while the previous latent is staged)
    // the input block's partial sums come from ppo_lstm_xproj (same k order from zero): continuing the accumulation on them with
    // the recurrent block gives the very sums of the full loop, with half the dependent chain
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int u = 0; u < UTW; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = r0 + 4 * kq + r;
          z[g * UTW + u][r] = row < a.n ? a.zinit[(size_t)row * 4 * NH + (g * UT + wid * UTW + u) * 16 + i] : 0.0f;
        }
  } else {
#pragma unroll
    for (int ct = 0; ct < NTW; ct++) z[ct] = (f32x4){0, 0, 0, 0};
  }
  // weight operands of the gate sums, by chunks of LCH k-steps (see the loop below); nothing here depends on the staged tiles
  constexpr int LCH = 4;
  const int xk0 = ZIN ? 0 : (N.emb_w ? E : D);
  const int xsteps = (xk0 + 3) >> 2, nsteps = xsteps + NH / 4, nchunks = (nsteps + LCH - 1) / LCH;
  const float PT_GAS* gwx = pt_global(N.wx);
  const float PT_GAS* gwh = pt_global(N.wh);
  auto fetch_w = [&](int c, float (&b)[LCH][NTW]) {
#pragma unroll
    for (int q = 0; q < LCH; q++) {
      const int s_ = c * LCH + q;
      if (s_ < xsteps) {
        const int k = 4 * s_ + kq;
        const bool ok = k < xk0;
        const float PT_GAS* wrow = gwx + (size_t)(ok ? k : 0) * 4 * NH;
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int u = 0; u < UTW; u++) { const float wv = wrow[(g * UT + wid * UTW + u) * 16 + i]; b[q][g * UTW + u] = ok ? wv : 0.0f; }   // (a plain load from the clamped row, then the select: `ok ? load : 0` is a predicated load with its own wait)
      } else if (s_ < nsteps) {
        const int k = 4 * (s_ - xsteps) + kq;
        const float PT_GAS* wrow = gwh + (size_t)k * 4 * NH;
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int u = 0; u < UTW; u++) b[q][g * UTW + u] = wrow[(g * UT + wid * UTW + u) * 16 + i];
      }
    }
  };
  float wring[3][LCH][NTW];
  fetch_w(0, wring[0]);
  if (nchunks > 1) fetch_w(1, wring[1]);
  if (!ZIN) stage_x(xbuf, XS, a.obs, a.obs_stride, D, nullptr, r0, a.n, tid, N.obs_mean, N.obs_invstd, N.obs_clip, NT);
  // h_prev * (1 - mask): eight state loads in flight per thread
  for (int e0 = 0; e0 < 16 * NH; e0 += 8 * NT) {
    float v[8], keep[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + NT * u + tid, r = e / NH, k = e - r * NH, row = r0 + r;
      const bool in = e < 16 * NH && row < a.n;
      const int rowc = row < a.n ? row : a.n - 1, kc = k < NH ? k : 0;     // plain loads at clamped indices, selects afterwards
      const float hv = a.h[(size_t)rowc * a.state_stride + kc];
      const float mv = a.mask ? a.mask[rowc] : 0.0f;
      v[u] = in ? hv : 0.0f;
      keep[u] = in ? 1.0f - mv : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + NT * u + tid, r = e / NH, k = e - r * NH, row = r0 + r;
      if (e < 16 * NH) {
        const float hv = a.mask ? v[u] * keep[u] : v[u];
        hprev[r * HP + k] = hv;
        if (a.sv_hprev && row < a.n) a.sv_hprev[(size_t)row * NH + k] = hv;
      }
    }
  }
  __syncthreads();
  // ---- optional embedding: relu(x * We + be), 64 wide (wave 0)
  const float* xin = xbuf; int xk = D, xs = XS;
  if (N.emb_w && !ZIN) {
    if (wid == 0) {
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) acc[ct] = (f32x4){0, 0, 0, 0};
      const int Dp = (D + 3) & ~3;
      for (int k0 = 0; k0 < Dp; k0 += 4) {
        int k = k0 + kq;
        float av = xbuf[i * XS + k];
        bool ok = k < D;
#pragma unroll
        for (int ct = 0; ct < 4; ct++) { float b = (ok && ct * 16 + i < E) ? N.emb_w[k * E + ct * 16 + i] : 0.0f; acc[ct] = MFMA(av, b, acc[ct]); }
      }
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        float bias = ct * 16 + i < E ? N.emb_b[ct * 16 + i] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) ebuf[(4 * kq + r) * HS + ct * 16 + i] = fmaxf(acc[ct][r] + bias, 0.0f);
      }
    }
    __syncthreads();
    xin = ebuf; xk = E; xs = HS;
  }
  // ---- gates z = x * wx + h_prev * wh + b for this wave's units: tiles (gate g, unit tile wid*UTW + u)
  {
    // k-steps run over the input block (wx) and then the recurrent block (wh), accumulation order per tile as ever.  The weight
    // operands come from L2 (~0.5 us a trip) and a k-step's eight products take a tenth of that: chunks of four k-steps, ring of three
    // register sets, two chunks (64 loads per lane) in flight ahead of the products -- the first two were issued at the top of the kernel
    for (int c0 = 0; c0 < nchunks; c0 += 3) {
#pragma unroll
      for (int b = 0; b < 3; b++) {
        if (c0 + b + 2 < nchunks) fetch_w(c0 + b + 2, wring[(b + 2) % 3]);
        if (c0 + b < nchunks) {
          float av[LCH];
#pragma unroll
          for (int q = 0; q < LCH; q++) {
            const int s_ = (c0 + b) * LCH + q;
            if (s_ < xsteps) { const int k = 4 * s_ + kq; av[q] = k < xk ? xin[i * xs + k] : 0.0f; }
            else av[q] = s_ < nsteps ? hprev[i * HP + 4 * (s_ - xsteps) + kq] : 0.0f;
          }
#pragma unroll
          for (int q = 0; q < LCH; q++)
            if ((c0 + b) * LCH + q < nsteps) {
#pragma unroll
              for (int ct = 0; ct < NTW; ct++) z[ct] = MFMA(av[q], wring[b][q][ct], z[ct]);
            }
        }
      }
    }
  }
  // ---- cell update.  Lane (i, kq) holds rows 4kq+r of columns ct*16+i: all four gates of unit j = ut*16+i
  constexpr int gi = 0, gf = ORDER == PPO_LSTM_GATES_IFOU ? 1 : 2, go = ORDER == PPO_LSTM_GATES_IFOU ? 2 : 3,
                gu = ORDER == PPO_LSTM_GATES_IFOU ? 3 : 1;
#pragma unroll
  for (int u = 0; u < UTW; u++) {
    const int ut = wid * UTW + u;
    const int j = ut * 16 + i;
    const f32x4 zi = z[gi * UTW + u], zf = z[gf * UTW + u], zo = z[go * UTW + u], zu = z[gu * UTW + u];
    const float bi = N.b[gi * NH + j], bf = N.b[gf * NH + j] + N.forget_bias, bo = N.b[go * NH + j], bu = N.b[gu * NH + j];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      float cp = 0.0f;
      if (row < a.n) { cp = a.c[(size_t)row * a.state_stride + j]; if (a.mask) cp *= 1.0f - a.mask[row]; }
      const LstmCell cell = lstm_cell(zi[r], zf[r], zo[r], zu[r], bi, bf, bo, bu, cp);   // shared with the fused rollout kernel (ppo_tile.h)
      const float ig = cell.ig, fg = cell.fg, og = cell.og, ug = cell.ug, cn = cell.cn, tcn = cell.tcn, hn = cell.hn;
      if (row < a.n) {
        a.c[(size_t)row * a.state_stride + j] = cn; a.h[(size_t)row * a.state_stride + j] = hn;
        if (a.sv_gates) {
          float* sg = a.sv_gates + (size_t)row * 4 * NH;
          sg[gi * NH + j] = ig; sg[gf * NH + j] = fg; sg[go * NH + j] = og; sg[gu * NH + j] = ug;
          a.sv_cprev[(size_t)row * NH + j] = cp; a.sv_tanhc[(size_t)row * NH + j] = tcn;
        }
        if (a.sv_latent) a.sv_latent[(size_t)row * NH + j] = hn;
      }
      hnew[(4 * kq + r) * HP + j] = hn;
    }
  }
  __syncthreads();
  // ---- heads on the new latent: wave 0 the Gaussian head, wave 1 the value head
  if (wid == 0 && N.head_w && (a.action || a.neglogp || a.mean)) {   // (the training forward records the latents only)
    f32x4 m4 = (f32x4){0, 0, 0, 0};
    const bool col = i < A;
    {
      float hw[NH / 4];   // all head-weight operands in flight before the first product
#pragma unroll
      for (int u = 0; u < NH / 4; u++) hw[u] = col ? N.head_w[(4 * u + kq) * A + i] : 0.0f;
#pragma unroll
      for (int u = 0; u < NH / 4; u++) m4 = MFMA(hnew[i * HP + 4 * u + kq], hw[u], m4);
    }
    const float hb = col ? N.head_b[i] : 0.0f, logstd = col ? N.logstd[i] : 0.0f, std = expf(logstd);
    const float sum_logstd = row16_sum(logstd);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      const bool ok = col && row < a.n;
      const float m = m4[r] + hb;
      float act = m;
      const bool sample = !a.given && a.noise;
      if (ok && a.given) act = a.given[(size_t)row * A + i];
      const float nlp = gauss_row(m, std, sum_logstd, ok, sample, (ok && sample) ? a.noise[(size_t)row * A + i] : 0.0f, act, A);
      if (ok) {
        if (a.action) a.action[(size_t)row * A + i] = act;
        if (a.mean) a.mean[(size_t)row * A + i] = m;
      }
      if (a.neglogp && i == 0 && row < a.n) a.neglogp[row] = nlp;
    }
  }
  if (wid == 1 && N.vf_w && a.value) {
    f32x4 v4 = (f32x4){0, 0, 0, 0};
    {
      float vw[NH / 4];
#pragma unroll
      for (int u = 0; u < NH / 4; u++) vw[u] = i == 0 ? N.vf_w[4 * u + kq] : 0.0f;
#pragma unroll
      for (int u = 0; u < NH / 4; u++) v4 = MFMA(hnew[i * HP + 4 * u + kq], vw[u], v4);
    }
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) { int row = r0 + 4 * kq + r; if (row < a.n) a.value[row] = v4[r] + N.vf_b[0]; }
    }
  }
}

// x * wx for ALL time steps of a training minibatch at once (rows = time x env): the input block of the gate sums does not
// depend on the recurrence, so it leaves the sequential part of the unrolled forward (the classic split of an LSTM forward into
// one large input GEMM and T small recurrent steps).  Same tiles, same k order from zero as ppo_lstm_step_kernel's input block:
// the step kernel continues the accumulation on these partial sums (LstmArgs::zinit) and reproduces its own full loop bit for bit.
template <int NH>
__global__ void __launch_bounds__(256) ppo_lstm_xproj_kernel(ppo_lstm_net N, const float* obs, int rows, int obs_stride, float* zout, int XS) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r0 = blockIdx.x * 16;
  const int D = N.ob_dim, i = lane & 15, kq = lane >> 4;
  float* xbuf = smem_f;
  stage_x(xbuf, XS, obs, obs_stride, D, nullptr, r0, rows, tid, N.obs_mean, N.obs_invstd, N.obs_clip, 256);
  __syncthreads();
  constexpr int UT = NH / 16, UTW = UT / 4, NTW = 4 * UTW;
  f32x4 z[NTW];
#pragma unroll
  for (int ct = 0; ct < NTW; ct++) z[ct] = (f32x4){0, 0, 0, 0};
  const int xsteps = (D + 3) >> 2;
  auto fetch = [&](int s_, float& av, float (&b)[NTW]) {
    const int k = 4 * s_ + kq;
    const bool ok = k < D;
    av = ok ? xbuf[i * XS + k] : 0.0f;
    const float* wrow = N.wx + (size_t)(ok ? k : 0) * 4 * NH;
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int u = 0; u < UTW; u++) { const float wv = wrow[(g * UT + wid * UTW + u) * 16 + i]; b[g * UTW + u] = ok ? wv : 0.0f; }
  };
  float a0 = 0.0f, a1 = 0.0f, b0[NTW], b1[NTW];
  fetch(0, a0, b0);
  for (int s_ = 0; s_ < xsteps; s_ += 2) {
    if (s_ + 1 < xsteps) fetch(s_ + 1, a1, b1);
#pragma unroll
    for (int ct = 0; ct < NTW; ct++) z[ct] = MFMA(a0, b0[ct], z[ct]);
    if (s_ + 2 < xsteps) fetch(s_ + 2, a0, b0);
    if (s_ + 1 < xsteps) {
#pragma unroll
      for (int ct = 0; ct < NTW; ct++) z[ct] = MFMA(a1, b1[ct], z[ct]);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int u = 0; u < UTW; u++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = r0 + 4 * kq + r;
        if (row < rows) zout[(size_t)row * 4 * NH + (g * UT + wid * UTW + u) * 16 + i] = z[g * UTW + u][r];
      }
}

static int lstm_launch(const ppo_lstm_net* net, const float* obs, int n, int obs_stride, const float* mask, float* c, float* h,
                       int state_stride, const float* noise, const float* given_action, float* action_out, float* neglogp_out,
                       float* value_out, float* mean_out, float* sv_gates, float* sv_cprev, float* sv_hprev, float* sv_tanhc,
                       void* stream, const ppo_lstm_net* pool_dev = nullptr, const int32_t* tile_net_dev = nullptr,
                       const float* zinit = nullptr, float* sv_latent = nullptr) {
  if (!net || (!obs && !zinit) || !c || !h || n <= 0) FAIL(-1, "bad arguments");
  if (zinit && (net->emb_w || net->obs_mean)) FAIL(-10, "precomputed input sums: nets without embedding / observation filter only");
  if (net->hidden != 64 && net->hidden != 128) FAIL(-2, "hidden %d: only 64 and 128 are built", net->hidden);
  if (net->ob_dim < 1 || net->ob_dim > 512 || obs_stride < net->ob_dim) FAIL(-3, "bad ob_dim/obs_stride");
  if (net->emb_w && (net->emb_dim < 1 || net->emb_dim > 64 || !net->emb_b)) FAIL(-4, "embedding width %d not in [1,64]", net->emb_dim);
  if (!net->wx || !net->wh || !net->b) FAIL(-5, "missing LSTM weights");
  if (net->head_w && (net->ac_dim < 1 || net->ac_dim > MAXA || !net->head_b || !net->logstd)) FAIL(-6, "bad Gaussian head");
  if ((net->obs_mean == nullptr) != (net->obs_invstd == nullptr)) FAIL(-7, "obs_mean and obs_invstd must be given together");
  if (state_stride < net->hidden) FAIL(-8, "state_stride < hidden");
  LstmArgs a;
  a.pool = pool_dev; a.tile_net = tile_net_dev;
  a.net = *net; a.obs = obs; a.mask = mask; a.noise = noise; a.given = given_action; a.c = c; a.h = h; a.action = action_out;
  a.neglogp = neglogp_out; a.value = value_out; a.mean = mean_out; a.n = n; a.obs_stride = obs_stride; a.state_stride = state_stride;
  a.sv_gates = sv_gates; a.sv_cprev = sv_cprev; a.sv_hprev = sv_hprev; a.sv_tanhc = sv_tanhc;
  a.zinit = zinit; a.sv_latent = sv_latent;
  a.XS = x_stride(net->ob_dim); a.HP = net->hidden + 2;
  size_t lds = (size_t)(16 * a.XS + 16 * HS + 2 * 16 * a.HP) * sizeof(float);
  int tiles = (n + 15) / 16;
  if (net->gate_order != PPO_LSTM_GATES_IFOU && net->gate_order != PPO_LSTM_GATES_IJFO) FAIL(-9, "unknown gate order %d", net->gate_order);
  const bool ifou = net->gate_order == PPO_LSTM_GATES_IFOU;
  dim3 g(tiles), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (zinit) {   // training forward on precomputed input sums (eight waves per tile where a wave still gets a whole unit tile)
    if (net->hidden == 64 && ifou) hipLaunchKernelGGL((ppo_lstm_step_kernel<64, PPO_LSTM_GATES_IFOU, true>), g, b, lds, st, a);
    else if (net->hidden == 64) hipLaunchKernelGGL((ppo_lstm_step_kernel<64, PPO_LSTM_GATES_IJFO, true>), g, b, lds, st, a);
    else if (ifou) hipLaunchKernelGGL((ppo_lstm_step_kernel<128, PPO_LSTM_GATES_IFOU, true, 8>), g, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((ppo_lstm_step_kernel<128, PPO_LSTM_GATES_IJFO, true, 8>), g, dim3(512), lds, st, a);
  } else
  if (net->hidden == 64 && ifou) hipLaunchKernelGGL((ppo_lstm_step_kernel<64, PPO_LSTM_GATES_IFOU>), g, b, lds, st, a);
  else if (net->hidden == 64) hipLaunchKernelGGL((ppo_lstm_step_kernel<64, PPO_LSTM_GATES_IJFO>), g, b, lds, st, a);
  else if (ifou) hipLaunchKernelGGL((ppo_lstm_step_kernel<128, PPO_LSTM_GATES_IFOU>), g, b, lds, st, a);
  else hipLaunchKernelGGL((ppo_lstm_step_kernel<128, PPO_LSTM_GATES_IJFO>), g, b, lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int ppo_lstm_step(const ppo_lstm_net* net, const float* obs, int n, int obs_stride, const float* mask, float* c, float* h,
                             int state_stride, const float* noise, const float* given_action, float* action_out,
                             float* neglogp_out, float* value_out, float* mean_out, void* stream) {
  return lstm_launch(net, obs, n, obs_stride, mask, c, h, state_stride, noise, given_action, action_out, neglogp_out, value_out, mean_out,
                     nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" int ppo_lstm_step_pool(const ppo_lstm_net* proto, const ppo_lstm_net* nets_dev, const int32_t* tile_net_dev, const float* obs, int n,
                                  int obs_stride, const float* mask, float* c, float* h, int state_stride, const float* noise,
                                  const float* given_action, float* action_out, float* neglogp_out, float* value_out, float* mean_out,
                                  void* stream) {
  if (!nets_dev || !tile_net_dev) FAIL(-1, "bad arguments");
  return lstm_launch(proto, obs, n, obs_stride, mask, c, h, state_stride, noise, given_action, action_out, neglogp_out, value_out, mean_out,
                     nullptr, nullptr, nullptr, nullptr, stream, nets_dev, tile_net_dev);
}
extern "C" int ppo_lstm_step_save(const ppo_lstm_net* net, const float* obs, int n, int obs_stride, const float* mask, float* c, float* h,
                                  int state_stride, float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc,
                                  void* stream) {
  if (!save_gates || !save_cprev || !save_hprev || !save_tanhc) FAIL(-1, "bad arguments");
  return lstm_launch(net, obs, n, obs_stride, mask, c, h, state_stride, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, save_gates,
                     save_cprev, save_hprev, save_tanhc, stream);
}

extern "C" int ppo_lstm_xproj(const ppo_lstm_net* net, const float* obs, int rows, int obs_stride, float* z_out, void* stream) {
  if (!net || !obs || !z_out || rows <= 0) FAIL(-1, "bad arguments");
  if (net->hidden != 64 && net->hidden != 128) FAIL(-2, "hidden %d: only 64 and 128 are built", net->hidden);
  if (net->ob_dim < 1 || net->ob_dim > 512 || obs_stride < net->ob_dim) FAIL(-3, "bad ob_dim/obs_stride");
  if (net->emb_w || net->obs_mean || net->obs_invstd) FAIL(-4, "nets without embedding / observation filter only");
  if (!net->wx) FAIL(-5, "missing LSTM weights");
  const int XS = x_stride(net->ob_dim);
  const size_t lds = (size_t)16 * XS * sizeof(float);
  dim3 g((rows + 15) / 16), b(256);
  if (net->hidden == 64) hipLaunchKernelGGL(ppo_lstm_xproj_kernel<64>, g, b, lds, (hipStream_t)stream, *net, obs, rows, obs_stride, z_out, XS);
  else hipLaunchKernelGGL(ppo_lstm_xproj_kernel<128>, g, b, lds, (hipStream_t)stream, *net, obs, rows, obs_stride, z_out, XS);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int ppo_lstm_step_save_z(const ppo_lstm_net* net, const float* z_t, int n, const float* mask, float* c, float* h, int state_stride,
                                    float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc, float* latent_out,
                                    void* stream) {
  if (!z_t || !save_gates || !save_cprev || !save_hprev || !save_tanhc) FAIL(-1, "bad arguments");
  return lstm_launch(net, nullptr, n, net ? net->ob_dim : 0, mask, c, h, state_stride, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, save_gates,
                     save_cprev, save_hprev, save_tanhc, stream, nullptr, nullptr, z_t, latent_out);
}

// ---- loss heads on stored latents: one thread per row (rows x hidden x (ac_dim + 1) MACs: small next to the recurrence)
__global__ void __launch_bounds__(128) ppo_lstm_head_grad_kernel(ppo_lstm_net N, const float* latent, int rows, const float* actions,
                                                                 const float* adv, const float* returns, const float* oldnlp,
                                                                 const float* weight, float inv_count, float cliprange, float vf_coef,
                                                                 float* dlatent, float* dmean_o, float* dvalue_o, float* dlogstd_rows,
                                                                 double* stats) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  // (loops over the action dimensions are unrolled to MAXA with guards: with a runtime trip count the per-row arrays are indexed
  // dynamically and live in scratch memory -- 1 500 cycles per unit of the latent)
  const int Hh = N.hidden, A = N.ac_dim;
  // the head weights, staged once per workgroup: [unit][action dims | value]; read from global memory inside the unit loops every
  // iteration waited for its own loads (the stores of the second loop keep the compiler from hoisting them)
  __shared__ float wsh[LSTM_MAXH * (MAXA + 1)];
  for (int e = threadIdx.x; e < Hh * (MAXA + 1); e += blockDim.x) {
    const int kk = e / (MAXA + 1), q = e - kk * (MAXA + 1);
    wsh[e] = q < A ? N.head_w[kk * A + q] : (q == MAXA ? N.vf_w[kk] : 0.0f);
  }
  __syncthreads();
  double s_pg = 0, s_vf = 0, s_kl = 0, s_cf = 0, s_n = 0;
  if (r < rows) {
    const float* L = latent + (size_t)r * Hh;
    float mean[MAXA], value = N.vf_b[0];
#pragma unroll
    for (int q = 0; q < MAXA; q++) mean[q] = q < A ? N.head_b[q] : 0.0f;
    for (int k = 0; k < Hh; k++) {
      const float lk = L[k];
      const float* wk = wsh + k * (MAXA + 1);
      value += lk * wk[MAXA];
#pragma unroll
      for (int q = 0; q < MAXA; q++) if (q < A) mean[q] += lk * wk[q];
    }
    float ss = 0, sum_logstd = 0, zq[MAXA], isd[MAXA];
#pragma unroll
    for (int q = 0; q < MAXA; q++) if (q < A) {
      const float ls = N.logstd[q];
      isd[q] = expf(-ls);
      zq[q] = (actions[(size_t)r * A + q] - mean[q]) * isd[q];
      ss += zq[q] * zq[q]; sum_logstd += ls;
    }
    const float nlp = 0.5f * ss + 0.5f * LOG2PI_F * (float)A + sum_logstd;
    const float old = oldnlp[r], ad = adv[r], w = weight[r], R = returns[r];
    float ratio = expf(old - nlp);
    const bool isnan_ = ratio != ratio;
    if (isnan_) ratio = 2.0f;                                                       // model.py:96
    const float rc = fminf(fmaxf(ratio, 1.0f - cliprange), 1.0f + cliprange);
    const float l1 = -ad * ratio, l2 = -ad * rc;
    const bool in_clip = ratio >= 1.0f - cliprange && ratio <= 1.0f + cliprange;
    float dratio = l1 >= l2 ? -ad : (in_clip ? -ad : 0.0f);                         // tf.maximum: ties go to the first argument
    dratio = isnan_ ? 0.0f : dratio * w * inv_count;
    const float dnlp = -dratio * ratio;
    const float dv = vf_coef * (value - R) * inv_count;
    s_pg = (double)(w * fmaxf(l1, l2)); s_vf = 0.5 * (double)(value - R) * (double)(value - R);
    s_kl = (double)(nlp - old); s_cf = fabsf(ratio - 1.0f) > cliprange ? 1.0 : 0.0; s_n = 1.0;
    float dm[MAXA];
#pragma unroll
    for (int q = 0; q < MAXA; q++) if (q < A) {
      dm[q] = dnlp * (-(zq[q] * isd[q]));
      dmean_o[(size_t)r * A + q] = dm[q];
      dlogstd_rows[(size_t)r * A + q] = dnlp * (1.0f - zq[q] * zq[q]);
    }
    dvalue_o[r] = dv;
    float* dl = dlatent + (size_t)r * Hh;
    for (int k = 0; k < Hh; k++) {
      const float* wk = wsh + k * (MAXA + 1);
      float acc = dv * wk[MAXA];
#pragma unroll
      for (int q = 0; q < MAXA; q++) if (q < A) acc += dm[q] * wk[q];
      dl[k] = acc;
    }
  }
  __shared__ double red[5][128];
  red[0][threadIdx.x] = s_pg; red[1][threadIdx.x] = s_vf; red[2][threadIdx.x] = s_kl; red[3][threadIdx.x] = s_cf; red[4][threadIdx.x] = s_n;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) for (int q = 0; q < 5; q++) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(stats + 0, red[0][0]); atomicAdd(stats + 1, red[1][0]); atomicAdd(stats + 3, red[2][0]); atomicAdd(stats + 4, red[3][0]);
    atomicAdd(stats + 6, red[4][0]);
  }
}
extern "C" int ppo_lstm_head_grad(const ppo_lstm_net* net, const float* latent, int rows, const float* actions, const float* adv,
                                  const float* returns, const float* old_neglogp, const float* is_weight, double inv_count,
                                  float cliprange, float vf_coef, float* dlatent, float* dmean, float* dvalue, float* dlogstd_rows,
                                  double* stats, void* stream) {
  if (!net || !latent || !actions || !adv || !returns || !old_neglogp || !is_weight || !dlatent || !dmean || !dvalue || !dlogstd_rows ||
      !stats || rows <= 0)
    FAIL(-1, "bad arguments");
  if (!net->head_w || !net->head_b || !net->logstd || !net->vf_w || !net->vf_b || net->ac_dim < 1 || net->ac_dim > MAXA)
    FAIL(-2, "both heads are needed");
  hipLaunchKernelGGL(ppo_lstm_head_grad_kernel, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, *net, latent, rows, actions,
                     adv, returns, old_neglogp, is_weight, (float)inv_count, cliprange, vf_coef, dlatent, dmean, dvalue, dlogstd_rows, stats);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- one BPTT step (gate order i,f,o,u).  One wave per 16-row tile; LDS: dz [16][4*NH + 2]
template <int NH>
__global__ void __launch_bounds__(256) ppo_lstm_bwd_step_kernel(const float* wh, int n, const float* dlat, const float* mask,
                                                                const float* gates, const float* cprev, const float* tanhc, float* dh_c,
                                                                float* dc_c, float* dz_out) {
  // four waves per 16-row tile: the elementwise part is spread over all 256 threads, the contraction dh_prev = dz * wh^T
  // over k is split in four (wave w contracts k in [w NH, (w+1) NH)) and the partial tiles are summed through LDS in wave
  // order -- a quarter of the dependent weight-load round trips per wave
  constexpr int ZS = 4 * NH + 2, PS = NH;
  float* dzb = smem_f;                 // [16][ZS]
  float* part = smem_f;                // [4][16][PS] partial dh tiles, over dzb once every wave is done reading it
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r0 = blockIdx.x * 16;
  for (int e = tid; e < 16 * NH; e += 256) {
    const int r = e / NH, j = e - r * NH, row = r0 + r;
    float dzi = 0, dzf = 0, dzo = 0, dzu = 0;
    if (row < n) {
      const size_t o = (size_t)row * NH + j;
      const float* g = gates + (size_t)row * 4 * NH;
      const float ig = g[j], fg = g[NH + j], og = g[2 * NH + j], ug = g[3 * NH + j], tc = tanhc[o], cp = cprev[o];
      const float dht = dlat[o] + dh_c[o];
      const float dct = dc_c[o] + dht * og * (1.0f - tc * tc);
      dzi = dct * ug * ig * (1.0f - ig); dzf = dct * cp * fg * (1.0f - fg); dzo = dht * tc * og * (1.0f - og); dzu = dct * ig * (1.0f - ug * ug);
      const float keep = mask ? 1.0f - mask[row] : 1.0f;
      dc_c[o] = dct * fg * keep;
      float* z = dz_out + (size_t)row * 4 * NH;
      z[j] = dzi; z[NH + j] = dzf; z[2 * NH + j] = dzo; z[3 * NH + j] = dzu;
    }
    dzb[r * ZS + j] = dzi; dzb[r * ZS + NH + j] = dzf; dzb[r * ZS + 2 * NH + j] = dzo; dzb[r * ZS + 3 * NH + j] = dzu;
  }
  __syncthreads();
  const int i = lane & 15, kq = lane >> 4;
  constexpr int NT = NH / 16;
  f32x4 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) acc[ct] = (f32x4){0, 0, 0, 0};
  // B[k][col] = wh[col][k]: a lane reads 16 contiguous bytes of its column (k = k0 + 4 kq .. + 3) and feeds them to four
  // k-steps (step r contracts k0 + 4 kq' + r over the four kq'), the next batch's loads in flight meanwhile
  auto fetchw = [&](int k0, float (&av)[4], float (&b)[NT][4]) {
    const int k = k0 + 4 * kq;
#pragma unroll
    for (int r = 0; r < 4; r++) av[r] = dzb[i * ZS + k + r];
#pragma unroll
    for (int ct = 0; ct < NT; ct++) {
      const float4 w4 = *reinterpret_cast<const float4*>(wh + (size_t)(ct * 16 + i) * 4 * NH + k);
      b[ct][0] = w4.x; b[ct][1] = w4.y; b[ct][2] = w4.z; b[ct][3] = w4.w;
    }
  };
  {
    const int kbeg = wid * NH, kend = kbeg + NH;   // NH is a multiple of 32
    float av0[4], av1[4], bw0[NT][4], bw1[NT][4];
    fetchw(kbeg, av0, bw0);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      fetchw(k0 + 16, av1, bw1);
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int ct = 0; ct < NT; ct++) acc[ct] = MFMA(av0[r], bw0[ct][r], acc[ct]);
      if (k0 + 32 < kend) fetchw(k0 + 32, av0, bw0);
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int ct = 0; ct < NT; ct++) acc[ct] = MFMA(av1[r], bw1[ct][r], acc[ct]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int ct = 0; ct < NT; ct++)
#pragma unroll
    for (int r = 0; r < 4; r++) part[(wid * 16 + 4 * kq + r) * PS + ct * 16 + i] = acc[ct][r];
  __syncthreads();
  for (int e = tid; e < 16 * NH; e += 256) {
    const int r = e / NH, j = e - r * NH, row = r0 + r;
    if (row < n) {
      const float sum = ((part[r * PS + j] + part[(16 + r) * PS + j]) + part[(32 + r) * PS + j]) + part[(48 + r) * PS + j];
      dh_c[(size_t)row * NH + j] = sum * (mask ? 1.0f - mask[row] : 1.0f);
    }
  }
}
extern "C" int ppo_lstm_bwd_step(const ppo_lstm_net* net, int n, const float* dlatent_t, const float* mask_t, const float* gates_t,
                                 const float* cprev_t, const float* tanhc_t, float* dh_carry, float* dc_carry, float* dz_out, void* stream) {
  if (!net || !net->wh || !dlatent_t || !gates_t || !cprev_t || !tanhc_t || !dh_carry || !dc_carry || !dz_out || n <= 0) FAIL(-1, "bad arguments");
  if (net->gate_order != PPO_LSTM_GATES_IFOU) FAIL(-2, "backward is built for the (i,f,o,u) gate order only");
  if (net->hidden != 64 && net->hidden != 128) FAIL(-3, "hidden %d: only 64 and 128 are built", net->hidden);
  const int tiles = (n + 15) / 16;
  const size_t lds = (size_t)16 * (4 * net->hidden + 2) * sizeof(float);
  if (net->hidden == 64)
    hipLaunchKernelGGL(ppo_lstm_bwd_step_kernel<64>, dim3(tiles), dim3(256), lds, (hipStream_t)stream, net->wh, n, dlatent_t, mask_t, gates_t,
                       cprev_t, tanhc_t, dh_carry, dc_carry, dz_out);
  else
    hipLaunchKernelGGL(ppo_lstm_bwd_step_kernel<128>, dim3(tiles), dim3(256), lds, (hipStream_t)stream, net->wh, n, dlatent_t, mask_t, gates_t,
                       cprev_t, tanhc_t, dh_carry, dc_carry, dz_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The T steps of the unrolled forward / of BPTT in ONE launch each (hidden 128, gate order i,f,o,u).  A training minibatch is a few
// 16-row tiles of env sequences (128 envs = 8 tiles) whose time steps depend on each other: launched step by step, every step pays a
// launch boundary and re-reads the recurrent weights from L2.  Here a workgroup of EIGHT waves owns a tile for the whole sequence; wave
// w owns the units 16 w .. 16 w + 15 and keeps ITS SLICE OF wh IN REGISTERS for all T steps (forward: the 4 x 32 B operands of its four
// gate tiles, backward: the 128 B operands of its output tile of dz * wh^T; 128 floats per lane either way), the cell / delta carries
// live in registers in the MFMA D layout (lane (i, kq): rows 4 kq + r, unit 16 w + i), the latent / the deltas cross the waves
// through a double-buffered LDS tile (one barrier per step), and the next step's global operands are requested during the products.
// ---------------------------------------------------------------------------------------------------------
struct LstmSeqArgs {
  const float *wh, *b; float forget_bias;
  const float* z0;        // [T][n][512] input sums (ppo_lstm_xproj)
  const float* mask;      // [T][n]
  float *c, *h; int state_stride;
  float *sv_gates, *sv_cprev, *sv_hprev, *sv_tanhc, *latent;   // [T][n][...]
  const float* dlat;      // backward: [T][n][128]
  float* dz;              // backward: [T][n][512]
  int T, n;
};

__global__ void __launch_bounds__(512) ppo_lstm_seq_fwd_kernel(LstmSeqArgs a) {
  constexpr int NH = 128, HP = NH + 4;
  float* hbuf = smem_f;                // [2][16][HP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r0 = blockIdx.x * 16;
  const int i = lane & 15, kq = lane >> 4, j = 16 * w + i;
  // this lane's B operands: k-step s (k = 4 s + kq), gate g, column g * NH + j
  float W[32][4];
#pragma unroll
  for (int s_ = 0; s_ < 32; s_++)
#pragma unroll
    for (int g = 0; g < 4; g++) W[s_][g] = a.wh[(size_t)(4 * s_ + kq) * 4 * NH + g * NH + j];
  const float bi = a.b[j], bf = a.b[NH + j] + a.forget_bias, bo = a.b[2 * NH + j], bu = a.b[3 * NH + j];
  float c[4], keep[4];
  f32x4 z[4];
  // (plain loads at clamped rows, masks applied where the values are consumed: with a select right behind each load the compiler
  // predicates the loads one by one and waits after each -- 28 serial trips to memory per time step instead of a prefetch)
  auto load_z = [&](int t, f32x4 (&zz)[4], float (&kp)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      const int rowc = row < a.n ? row : a.n - 1;
      kp[r] = a.mask ? a.mask[(size_t)t * a.n + rowc] : 0.0f;          // the done flag itself; 1 - flag in finish_z
#pragma unroll
      for (int g = 0; g < 4; g++) zz[g][r] = a.z0[((size_t)t * a.n + rowc) * 4 * NH + g * NH + j];
    }
  };
  auto finish_z = [&](f32x4 (&zz)[4], float (&kp)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool ok = r0 + 4 * kq + r < a.n;
      kp[r] = ok ? 1.0f - kp[r] : 1.0f;
#pragma unroll
      for (int g = 0; g < 4; g++) zz[g][r] = ok ? zz[g][r] : 0.0f;
    }
  };
  load_z(0, z, keep);
  finish_z(z, keep);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = r0 + 4 * kq + r;
    float h0 = 0.0f;
    c[r] = 0.0f;
    if (row < a.n) { c[r] = a.c[(size_t)row * a.state_stride + j]; h0 = a.h[(size_t)row * a.state_stride + j]; }
    const float hv = h0 * keep[r];
    hbuf[(4 * kq + r) * HP + (j & 3) * (NH / 4) + (j >> 2)] = hv;   // k-step order: unit k = 4 s + kq sits at kq * 32 + s (128-bit A-operand reads)
    if (row < a.n) a.sv_hprev[(size_t)row * NH + j] = hv;
  }
  float hn_last[4] = {0, 0, 0, 0};
  for (int t = 0; t < a.T; t++) {
    const float* hb = hbuf + (t & 1) * 16 * HP;
    float* hb_next = hbuf + ((t + 1) & 1) * 16 * HP;
    __syncthreads();   // the masked previous latent of every unit is in hb
    f32x4 zn[4];
    float keepn[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (t + 1 < a.T) load_z(t + 1, zn, keepn);   // in flight during the products
    const float4* ha = (const float4*)(hb + i * HP + kq * (NH / 4));
#pragma unroll
    for (int v = 0; v < 8; v++) {
      const float4 q = ha[v];
      const float av4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int g = 0; g < 4; g++) z[g] = MFMA(av4[u], W[4 * v + u][g], z[g]);
    }
    if (t + 1 < a.T) finish_z(zn, keepn);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      const float cp = c[r] * keep[r];
      const LstmCell cell = lstm_cell(z[0][r], z[1][r], z[2][r], z[3][r], bi, bf, bo, bu, cp);
      c[r] = cell.cn; hn_last[r] = cell.hn;
      const float hv = cell.hn * keepn[r];              // what the next step reads: masked by ITS done flag
      hb_next[(4 * kq + r) * HP + (j & 3) * (NH / 4) + (j >> 2)] = hv;
      if (row < a.n) {
        const size_t o = ((size_t)t * a.n + row) * NH + j;
        float* sg = a.sv_gates + ((size_t)t * a.n + row) * 4 * NH;
        sg[j] = cell.ig; sg[NH + j] = cell.fg; sg[2 * NH + j] = cell.og; sg[3 * NH + j] = cell.ug;
        a.sv_cprev[o] = cp; a.sv_tanhc[o] = cell.tcn; a.latent[o] = cell.hn;
        if (t + 1 < a.T) a.sv_hprev[o + (size_t)a.n * NH] = hv;
      }
    }
    if (t + 1 < a.T) {
#pragma unroll
      for (int g = 0; g < 4; g++) z[g] = zn[g];
#pragma unroll
      for (int r = 0; r < 4; r++) keep[r] = keepn[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = r0 + 4 * kq + r;
    if (row < a.n) { a.c[(size_t)row * a.state_stride + j] = c[r]; a.h[(size_t)row * a.state_stride + j] = hn_last[r]; }
  }
}

__global__ void __launch_bounds__(512) ppo_lstm_seq_bwd_kernel(LstmSeqArgs a) {
  constexpr int NH = 128, ZS = 4 * NH + 4;
  float* dzb = smem_f;                 // [2][16][ZS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r0 = blockIdx.x * 16;
  const int i = lane & 15, kq = lane >> 4, j = 16 * w + i;
  // B operands of dh_prev = dz * wh^T for this wave's output tile: B[k = column][unit j] = wh[j][column], k-step s: column 4 s + kq
  float W[128];
#pragma unroll
  for (int s_ = 0; s_ < 128; s_++) W[s_] = a.wh[(size_t)j * 4 * NH + 4 * s_ + kq];
  float dh[4] = {0, 0, 0, 0}, dc[4] = {0, 0, 0, 0};
  struct In { float ig, fg, og, ug, tc, cp, dl, keep; };
  auto load_in = [&](int t, In (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      v[r] = In{0, 0, 0, 0, 0, 0, 0, 1.0f};
      if (row < a.n) {
        const size_t o = ((size_t)t * a.n + row) * NH + j;
        const float* g = a.sv_gates + ((size_t)t * a.n + row) * 4 * NH;
        v[r].ig = g[j]; v[r].fg = g[NH + j]; v[r].og = g[2 * NH + j]; v[r].ug = g[3 * NH + j];
        v[r].tc = a.sv_tanhc[o]; v[r].cp = a.sv_cprev[o]; v[r].dl = a.dlat[o];
        if (a.mask) v[r].keep = 1.0f - a.mask[(size_t)t * a.n + row];
      }
    }
  };
  In cur[4], nxt[4];
  load_in(a.T - 1, cur);
  for (int t = a.T - 1; t >= 0; t--) {
    float* zb = dzb + (t & 1) * 16 * ZS;
    if (t > 0) load_in(t - 1, nxt);      // in flight during this step
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 4 * kq + r;
      const In& v = cur[r];
      float dzi = 0, dzf = 0, dzo = 0, dzu = 0;
      if (row < a.n) {
        const float dht = v.dl + dh[r];
        const float dct = dc[r] + dht * v.og * (1.0f - v.tc * v.tc);
        dzi = dct * v.ug * v.ig * (1.0f - v.ig); dzf = dct * v.cp * v.fg * (1.0f - v.fg); dzo = dht * v.tc * v.og * (1.0f - v.og);
        dzu = dct * v.ig * (1.0f - v.ug * v.ug);
        dc[r] = dct * v.fg * v.keep;
        float* zo = a.dz + ((size_t)t * a.n + row) * 4 * NH;
        zo[j] = dzi; zo[NH + j] = dzf; zo[2 * NH + j] = dzo; zo[3 * NH + j] = dzu;
      }
      // LDS tile in k-step order: column c = 4 s + kq of the delta row sits at kq * 128 + s, so that the 128 A operands of a lane are 32
      // consecutive 128-bit words (the products keep their order; a quarter of the LDS read instructions: 864 -> 812 us)
      float* zr = zb + (4 * kq + r) * ZS + (j & 3) * NH + (j >> 2);
      zr[0] = dzi; zr[NH / 4] = dzf; zr[2 * (NH / 4)] = dzo; zr[3 * (NH / 4)] = dzu;
    }
    __syncthreads();   // the tile's deltas of step t are in zb (the other buffer is free again: its readers passed this barrier)
    f32x4 acc0 = (f32x4){0, 0, 0, 0}, acc1 = (f32x4){0, 0, 0, 0};
    const float4* za = (const float4*)(zb + i * ZS + kq * NH);
#pragma unroll
    for (int v = 0; v < 32; v++) {
      const float4 q = za[v];
      acc0 = MFMA(q.x, W[4 * v], acc0); acc1 = MFMA(q.y, W[4 * v + 1], acc1);
      acc0 = MFMA(q.z, W[4 * v + 2], acc0); acc1 = MFMA(q.w, W[4 * v + 3], acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) dh[r] = (acc0[r] + acc1[r]) * cur[r].keep;
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) cur[r] = nxt[r];
    }
  }
}

static int lstm_seq_check(const ppo_lstm_net* net, int T, int n) {
  if (!net || !net->wh || !net->b || T <= 0 || n <= 0) FAIL(-1, "bad arguments");
  if (net->hidden != 128 || net->gate_order != PPO_LSTM_GATES_IFOU) FAIL(-2, "sequence kernels: hidden 128, gate order i,f,o,u (got %d, %d)", net->hidden, net->gate_order);
  return 0;
}
extern "C" int ppo_lstm_seq_forward(const ppo_lstm_net* net, const float* z0, int T, int n, const float* mask, float* c, float* h, int state_stride,
                                    float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc, float* latent, void* stream) {
  if (int rc = lstm_seq_check(net, T, n)) return rc;
  if (!z0 || !c || !h || !save_gates || !save_cprev || !save_hprev || !save_tanhc || !latent || state_stride < 128) FAIL(-1, "bad arguments");
  LstmSeqArgs a = LstmSeqArgs();
  a.wh = net->wh; a.b = net->b; a.forget_bias = net->forget_bias; a.z0 = z0; a.mask = mask; a.c = c; a.h = h; a.state_stride = state_stride;
  a.sv_gates = save_gates; a.sv_cprev = save_cprev; a.sv_hprev = save_hprev; a.sv_tanhc = save_tanhc; a.latent = latent; a.T = T; a.n = n;
  const size_t lds = (size_t)2 * 16 * (128 + 4) * sizeof(float);
  hipLaunchKernelGGL(ppo_lstm_seq_fwd_kernel, dim3((n + 15) / 16), dim3(512), lds, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int ppo_lstm_seq_backward(const ppo_lstm_net* net, int T, int n, const float* dlatent, const float* mask, const float* gates,
                                     const float* cprev, const float* tanhc, float* dz_out, void* stream) {
  if (int rc = lstm_seq_check(net, T, n)) return rc;
  if (!dlatent || !gates || !cprev || !tanhc || !dz_out) FAIL(-1, "bad arguments");
  LstmSeqArgs a = LstmSeqArgs();
  a.wh = net->wh; a.mask = mask; a.sv_gates = const_cast<float*>(gates); a.sv_cprev = const_cast<float*>(cprev); a.sv_tanhc = const_cast<float*>(tanhc);
  a.dlat = dlatent; a.dz = dz_out; a.T = T; a.n = n;
  const size_t lds = (size_t)2 * 16 * (4 * 128 + 4) * sizeof(float);
  static thread_local bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)ppo_lstm_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(ppo_lstm_seq_bwd_kernel, dim3((n + 15) / 16), dim3(512), lds, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradients of the recurrent PPO step: C = A'^T B over all (time, env) rows of the minibatch, A' = [A0 | A1 | 1]
// (input, previous hidden state, ones column for the bias), B = [B0 | B1 | B2] (gate / head deltas).  The contraction runs
// over the ROWS (thousands), the output is small (250 x 512): split-K over workgroups -- every workgroup owns a chunk of rows,
// each of its four waves a block of MT x NT output tiles held in MFMA accumulators (v_mfma_f32_16x16x4_f32: A operand =
// 4 rows x 16 columns of A', B operand = 4 rows x 16 columns of B, both coalesced row reads) -- and one slab per chunk is
// reduced in fixed order by a second kernel (deterministic, no float atomics: same scheme as ppo_grad).
// ---------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* a_ptr[2]; int a_ld[2], a_cols[2]; int a_ones;   // A' columns: segment 0 | segment 1 | optional column of ones
  const float* b_ptr[3]; int b_ld[3], b_cols[3];               // B columns: segment 0 | 1 | 2
  int rows, rows_per_chunk, Mpad, Npad;
  float* slabs;                                                // [chunks][Mpad][Npad]
};

// (the segment tables are kernel arguments: with the segment loop unrolled over a compile-time count they are read as scalars; a
// run-time loop indexed them per lane through memory, two dependent vector loads per column before the first product)
template <int NSEG>
__device__ __forceinline__ void wgrad_col(const float* const (&ptr)[NSEG], const int (&ld)[NSEG], const int (&cols)[NSEG], int ones, int col,
                                          const float*& base, int& stride, float& cval) {
  base = nullptr; stride = 0; cval = 0.0f;
  int c = col;
  bool done = false;
#pragma unroll
  for (int sgm = 0; sgm < NSEG; sgm++) {
    const bool here = !done && c < cols[sgm];
    if (here && ptr[sgm]) { base = ptr[sgm] + c; stride = ld[sgm]; }
    if (!done && !here) c -= cols[sgm];
    done = done || here;
  }
  if (!done && ones && c == 0) cval = 1.0f;
}

template <int MT, int NT>
__global__ void __launch_bounds__(256) ppo_wgrad_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
  const int mt0 = (blockIdx.y * 4 + wid) * MT, nt0 = blockIdx.z * NT;
  const int r_lo = blockIdx.x * a.rows_per_chunk, r_hi = min(a.rows, r_lo + a.rows_per_chunk);
  const float* ap[MT]; int as[MT]; float ac[MT];
  const float* bp[NT]; int bs[NT]; float bc[NT];
#pragma unroll
  for (int m = 0; m < MT; m++) wgrad_col<2>(a.a_ptr, a.a_ld, a.a_cols, a.a_ones, (mt0 + m) * 16 + i, ap[m], as[m], ac[m]);
#pragma unroll
  for (int n = 0; n < NT; n++) wgrad_col<3>(a.b_ptr, a.b_ld, a.b_cols, 0, (nt0 + n) * 16 + i, bp[n], bs[n], bc[n]);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < NT; n++) acc[m][n] = (f32x4){0, 0, 0, 0};
  // operand columns that are constants (bias ones, padding) read a valid dummy address with stride 0: every load of a trip is a plain
  // unconditional load at a clamped row and the selects come afterwards -- with `ok ? p[row] : 0` the compiler predicated the loads
  // one by one and waited after each (all 24 loads of a trip serial; tools: the serialized-load count of the disassembly)
  const float PT_GAS* apx[MT]; const float PT_GAS* bpx[NT];   // (global address space: plain global loads, not flat ones)
#pragma unroll
  for (int m = 0; m < MT; m++) { apx[m] = pt_global(ap[m] ? ap[m] : a.slabs); if (!ap[m]) as[m] = 0; }   // (any readable address: the value is discarded)
#pragma unroll
  for (int n = 0; n < NT; n++) { bpx[n] = pt_global(bp[n] ? bp[n] : a.slabs); if (!bp[n]) bs[n] = 0; }
  // two k-steps (8 rows) per trip; the NEXT trip's 2 (MT + NT) operand loads are issued before this trip's products (double-buffered)
  float av[2][2][MT], bv[2][2][NT];
  auto load_trip = [&](int buf, int r0) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int row = r0 + 4 * u + kq;
      const int rowc = row < r_hi ? row : r_hi - 1;
#pragma unroll
      for (int m = 0; m < MT; m++) av[buf][u][m] = apx[m][(size_t)rowc * as[m]];
#pragma unroll
      for (int n = 0; n < NT; n++) bv[buf][u][n] = bpx[n][(size_t)rowc * bs[n]];
    }
  };
  auto trip = [&](int buf, int r0) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const bool ok = r0 + 4 * u + kq < r_hi;
      float x[MT], y[NT];
#pragma unroll
      for (int m = 0; m < MT; m++) x[m] = ok ? (ap[m] ? av[buf][u][m] : ac[m]) : 0.0f;
#pragma unroll
      for (int n = 0; n < NT; n++) y[n] = ok ? (bp[n] ? bv[buf][u][n] : bc[n]) : 0.0f;
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = MFMA(x[m], y[n], acc[m][n]);
    }
  };
  if (r_lo < r_hi) load_trip(0, r_lo);
  for (int r0 = r_lo; r0 < r_hi; r0 += 16) {     // (two trips per loop iteration so that the buffer index is a literal: registers, not scratch)
    if (r0 + 8 < r_hi) load_trip(1, r0 + 8);
    __builtin_amdgcn_sched_barrier(0);           // loads first: the scheduler otherwise sinks each load to its use and waits there
    trip(0, r0);
    if (r0 + 8 < r_hi) {
      if (r0 + 16 < r_hi) load_trip(0, r0 + 16);
      __builtin_amdgcn_sched_barrier(0);
      trip(1, r0 + 8);
    }
  }
  // D layout: lane (i, kq) holds rows 4 kq + r (A' column index within the tile), column i (B column within the tile)
  float* slab = a.slabs + (size_t)blockIdx.x * a.Mpad * a.Npad;
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = (mt0 + m) * 16 + 4 * kq + r, col = (nt0 + n) * 16 + i;
        if (row < a.Mpad && col < a.Npad) slab[(size_t)row * a.Npad + col] = acc[m][n][r];
      }
}

// fixed-order sum of the chunk slabs.  mode 0: dst[m][n] (leading dimension N) for m < M, n < N -- the LSTM's wx | wh | b
// gradients are exactly the rows of [x | h_prev | 1]^T dz.  mode 1: the heads' gradients scattered into checkpoint order
// pi/w [H][A] | pi/b [A] | logstd [A] | vf/w [H] | vf/b [1] from [latent | 1]^T [dmean (A) | dvalue (1) | dlogstd rows (A)];
// `logstd_shift` (the entropy term, - ent_coef / world) is added to the logstd gradient.
__global__ void __launch_bounds__(256) ppo_wgrad_reduce_kernel(const float* slabs, int chunks, int Mpad, int Npad, int M, int N, int mode, int Hh,
                                                               int Aa, float logstd_shift, float* dst) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N) return;
  const int m = t / N, n = t - m * N;
  float s = 0.0f;
  for (int c = 0; c < chunks; c++) s += slabs[((size_t)c * Mpad + m) * Npad + n];
  if (mode == 0) { dst[(size_t)m * N + n] = s; return; }
  const int o_pw = 0, o_pb = Hh * Aa, o_ls = o_pb + Aa, o_vw = o_ls + Aa, o_vb = o_vw + Hh;
  if (m < Hh) {
    if (n < Aa) dst[o_pw + m * Aa + n] = s;
    else if (n == Aa) dst[o_vw + m] = s;
  } else {           // the ones row: column sums
    if (n < Aa) dst[o_pb + n] = s;
    else if (n == Aa) dst[o_vb] = s;
    else dst[o_ls + (n - Aa - 1)] = s + logstd_shift;
  }
}

extern "C" size_t ppo_lstm_wgrad_workspace_bytes(int ob_dim, int hidden, int ac_dim) {
  const size_t Mp = (size_t)((ob_dim + hidden + 1 + 63) / 64) * 64, Np = (size_t)4 * hidden;
  return (size_t)64 * Mp * Np * sizeof(float);     // 64 chunks at most; the head problem is smaller and reuses the buffer
}

extern "C" int ppo_lstm_wgrad(const ppo_lstm_net* net, int rows, const float* x, const float* hprev, const float* dz, const float* latent,
                              const float* dmean, const float* dvalue, const float* dlogstd_rows, float logstd_shift, float* grads,
                              void* workspace, void* stream) {
  if (!net || !x || !hprev || !dz || !latent || !dmean || !dvalue || !dlogstd_rows || !grads || !workspace || rows <= 0) FAIL(-1, "bad arguments");
  if (net->emb_w) FAIL(-2, "the baselines LSTM has no embedding layer");
  const int D = net->ob_dim, Hh = net->hidden, Aa = net->ac_dim;
  hipStream_t st = (hipStream_t)stream;
  int chunks = (rows + 255) / 256;
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  int rpc = ((rows + chunks - 1) / chunks + 7) & ~7;
  chunks = (rows + rpc - 1) / rpc;
  {  // [x | h_prev | 1]^T dz  ->  wx | wh | b  (the first (D + H + 1) * 4H entries of the flat gradient, checkpoint order)
    WgradArgs a = WgradArgs();
    a.a_ptr[0] = x; a.a_ld[0] = D; a.a_cols[0] = D; a.a_ptr[1] = hprev; a.a_ld[1] = Hh; a.a_cols[1] = Hh; a.a_ones = 1;
    a.b_ptr[0] = dz; a.b_ld[0] = 4 * Hh; a.b_cols[0] = 4 * Hh;
    const int M = D + Hh + 1, N = 4 * Hh;
    a.rows = rows; a.rows_per_chunk = rpc; a.Mpad = ((M + 63) / 64) * 64; a.Npad = N; a.slabs = (float*)workspace;
    hipLaunchKernelGGL((ppo_wgrad_kernel<4, 8>), dim3(chunks, (a.Mpad + 255) / 256, (N + 127) / 128), dim3(256), 0, st, a);
    hipLaunchKernelGGL(ppo_wgrad_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, a.slabs, chunks, a.Mpad, a.Npad, M, N, 0, Hh, Aa, 0.0f, grads);
  }
  {  // [latent | 1]^T [dmean | dvalue | dlogstd rows]  ->  pi/w | pi/b | logstd | vf/w | vf/b
    WgradArgs a = WgradArgs();
    a.a_ptr[0] = latent; a.a_ld[0] = Hh; a.a_cols[0] = Hh; a.a_cols[1] = 0; a.a_ones = 1;
    a.b_ptr[0] = dmean; a.b_ld[0] = Aa; a.b_cols[0] = Aa; a.b_ptr[1] = dvalue; a.b_ld[1] = 1; a.b_cols[1] = 1;
    a.b_ptr[2] = dlogstd_rows; a.b_ld[2] = Aa; a.b_cols[2] = Aa;
    const int M = Hh + 1, N = 2 * Aa + 1;
    a.rows = rows; a.rows_per_chunk = rpc; a.Mpad = ((M + 63) / 64) * 64; a.Npad = ((N + 31) / 32) * 32; a.slabs = (float*)workspace;
    hipLaunchKernelGGL((ppo_wgrad_kernel<4, 2>), dim3(chunks, (a.Mpad + 255) / 256, a.Npad / 32), dim3(256), 0, st, a);
    hipLaunchKernelGGL(ppo_wgrad_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, a.slabs, chunks, a.Mpad, a.Npad, M, N, 1, Hh, Aa, logstd_shift,
                       grads + (size_t)(D + Hh + 1) * 4 * Hh);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// rollout arithmetic
// ---------------------------------------------------------------------------------------------------------
__global__ void ppo_reward_mix_kernel(const double* info, int n, double alpha, float* out, int agent_stride) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n) return;
  int e = t >> 1, g = t & 1;
  const double* I = info + (size_t)(2 * e + g) * 8;
  out[(size_t)g * agent_stride + e] = reward_mix(alpha, I[6], I[3]);
}
// reward mix plus the episode records of the step (monitor.py:63-78 harvest): one launch instead of four
__global__ void ppo_post_step_kernel(const double* info, int n, double alpha, float* out, int agent_stride, const uint8_t* done,
                                     const double* ep_r, const int32_t* ep_l, uint8_t* ep_done_out, double* ep_r_out, int32_t* ep_l_out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n) return;
  int e = t >> 1, g = t & 1;
  const double* I = info + (size_t)(2 * e + g) * 8;
  out[(size_t)g * agent_stride + e] = reward_mix(alpha, I[6], I[3]);
  if (g == 0) { ep_done_out[e] = done[2 * e]; ep_r_out[e] = ep_r[e]; ep_l_out[e] = ep_l[e]; }
}
extern "C" int ppo_post_step(const double* info, int n, double alpha, float* reward_out, int agent_stride, const uint8_t* done,
                             const double* ep_r, const int32_t* ep_l, uint8_t* ep_done_out, double* ep_r_out, int32_t* ep_l_out,
                             void* stream) {
  if (!info || !reward_out || !done || !ep_r || !ep_l || !ep_done_out || !ep_r_out || !ep_l_out || n <= 0) FAIL(-1, "bad arguments");
  hipLaunchKernelGGL(ppo_post_step_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, info, n, alpha, reward_out,
                     agent_stride, done, ep_r, ep_l, ep_done_out, ep_r_out, ep_l_out);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int ppo_reward_mix(const double* info, int n, double alpha, float* reward_out, int agent_stride, void* stream) {
  if (!info || !reward_out || n <= 0) FAIL(-1, "bad arguments");
  hipLaunchKernelGGL(ppo_reward_mix_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, info, n, alpha, reward_out,
                     agent_stride);
  HIPCHK(hipGetLastError());
  return 0;
}

// one thread per env; reverse scan over T for both agents.  Mixed precision follows numpy's evaluation of
// runner.py:186-190: gamma*nextvalues in float32, everything else promoted to float64, result stored as float32.
__global__ void ppo_vtrace_kernel(const float* rew, const float* val, const float* nlp, const float* onlp, const uint8_t* dones,
                                  const uint8_t* last_dones, const float* last_values, int T, int N, double gamma, float lam,
                                  float rho_bar, float c_bar, float* ret, float* off_policy, float* off_env, float* ratio_out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const size_t TN = (size_t)T * N;
  const float g32 = (float)gamma;
  double acc0 = 0, acc1 = 0;
  for (int t = T - 1; t >= 0; t--) {
    size_t o = (size_t)t * N + e;
    float opr = expf(onlp[TN + o] - nlp[TN + o]);   // runner.py:170
    float oer = expf(nlp[o] - onlp[o]);             // runner.py:171
    float rt = opr * oer;
    off_policy[o] = opr; off_env[o] = oer; ratio_out[o] = rt;
    float rho1 = fminf(rt, rho_bar), c1 = fminf(rt, c_bar) * lam;
    if (rt != rt) { rho1 = rt; c1 = rt; }           // np.clip propagates NaN
    float c0 = 1.0f * lam;
    for (int g = 0; g < 2; g++) {
      double nnt;
      float nextv;
      if (t == T - 1) { nnt = 1.0 - (double)(last_dones[2 * e + g] != 0); nextv = last_values[(size_t)g * N + e]; }
      else { nnt = 1.0 - (double)(dones[g * TN + o + N] != 0); nextv = val[g * TN + o + N]; }
      float rho = g == 0 ? 1.0f : rho1, c = g == 0 ? c0 : c1;
      double v = (double)val[g * TN + o];
      double delta = (double)rho * (((double)rew[g * TN + o] + (double)(g32 * nextv) * nnt) - v);
      double& acc = g == 0 ? acc0 : acc1;
      acc = delta + ((gamma * nnt) * (double)c) * acc;
      ret[g * TN + o] = (float)(v + acc);
    }
  }
}
extern "C" int ppo_vtrace(const float* rewards, const float* values, const float* neglogp, const float* opp_neglogp,
                          const uint8_t* dones, const uint8_t* last_dones, const float* last_values, int T, int N, double gamma,
                          double lam, double rho_bar, double c_bar, float* returns, float* off_policy_ratio, float* off_env_ratio,
                          float* ratio, void* stream) {
  if (!rewards || !values || !neglogp || !opp_neglogp || !dones || !last_dones || !last_values || !returns || !off_policy_ratio ||
      !off_env_ratio || !ratio || T <= 0 || N <= 0)
    FAIL(-1, "bad arguments");
  hipLaunchKernelGGL(ppo_vtrace_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, rewards, values, neglogp, opp_neglogp,
                     dones, last_dones, last_values, T, N, gamma, (float)lam, (float)rho_bar, (float)c_bar, returns, off_policy_ratio,
                     off_env_ratio, ratio);
  HIPCHK(hipGetLastError());
  return 0;
}

// Advantage moments of a minibatch: blocks of 1024 threads sum grid-strided 4096-row chunks, write their partials (sc1, visible to every
// XCD) into the CALLER's workspace and count themselves in; the last block adds the partials in block order (deterministic).  The
// workspace (ppo_adv_moments_workspace_bytes; its arrival counter zero-initialised once, left at zero by every call) belongs to the
// caller -- one model, one stream -- so two models stepping on different streams do not share state (round 2 kept it in module-level
// device memory).  Any n: a block takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ...
#define ADV_MAX_BLOCKS 256
struct AdvWs { double part[2 * ADV_MAX_BLOCKS]; unsigned int arrived; unsigned int pad; };
__global__ void __launch_bounds__(1024) ppo_adv_moments_kernel(const float* ret, const float* val, const int32_t* idx, int n, double* mom, AdvWs* ws) {
  __shared__ double s1[16], s2[16];
  __shared__ int last;
  double a = 0, b = 0;
  for (int base = blockIdx.x * 4096; base < n; base += gridDim.x * 4096) {
    int r[4];
    float x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int k = base + threadIdx.x + 1024 * u;
      r[u] = k < n ? (idx ? idx[k] : k) : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      x[u] = r[u] >= 0 ? ret[r[u]] : 0.f;
      y[u] = r[u] >= 0 ? val[r[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double d = (double)(x[u] - y[u]);  // float32 subtraction first, as numpy does
      a += d; b += d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s1[w] = a; s2[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
    for (int i = 0; i < 16; i++) { ta += s1[i]; tb += s2[i]; }
    __hip_atomic_store(&ws->part[2 * blockIdx.x], ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1: visible to every XCD
    __hip_atomic_store(&ws->part[2 * blockIdx.x + 1], tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int arrived = __hip_atomic_fetch_add(&ws->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (arrived == gridDim.x - 1);
    if (last) {
      double sa = 0, sb = 0;
      for (unsigned int i0 = 0; i0 < gridDim.x; i0 += 8) {   // eight pairs in flight at a time, summed in their fixed order
        double va[8], vb[8];
#pragma unroll
        for (unsigned int k = 0; k < 8; k++) {
          const unsigned int i = i0 + k < gridDim.x ? i0 + k : i0;
          va[k] = __hip_atomic_load(&ws->part[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          vb[k] = __hip_atomic_load(&ws->part[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (unsigned int k = 0; k < 8; k++)
          if (i0 + k < gridDim.x) { sa += va[k]; sb += vb[k]; }
      }
      mom[0] = sa; mom[1] = sb; mom[2] = (double)n;
      __hip_atomic_store(&ws->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                    // ready for the next call
    }
  }
}
__global__ void ppo_adv_normalize_kernel(const float* ret, const float* val, const int32_t* idx, int n, const double* mom, float* out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double cnt = mom[2], mean = mom[0] / cnt, var = mom[1] / cnt - mean * mean;
  if (var < 0) var = 0;
  float m32 = (float)mean, s32 = (float)sqrt(var);
  int r = idx ? idx[k] : k;
  out[k] = ((ret[r] - val[r]) - m32) / (s32 + 1e-8f);
}
extern "C" size_t ppo_adv_moments_workspace_bytes(void) { return sizeof(AdvWs); }
extern "C" int ppo_adv_moments_ws(const float* returns, const float* values, const int32_t* idx, int n, double* moments, void* workspace,
                                  void* stream) {
  if (!returns || !values || !moments || !workspace || n <= 0) FAIL(-1, "bad arguments");
  int nb = (n + 4095) / 4096;
  if (nb > ADV_MAX_BLOCKS) nb = ADV_MAX_BLOCKS;     // larger minibatches: the blocks stride over the chunks
  hipLaunchKernelGGL(ppo_adv_moments_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, returns, values, idx, n, moments, (AdvWs*)workspace);
  HIPCHK(hipGetLastError());
  return 0;
}
// convenience form without a caller workspace: library-owned workspaces handed out round robin (16 of them), so calls in flight on
// different streams do not meet unless more than 16 overlap; a caller that steps several models concurrently passes its own (…_ws)
extern "C" int ppo_adv_moments(const float* returns, const float* values, const int32_t* idx, int n, double* moments, void* stream) {
  static AdvWs* pool = nullptr;
  static unsigned int next = 0;
  if (!pool) {
    HIPCHK(hipMalloc((void**)&pool, 16 * sizeof(AdvWs)));
    HIPCHK(hipMemset(pool, 0, 16 * sizeof(AdvWs)));
  }
  return ppo_adv_moments_ws(returns, values, idx, n, moments, pool + (next++ & 15), stream);
}
extern "C" int ppo_adv_normalize(const float* returns, const float* values, const int32_t* idx, int n, const double* moments,
                                 float* adv_out, void* stream) {
  if (!returns || !values || !moments || !adv_out || n <= 0) FAIL(-1, "bad arguments");
  hipLaunchKernelGGL(ppo_adv_normalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, returns, values, idx, n, moments,
                     adv_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// training: loss + gradients
// ---------------------------------------------------------------------------------------------------------
struct GradArgs {
  const float *params, *obs, *actions, *adv, *returns, *oldnlp, *weight;
  const int32_t* idx;
  float* slabs;       // [2][nwaves][P]   (net, wave, param) -- only the net's own ranges are written
  double* wstats;     // [2][nwaves][8]
  float* log_ratio;
  int n, obs_stride, XS, nwaves;
  float inv_count, cliprange, ent_coef, vf_coef;
  ParamLayout L;
};

// ---------------------------------------------------------------------------------------------------------
// Gradient kernel: FOUR waves share a 16-row tile; wave w owns the hidden units 16w .. 16w+15 of both hidden layers.
//   * the wave's slices of every weight matrix it multiplies by -- W0[:, slice], W1[:, slice], W1[slice, :]^T, W2 (head) and
//     W2[slice, :]^T -- stay in registers as MFMA B operands for the whole launch: no weight load inside the tile loop;
//   * activations and deltas cross the waves through one set of LDS tiles (four block barriers per tile);
//   * the wave accumulates ITS columns of the weight gradients (gW0[:, slice], gW1[:, slice]; rows 16w.. of gW2) in MFMA
//     accumulators over all tiles of the block and the block writes ONE slab at the end (a quarter of the slab traffic of the
//     wave-per-tile form, a quarter of its per-tile dependent chain: 133 instead of 484 MFMAs per tile and wave);
//   * the next tile's rows are requested (gathered through idx) before the backward half and committed to the other half of a
//     double-buffered observation tile after it.
// The head (16 MFMAs) and the loss deltas are evaluated redundantly by every wave -- cheaper than a fifth barrier with three
// waves idle -- and only wave 0 keeps the statistics, the head-bias and the logstd gradients.
// (round 2's kernel: one wave per tile with the whole gradient of a net in its accumulators, 110 us per 16 384-row minibatch)
// ---------------------------------------------------------------------------------------------------------
#ifdef PPO_GRAD_PROBE   /* development build: shader-clock cycles per phase of workgroup (0, policy net), wave 0 (tools/grad_probe.py) */
__device__ unsigned long long g_gprobe[16];
#define GP_INIT() unsigned long long gp_t = __builtin_amdgcn_s_memtime(), gp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gp_acc[k] += t_ - gp_t; gp_t = t_; } while (0)
#define GP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GP_FLUSH() do { if (PI && blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 8; k_++) g_gprobe[k_] = gp_acc[k_]; } while (0)
#else
#define GP_INIT() do { } while (0)
#define GP(k) do { } while (0)
#define GP_DRAIN() do { } while (0)
#define GP_FLUSH() do { } while (0)
#endif
template <int R>
__device__ __forceinline__ float quad_bcast(float v) {   // the value of lane R of each quad of lanes, in all four
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), R * 0x55, 0xf, 0xf, false));
}
#define GRAD_HS 68   /* activation-tile row stride of the gradient kernel (floats): 16-byte aligned rows, 128-bit reads conflict free */
template <int KT, bool PI>
__device__ __forceinline__ void grad_net_coop(const GradArgs& a, float* lds, int lane, int w) {
  const ParamLayout& L = a.L;
  const int XS = a.XS, D = L.D, A = L.A;
  const int i = lane & 15, kq = lane >> 4, tid = threadIdx.x;
  GP_INIT();
  constexpr int K0 = 4 * KT;   // k-steps of the first layer (four features per step)
  // Activation tiles: row stride GHS = 68 floats (16-byte aligned rows whose 128-bit reads are bank-conflict free: 68 = 4 mod 32).
  // A-operand convention of this kernel: in k-step u the lane (i, kq) supplies feature kq * (K / 4) + u of row i -- each kq group
  // covers a CONTIGUOUS quarter of the contraction range -- so a lane's A operands of a whole layer are K / 4 consecutive floats and
  // come in as 128-bit LDS reads (4 per 64-wide layer instead of 16 dword reads); the resident B operands use the same permutation.
  constexpr int GHS = GRAD_HS;
  float *xb0 = lds, *xb1 = xb0 + 16 * XS, *h1 = xb1 + 16 * XS, *h2 = h1 + 16 * GHS, *d1 = h2 + 16 * GHS, *d2 = d1 + 16 * GHS;
  float* mydout = d2 + 16 * GHS + w * 16 * 18;   // this wave's copy of the head deltas [16][18]
  float *rd0 = d2 + 16 * GHS + 4 * 16 * 18, *rd1 = rd0 + 16 * 20;   // per-row loss inputs of the tile (double-buffered like the rows)
  float4* w2s = (float4*)(rd1 + 16 * 20);   // the head's B operands [4][64 lanes] (the same for every wave): read per tile, 16 registers less
  const Net net = PI ? pi_net(a.params, L) : vf_net(a.params, L);
  const int cw = 16 * w + i;   // the hidden unit this lane's column stands for
  const bool col = i < net.nout;
  const int ntiles = (a.n + 15) / 16;
  // ---- observation rows: thread (row = tid / 16, c = tid % 16 + 16 j) -- 16 threads read a row in 64-byte pieces.  The same 16
  // threads fetch what the loss needs of that row (the action, old neglogp / return, advantage, weight) into a side tile, so that the
  // head phase reads LDS only: with the gathers (idx -> row -> fields, two dependent trips to HBM at shuffled rows) inside the head
  // phase it was 55 % of the kernel (tools/grad_probe.py).  The row index itself is fetched one tile further ahead.
  const int xr = tid >> 4, xc = tid & 15;
  constexpr int NXJ = KT;   // 16 * KT columns cover D
  constexpr int RDS = 20;   // side-tile row: actions 0..15 | 16 old neglogp (policy) / return (value) | 17 advantage | 18 weight
  float xv[NXJ], xact = 0.0f, xsc = 0.0f;
  auto fetch_src = [&](int tile) -> int {
    const int row = tile * 16 + xr;
    return (tile < ntiles && row < a.n) ? (a.idx ? a.idx[row] : row) : -1;
  };
  // (request issues plain unconditional loads -- clamped addresses, no select on the loaded value -- so that nothing waits for them
  // before commit, which applies the masks: a select right behind each load made the compiler predicate the loads pairwise with a wait
  // after every pair, four serial trips to memory inside the head phase)
  bool xrok = false;
  auto request = [&](int tile, int src_) {
    xrok = src_ >= 0;
    const int src = xrok ? src_ : 0, row = xrok ? tile * 16 + xr : 0;
    const float* orow = a.obs + (size_t)src * a.obs_stride;
#pragma unroll
    for (int j = 0; j < NXJ; j++) { const int c = xc + 16 * j; xv[j] = orow[c < D ? c : D - 1]; }
    if (PI) {
      xact = a.actions[(size_t)src * A + (xc < A ? xc : 0)];
      const float* sp = xc == 0 ? a.oldnlp + src : (xc == 1 ? a.adv + row : a.weight + src);
      xsc = sp[0];
    } else {
      xsc = a.returns[src];
    }
  };
  auto commit = [&](float* xb, float* rd) {   // columns D .. 16 KT - 1 are written as zeros (the launch sizes the row stride for the variant: XS >= 16 KT)
#pragma unroll
    for (int j = 0; j < NXJ; j++) xb[xr * XS + xc + 16 * j] = (xrok && xc + 16 * j < D) ? xv[j] : 0.0f;
    if (PI) rd[xr * RDS + xc] = xact;
    if (xc < 3) rd[xr * RDS + 16 + xc] = xrok ? xsc : 0.0f;
  };
  int tile = blockIdx.x;
  request(tile, fetch_src(tile));   // (in flight while the resident operands load)
  // ---- resident B operands (lane (i, kq) holds B[k = kq * (K / 4) + u][column i] of k-step u; the head-delta products keep 4u + kq)
  float bW0[K0], bW1[16], bW1T[16], bW2T[4];
#pragma unroll
  for (int u = 0; u < K0; u++) { const int k = kq * K0 + u; const float v = net.w0[(k < D ? k : D - 1) * H + cw]; bW0[u] = k < D ? v : 0.0f; }
#pragma unroll
  for (int u = 0; u < 16; u++) {
    bW1[u] = net.w1[(kq * 16 + u) * H + cw];
    bW1T[u] = net.w1[cw * H + kq * 16 + u];                                  // (delta W1^T)[row][f] = sum_k delta[row][k] W1[f][k]
  }
  if (w == 0) {
#pragma unroll
    for (int v4 = 0; v4 < 4; v4++) {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const float v = net.w2[(kq * 16 + 4 * v4 + u) * net.nout + (col ? i : 0)]; t[u] = col ? v : 0.0f; }
      w2s[v4 * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; u++) { const int k = 4 * u + kq; const float v = net.w2[cw * net.nout + (k < net.nout ? k : 0)]; bW2T[u] = k < net.nout ? v : 0.0f; }
  const float bias0 = net.b0[cw], bias1 = net.b1[cw];
  const float bias2 = col ? net.b2[i] : 0.0f;
  f32x4 gW0[KT], gW1[4], gW2 = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int ft = 0; ft < KT; ft++) gW0[ft] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int ft = 0; ft < 4; ft++) gW1[ft] = (f32x4){0, 0, 0, 0};
  float gb0 = 0, gb1 = 0, gb2 = 0, glogstd = 0;
  double st_pg = 0, st_vf = 0, st_kl = 0, st_clip = 0, st_cnt = 0;
  const float logstd = (PI && col) ? a.params[L.logstd + i] : 0.0f;
  const float inv_std = 1.0f / expf(logstd);
  const float sum_logstd = row16_sum(logstd);
  commit(xb0, rd0);
  int src_next = fetch_src(tile + gridDim.x);
  __syncthreads();
  GP(0);   // resident operands + first tile
  for (int it = 0; tile < ntiles; tile += gridDim.x, it++) {
    float* xbuf = (it & 1) ? xb1 : xb0;
    const float* rd = (it & 1) ? rd1 : rd0;
    const int r0 = tile * 16;
    // ---- first layer, this wave's 16 units: two accumulation chains over the even / odd k-steps
    f32x4 acc = (f32x4){0, 0, 0, 0}, acc2 = (f32x4){0, 0, 0, 0};
    {
      const float4* xa = (const float4*)(xbuf + i * XS + kq * K0);
#pragma unroll
      for (int v = 0; v < KT; v++) {
        const float4 q = xa[v];
        acc = MFMA(q.x, bW0[4 * v], acc); acc2 = MFMA(q.y, bW0[4 * v + 1], acc2);
        acc = MFMA(q.z, bW0[4 * v + 2], acc); acc2 = MFMA(q.w, bW0[4 * v + 3], acc2);
      }
    }
    float h1v[4], h2v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { h1v[r] = fmaxf(acc[r] + acc2[r] + bias0, 0.0f); h1[(4 * kq + r) * GHS + cw] = h1v[r]; }
    __syncthreads();
    GP(1);
    // ---- second layer
    acc = (f32x4){0, 0, 0, 0}; acc2 = (f32x4){0, 0, 0, 0};
    {
      const float4* ha = (const float4*)(h1 + i * GHS + kq * 16);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float4 q = ha[v];
        acc = MFMA(q.x, bW1[4 * v], acc); acc2 = MFMA(q.y, bW1[4 * v + 1], acc2);
        acc = MFMA(q.z, bW1[4 * v + 2], acc); acc2 = MFMA(q.w, bW1[4 * v + 3], acc2);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { h2v[r] = fmaxf(acc[r] + acc2[r] + bias1, 0.0f); h2[(4 * kq + r) * GHS + cw] = h2v[r]; }
    __syncthreads();
    GP(2);
    request(tile + gridDim.x, src_next);   // the next tile's rows travel during the head and the backward half
    src_next = fetch_src(tile + 2 * gridDim.x);
    // ---- head (every wave) + loss deltas (D layout: rows 4kq+r, column i)
    f32x4 out = (f32x4){0, 0, 0, 0};
    {
      const float4* ha = (const float4*)(h2 + i * GHS + kq * 16);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float4 q = ha[v], b = w2s[v * 64 + lane];
        out = MFMA(q.x, b.x, out); out = MFMA(q.y, b.y, out);
        out = MFMA(q.z, b.z, out); out = MFMA(q.w, b.w, out);
      }
    }
    if (PI) {
      // The per-ROW part of the loss (neglogp -> ratio -> clip -> d loss / d neglogp, ~100 instructions) is evaluated ONCE per row: lane
      // (i, kq) takes row 4 kq + (i & 3), so every quad of lanes holds its group's four rows and hands the result to the other
      // lanes through quad-broadcast DPP moves.  (With every lane walking all four rows this phase was VALU-bound: 38 % of the kernel.)
      float zr[4], ssm = 0.0f;
      const int rm = i & 3;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bool rok = r0 + 4 * kq + r < a.n;
        const float m = out[r] + bias2;
        const float act = (rok && col) ? rd[(4 * kq + r) * RDS + i] : m;
        float z = (act - m) * inv_std;
        if (!(rok && col)) z = 0.0f;
        zr[r] = z;
        const float ss = row16_sum(z * z);
        ssm = rm == r ? ss : ssm;
      }
      float dnlp_m;
      {
        const int row = r0 + 4 * kq + rm;
        const bool rok = row < a.n;
        const float* rdr = rd + (4 * kq + rm) * RDS;
        const float nlp = 0.5f * ssm + 0.5f * LOG2PI_F * (float)A + sum_logstd;
        const float old = rok ? rdr[16] : nlp;
        const float lr = old - nlp;
        float ratio = expf(lr);
        const bool isnan_ = ratio != ratio;
        if (isnan_) ratio = 2.0f;                                         // model.py:96
        const float adv = rok ? rdr[17] : 0.0f, wt = rok ? rdr[18] : 0.0f;
        const float lo = 1.0f - a.cliprange, hi = 1.0f + a.cliprange;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float l1 = -adv * ratio, l2 = -adv * rc;
        const bool first = l1 >= l2;
        const bool in_clip = ratio >= lo && ratio <= hi;
        float dratio = wt * a.inv_count * (first ? -adv : (in_clip ? -adv : 0.0f));
        if (isnan_) dratio = 0.0f;
        dnlp_m = -dratio * ratio;
        if (w == 0 && rok && i < 4) {
          st_pg += (double)(wt * fmaxf(l1, l2));
          st_kl += (double)(nlp - old);
          st_clip += (fabsf(ratio - 1.0f) > a.cliprange) ? 1.0 : 0.0;
          st_cnt += 1.0;
          if (a.log_ratio) a.log_ratio[row] = lr;
        }
      }
      float dn[4];
      dn[0] = quad_bcast<0>(dnlp_m); dn[1] = quad_bcast<1>(dnlp_m); dn[2] = quad_bcast<2>(dnlp_m); dn[3] = quad_bcast<3>(dnlp_m);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bool rok = r0 + 4 * kq + r < a.n;
        const float dhead = col ? dn[r] * (-(zr[r] * inv_std)) : 0.0f;
        if (rok && col) glogstd += dn[r] * (1.0f - zr[r] * zr[r]) - a.ent_coef * a.inv_count;
        mydout[(4 * kq + r) * 18 + i] = dhead;
        gb2 += dhead;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bool rok = r0 + 4 * kq + r < a.n;
        const float v = out[r] + bias2;
        const float R = rok ? rd[(4 * kq + r) * RDS + 16] : v;
        const float dv = v - R;
        const float dhead = (rok && col) ? a.vf_coef * a.inv_count * dv : 0.0f;
        if (w == 0 && rok && i == 0) { st_vf += 0.5 * (double)dv * (double)dv; st_cnt += 1.0; }
        mydout[(4 * kq + r) * 18 + i] = dhead;
        gb2 += dhead;
      }
    }
    wave_sync();
    GP(3);
    // ---- head weight gradient, rows 16w .. 16w+15: gW2 += h2[:, slice]^T dhead
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) { const int row = 4 * s_ + kq; gW2 = MFMA(h2[row * GHS + cw], mydout[row * 18 + i], gW2); }
    // ---- dh2[:, slice] = dhead W2[slice, :]^T, masked by h2 > 0
    acc = (f32x4){0, 0, 0, 0};
    {
      const int nk = (net.nout + 3) >> 2;
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (u < nk) acc = MFMA(mydout[i * 18 + 4 * u + kq], bW2T[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { const float d = h2v[r] > 0.0f ? acc[r] : 0.0f; d2[(4 * kq + r) * GHS + cw] = d; gb1 += d; }
    __syncthreads();
    GP(4);
    // ---- gW1[:, slice] += h1^T d2[:, slice]
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
      const int row = 4 * s_ + kq;
      const float b = d2[row * GHS + cw];
#pragma unroll
      for (int ft = 0; ft < 4; ft++) gW1[ft] = MFMA(h1[row * GHS + ft * 16 + i], b, gW1[ft]);
    }
    // ---- dh1[:, slice] = d2 W1[slice, :]^T, masked by h1 > 0 (only this wave reads its slice of d1 back)
    acc = (f32x4){0, 0, 0, 0}; acc2 = (f32x4){0, 0, 0, 0};
    {
      const float4* da = (const float4*)(d2 + i * GHS + kq * 16);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float4 q = da[v];
        acc = MFMA(q.x, bW1T[4 * v], acc); acc2 = MFMA(q.y, bW1T[4 * v + 1], acc2);
        acc = MFMA(q.z, bW1T[4 * v + 2], acc); acc2 = MFMA(q.w, bW1T[4 * v + 3], acc2);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { const float d = h1v[r] > 0.0f ? acc[r] + acc2[r] : 0.0f; d1[(4 * kq + r) * GHS + cw] = d; gb0 += d; }
    wave_sync();
    GP(5);
    // ---- gW0[:, slice] += x^T d1[:, slice]
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
      const int row = 4 * s_ + kq;
      const float b = d1[row * GHS + cw];
#pragma unroll
      for (int ft = 0; ft < KT; ft++) gW0[ft] = MFMA(xbuf[row * XS + ft * 16 + i], b, gW0[ft]);
    }
    commit((it & 1) ? xb0 : xb1, (it & 1) ? rd0 : rd1);
    __syncthreads();   // the tile's buffers are free again, the next tile's rows are in place
    GP(6);
  }
  // ---- this block's partial gradients (slab) -- D layout: rows 4kq+r of the tile, column i
  float* slab = a.slabs + ((size_t)(PI ? 0 : 1) * a.nwaves + blockIdx.x) * L.P;
  const int o_w0 = PI ? L.pi_w0 : L.vf_w0, o_b0 = PI ? L.pi_b0 : L.vf_b0, o_w1 = PI ? L.pi_w1 : L.vf_w1, o_b1 = PI ? L.pi_b1 : L.vf_b1;
  const int o_w2 = PI ? L.pi_w : L.vf_w, o_b2 = PI ? L.pi_b : L.vf_b;
#pragma unroll
  for (int ft = 0; ft < KT; ft++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int f = ft * 16 + 4 * kq + r;
      if (f < D) slab[o_w0 + f * H + cw] = gW0[ft][r];
    }
#pragma unroll
  for (int ft = 0; ft < 4; ft++)
#pragma unroll
    for (int r = 0; r < 4; r++) slab[o_w1 + (ft * 16 + 4 * kq + r) * H + cw] = gW1[ft][r];
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (col) slab[o_w2 + (16 * w + 4 * kq + r) * net.nout + i] = gW2[r];
  {
    const float s0 = kq_sum(gb0), s1 = kq_sum(gb1);
    if (kq == 0) { slab[o_b0 + cw] = s0; slab[o_b1 + cw] = s1; }
  }
  if (w == 0) {
    const float s2 = kq_sum(gb2), sl = kq_sum(glogstd);
    if (kq == 0 && col) { slab[o_b2 + i] = s2; if (PI) slab[L.logstd + i] = sl; }
    // stats: the lanes that took rows (policy net: i = 0..3 of each group; value net: i == 0) hold the partials
    double t_pg = st_pg, t_vf = st_vf, t_kl = st_kl, t_clip = st_clip, t_cnt = st_cnt;
    if (PI) {   // (policy net: the lanes i = 0..3 of a group hold the partials of their rows)
      for (int o = 1; o < 4; o <<= 1) {
        t_pg += __shfl_xor(t_pg, o, WAVE); t_kl += __shfl_xor(t_kl, o, WAVE); t_clip += __shfl_xor(t_clip, o, WAVE); t_cnt += __shfl_xor(t_cnt, o, WAVE);
      }
    }
    for (int o = 16; o < 64; o <<= 1) {
      t_pg += __shfl_xor(t_pg, o, WAVE); t_vf += __shfl_xor(t_vf, o, WAVE); t_kl += __shfl_xor(t_kl, o, WAVE);
      t_clip += __shfl_xor(t_clip, o, WAVE); t_cnt += __shfl_xor(t_cnt, o, WAVE);
    }
    if (lane == 0) {
      double* ws = a.wstats + ((size_t)(PI ? 0 : 1) * a.nwaves + blockIdx.x) * 8;
      ws[0] = t_pg; ws[1] = t_vf; ws[2] = 0; ws[3] = t_kl; ws[4] = t_clip; ws[5] = 0; ws[6] = PI ? t_cnt : 0.0; ws[7] = 0;
    }
  }
  GP_DRAIN();
  GP(7);   // slab
  GP_FLUSH();
}

#define GRAD_LDS_FLOATS(XS) (2 * 16 * (XS) + 4 * 16 * GRAD_HS + 4 * 16 * 18 + 2 * 16 * 20 + 4 * 64 * 4)
// Ant (KT = 8): 255 registers, two workgroups per CU.  The wider first layers of the Bug / Spider nets (KT = 11 / 14: 44 / 56 resident
// B operands and as many accumulators) get the whole register file of a SIMD instead of spilling: one workgroup per CU.
template <int KT>
__global__ void __launch_bounds__(256, (KT <= 8 ? 2 : 1)) ppo_grad_kernel(GradArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (blockIdx.y == 0) grad_net_coop<KT, true>(a, smem_f, lane, wid);
  else grad_net_coop<KT, false>(a, smem_f, lane, wid);
}

// grads[p] = sum over workgroups of the slab of the net that owns p; stats += per-workgroup stats.  ONE launch, two levels, fixed
// association (deterministic, no float atomics): workgroup (chunk, g) sums the RED_GROUP consecutive slabs of group g for its 256
// parameters into partial[g][p]; the LAST workgroup of a chunk to finish (a counter per chunk; the partials travel through sc1
// stores / loads, so the other XCDs' L2s do not matter) adds the G partials in order and writes grads.  The last workgroup of chunk 0
// also sums the statistics records.  The counters live in the caller's workspace, start at zero and are left at zero.
#define RED_GROUP 16
__global__ void __launch_bounds__(256) ppo_grad_reduce_kernel(const float* slabs, int nslabs, ParamLayout L, float* partial, int G,
                                                              unsigned int* arrived, const double* wstats, float* grads, double* stats) {
  if (blockIdx.x == gridDim.x - 1) {   // one extra workgroup (column gridDim.x - 1, row 0; the rest of that column leaves): the
    if (blockIdx.y != 0) return;       // statistics, 8 x 2*nslabs records written by the gradient kernel, i.e. before this launch --
    __shared__ double chunk[32][8];    // 32 chunks per statistic in parallel, then the chunks in order
    const int k = threadIdx.x & 7, c = threadIdx.x >> 3, n = 2 * nslabs, per = (n + 31) / 32;
    double acc = 0;
    for (int w = c * per; w < (c + 1) * per && w < n; w++) acc += wstats[(size_t)w * 8 + k];
    chunk[c][k] = acc;
    __syncthreads();
    if (threadIdx.x < 8) {
      double t = 0;
      for (int q = 0; q < 32; q++) t += chunk[q][threadIdx.x];
      stats[threadIdx.x] += t;
    }
    return;
  }
  const int p = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  __shared__ int last;
  if (p < L.P) {
    const bool is_vf = (p >= L.vf_w0 && p < L.pi_w) || p >= L.vf_w;
    const float* s = slabs + (size_t)(is_vf ? 1 : 0) * nslabs * L.P + p;
    const int w0 = g * RED_GROUP, w1 = w0 + RED_GROUP < nslabs ? w0 + RED_GROUP : nslabs;
    float v[RED_GROUP];
#pragma unroll
    for (int k = 0; k < RED_GROUP; k++) v[k] = w0 + k < w1 ? s[(size_t)(w0 + k) * L.P] : 0.0f;   // all loads in flight
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < RED_GROUP; k++) acc += v[k];
    __hip_atomic_store(partial + (size_t)g * L.P + p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                       // every thread's partial has left for memory
  if (threadIdx.x == 0) {
    const unsigned int n = __hip_atomic_fetch_add(arrived + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (n == (unsigned int)G - 1);
    if (last) __hip_atomic_store(arrived + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  }
  __syncthreads();
  if (!last) return;
  if (p < L.P) {   // the partials in their fixed order, 16 loads in flight at a time (one dependent trip per load was most of this kernel)
    float acc = 0.0f;
    for (int q0 = 0; q0 < G; q0 += 16) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; k++)
        v[k] = q0 + k < G ? __hip_atomic_load(partial + (size_t)(q0 + k) * L.P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
#pragma unroll
      for (int k = 0; k < 16; k++) acc += v[k];
    }
    grads[p] = acc;
  }
}

static int grad_nwaves(void) { return 256 * 4; }   // slabs per net the workspace is sized for
extern "C" size_t ppo_grad_workspace_bytes(int ob_dim, int ac_dim) {
  ParamLayout L = make_layout(ob_dim, ac_dim);
  return (size_t)2 * grad_nwaves() * L.P * sizeof(float) + (size_t)2 * grad_nwaves() * 8 * sizeof(double) +
         (size_t)(grad_nwaves() / RED_GROUP) * L.P * sizeof(float) +      // slabs | per-workgroup stats | partial sums
         (size_t)((L.P + 255) / 256) * sizeof(unsigned int);              // | arrival counter per 256-parameter chunk (zero between calls)
}

extern "C" int ppo_grad(const float* params, const float* obs, int obs_stride, int ob_dim, int ac_dim, const float* actions,
                        const float* adv_mb, const float* returns, const float* old_neglogp, const float* is_weight, const int32_t* idx,
                        int n, double inv_count, float cliprange, float ent_coef, float vf_coef, float* grads, double* stats,
                        float* log_ratio_out, void* workspace, void* stream) {
  if (!params || !obs || !actions || !adv_mb || !returns || !old_neglogp || !is_weight || !grads || !stats || !workspace || n <= 0)
    FAIL(-1, "bad arguments");
  if (ac_dim < 1 || ac_dim > MAXA) FAIL(-2, "ac_dim %d not in [1,%d]", ac_dim, MAXA);
  int KT = (ob_dim + 15) / 16;
  if (KT > 14) FAIL(-3, "ob_dim %d too large (max 224)", ob_dim);
  GradArgs a;
  a.params = params; a.obs = obs; a.actions = actions; a.adv = adv_mb; a.returns = returns; a.oldnlp = old_neglogp; a.weight = is_weight;
  // the staged tile has the variant's full width (16 KT columns, zeros beyond ob_dim): no guards inside the tile loop
  a.idx = idx; a.n = n; a.obs_stride = obs_stride; a.XS = 16 * (KT <= 8 ? 8 : (KT <= 11 ? 11 : 14)) + 4;   /* 132 / 180 / 228 floats: rows 16-byte aligned, 128-bit A-operand reads conflict free */ a.inv_count = (float)inv_count; a.cliprange = cliprange;
  a.ent_coef = ent_coef; a.vf_coef = vf_coef; a.log_ratio = log_ratio_out; a.L = make_layout(ob_dim, ac_dim);
  int ntiles = (n + 15) / 16;
  int nblocks = ntiles;      // workgroups per net; a workgroup (4 waves) takes the tiles blockIdx.x, blockIdx.x + nblocks, ...
  {  // default 256 per net = two resident workgroups per CU over both nets: every workgroup writes one slab (P/2 floats), so more of
     // them shorten the tile loops but lengthen the slab reduction
    const char* nbc = getenv("PPO_GRAD_BLOCKS");
    int cap = nbc ? atoi(nbc) : 256;
    if (cap < 1) cap = 1;
    if (cap > grad_nwaves()) cap = grad_nwaves();   // the workspace holds grad_nwaves() slabs per net
    if (nblocks > cap) nblocks = cap;
  }
  a.nwaves = nblocks;        // slabs (and statistics records) per net
  a.slabs = (float*)workspace;
  a.wstats = (double*)((char*)workspace + (size_t)2 * grad_nwaves() * a.L.P * sizeof(float));
  size_t lds = (size_t)GRAD_LDS_FLOATS(a.XS) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  // the slabs are only partially written by each net (its own ranges); the reduce reads only those ranges
#define LAUNCH(KTV)                                                                                                     \
  do {                                                                                                                  \
    static thread_local size_t lds_set_##KTV = 0;                                                                       \
    if (lds > 64 * 1024 && lds > lds_set_##KTV) {                                                                       \
      HIPCHK(hipFuncSetAttribute((const void*)ppo_grad_kernel<KTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_set_##KTV = lds;                                                                                              \
    }                                                                                                                   \
    hipLaunchKernelGGL(ppo_grad_kernel<KTV>, dim3(nblocks, 2), dim3(256), lds, s, a);                                  \
  } while (0)
  if (KT <= 8) LAUNCH(8);
  else if (KT <= 11) LAUNCH(11);
  else LAUNCH(14);   // (the tile's 16 KT staged columns must fit its LDS row: x_stride(D) >= 16 KT for every D of a variant)
#undef LAUNCH
  HIPCHK(hipGetLastError());
  {
    float* partial = (float*)((char*)workspace + (size_t)2 * grad_nwaves() * a.L.P * sizeof(float) + (size_t)2 * grad_nwaves() * 8 * sizeof(double));
    const int G = (a.nwaves + RED_GROUP - 1) / RED_GROUP;
    unsigned int* arrived = (unsigned int*)(partial + (size_t)(grad_nwaves() / RED_GROUP) * a.L.P);
    hipLaunchKernelGGL(ppo_grad_reduce_kernel, dim3((a.L.P + 255) / 256 + 1, G), dim3(256), 0, s, a.slabs, a.nwaves, a.L, partial, G, arrived,
                       a.wstats, grads, stats);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

#ifdef PPO_GRAD_PROBE
extern "C" int ppo_debug_gprobe(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gprobe), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------------------------------------------------
// clip_by_global_norm + TF1 Adam.  Every block computes the global norm itself (same loads, same order -> the same
// value in every block; P ~ 25k floats sit in L2) and then updates its own 1024 parameters.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ppo_clip_adam_kernel(float* params, const float* grads, float* m, float* v, int P, float lr_t,
                                                             float beta1, float beta2, float eps, float max_norm, double* stats) {
  __shared__ double red[16];
  double s = 0;
  for (int k0 = threadIdx.x; k0 < P; k0 += 8 * 1024) {
    float g[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { int k = k0 + 1024 * u; g[u] = k < P ? grads[k] : 0.0f; }
#pragma unroll
    for (int u = 0; u < 8; u++) s += (double)g[u] * (double)g[u];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  double tot = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) tot += red[i];
  float norm = (float)sqrt(tot);
  float scale = 1.0f;
  if (max_norm > 0.0f) scale = max_norm / fmaxf(norm, max_norm);          // tf.clip_by_global_norm
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats) stats[7] = (double)norm;
  const int k = blockIdx.x * 1024 + threadIdx.x;
  if (k < P) {
    float g = grads[k] * scale;
    float mk = beta1 * m[k] + (1.0f - beta1) * g;
    float vk = beta2 * v[k] + (1.0f - beta2) * g * g;
    m[k] = mk; v[k] = vk;
    params[k] -= lr_t * mk / (sqrtf(vk) + eps);                            // TF1: epsilon outside the bias correction
  }
}
// The five loss statistics a minibatch step reports (model.py:138 loss_names) from the accumulated sums + the entropy of the
// diagonal Gaussian the loss was evaluated with (distributions.py:246-247: sum(logstd + 0.5 log(2 pi e)), float64 like the
// host code it replaces): one launch instead of ten elementwise ones inside the step graph.
__global__ void ppo_loss_stats_kernel(const double* stats, const float* logstd, int A, double* out5) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ent = 0.0;
  for (int i = 0; i < A; i++) ent += (double)logstd[i] + 0.5 * 2.8378770664093453;   // log(2 pi e)
  const double cnt = stats[6];
  out5[0] = stats[0] / cnt; out5[1] = stats[1] / cnt; out5[2] = ent; out5[3] = stats[3] / cnt; out5[4] = stats[4] / cnt;
}
extern "C" int ppo_loss_stats(const double* stats, const float* logstd, int ac_dim, double* out5, void* stream) {
  if (!stats || !logstd || !out5 || ac_dim < 1) FAIL(-1, "bad arguments");
  hipLaunchKernelGGL(ppo_loss_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, logstd, ac_dim, out5);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int ppo_clip_adam(float* params, const float* grads, float* m, float* v, int P, int t, double lr, double beta1, double beta2,
                             double eps, double max_grad_norm, double* stats, void* stream) {
  if (!params || !grads || !m || !v || P <= 0 || t < 1) FAIL(-1, "bad arguments");
  double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
  hipLaunchKernelGGL(ppo_clip_adam_kernel, dim3((P + 1023) / 1024), dim3(1024), 0, (hipStream_t)stream, params, grads, m, v, P, (float)lr_t, (float)beta1,
                     (float)beta2, (float)eps, (float)max_grad_norm, stats);
  HIPCHK(hipGetLastError());
  return 0;
}
