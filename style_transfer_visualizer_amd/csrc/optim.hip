// Device-resident optimizer updates for the image being optimised.
//
// L-BFGS: restates torch.optim.LBFGS.step (torch 2.10 optim/lbfgs.py, the
// implementation the reference constructs at core_model.py:344-349) for
// max_iter = 1 / no line search, which is the reference default
// (config_defaults.py:12-13).  torch's version reads 2m+4 scalars back to the
// host per step; here every scalar (|g|max, y.s, y.y, alpha_i, beta_i, g.d) and
// every branch that depends on one (early return when |g|max <= tol_grad, the
// y.s > 1e-10 history push, the g.d > -tol_change break) lives in a small
// device state block, so a step enqueues 2*m_max+4 kernels and never syncs.
// The two-loop recursion keeps torch's operation order: one fused
// "apply previous axpy, then dot with the next history vector" pass per
// iteration; dot partials are reduced in a fixed order (deterministic).
//
// Adam: torch _single_tensor_adam (no amsgrad / weight decay) as one pass.
#include "stv_common.h"

namespace {

constexpr int LB = 512;        // blocks per vector pass (= partial sums per dot)
constexpr int MAX_HIST = 128;

struct LbfgsState {
  int n_iter;
  int hist_len;
  int head;
  int skip;
  int no_update;
  int steps_seen;
  float t;
  float H_diag;
  float gtd;
  float gmax;
  float ys;
  float yy;
  float ro[MAX_HIST + 1];  // by ring slot
  float al[MAX_HIST + 1];  // by logical index (0 = oldest)
};

struct LbfgsWs {
  float* d;       // direction; doubles as q / r during the two-loop
  float* prev_g;
  float* S;       // [(hist+1)][n] ring, s_i = t * d
  float* Y;       // [(hist+1)][n] ring, y_i = g - g_prev
  float* part;    // [8][LB]
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline LbfgsWs carve(void* workspace, size_t n, int hist) {
  const size_t nn = align_up(n, 64);
  float* p = static_cast<float*>(workspace);
  LbfgsWs w;
  w.d = p; p += nn;
  w.prev_g = p; p += nn;
  w.S = p; p += nn * (hist + 1);
  w.Y = p; p += nn * (hist + 1);
  w.part = p;
  return w;
}

__device__ __forceinline__ float reduce_partials(const float* __restrict__ part, float* red4) {
  // every block sums the same LB partials in the same order -> identical value
  float s = part[threadIdx.x] + part[threadIdx.x + 256];
  return block_sum_256(s, red4);
}

// ---- pass 1: statistics of g and the candidate (y, s) pair -------------------
__global__ __launch_bounds__(256) void lbfgs_pre_kernel(const float* __restrict__ g, LbfgsState* st,
                                                        LbfgsWs w, size_t n, size_t nn, int hist) {
  __shared__ float red[4];
  const float t = st->t;
  const int slot = (st->head + st->hist_len) % (hist + 1);
  float* __restrict__ yc = w.Y + (size_t)slot * nn;
  float* __restrict__ sc = w.S + (size_t)slot * nn;
  float amax = 0.0f, l1 = 0.0f, ys = 0.0f, yy = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)LB * 256) {
    const float gv = g[i];
    const float y = gv - w.prev_g[i];
    const float s = w.d[i] * t;
    yc[i] = y;
    sc[i] = s;
    amax = fmaxf(amax, fabsf(gv));
    l1 += fabsf(gv);
    ys = fmaf(y, s, ys);
    yy = fmaf(y, y, yy);
  }
  amax = block_max_256(amax, red);
  l1 = block_sum_256(l1, red);
  ys = block_sum_256(ys, red);
  yy = block_sum_256(yy, red);
  if (threadIdx.x == 0) {
    w.part[0 * LB + blockIdx.x] = amax;
    w.part[1 * LB + blockIdx.x] = l1;
    w.part[2 * LB + blockIdx.x] = ys;
    w.part[3 * LB + blockIdx.x] = yy;
  }
}

// ---- the scalar control flow of LBFGS.step, on device ------------------------
__global__ __launch_bounds__(256) void lbfgs_decide_kernel(LbfgsState* st, LbfgsWs w, int hist, float lr,
                                                           float tol_grad) {
  __shared__ float red[4];
  const float amax = block_max_256(fmaxf(w.part[threadIdx.x], w.part[threadIdx.x + 256]), red);
  const float l1 = reduce_partials(w.part + 1 * LB, red);
  const float ys = reduce_partials(w.part + 2 * LB, red);
  const float yy = reduce_partials(w.part + 3 * LB, red);
  if (threadIdx.x != 0) return;
  st->steps_seen += 1;
  st->gmax = amax;
  st->no_update = 0;
  const int skip = (amax <= tol_grad) ? 1 : 0;  // opt_cond: return before touching any state
  st->skip = skip;
  if (skip) return;
  st->n_iter += 1;
  if (st->n_iter == 1) {
    st->hist_len = 0;
    st->head = 0;
    st->H_diag = 1.0f;
  } else {
    st->ys = ys;
    st->yy = yy;
    if (ys > 1e-10f) {
      const int slot = (st->head + st->hist_len) % (hist + 1);
      if (st->hist_len == hist) st->head = (st->head + 1) % (hist + 1);
      else st->hist_len += 1;
      st->ro[slot] = 1.0f / ys;
      st->H_diag = ys / yy;
    }
  }
  if (st->n_iter == 1) {
    const float inv = 1.0f / l1;
    st->t = ((inv < 1.0f) ? inv : 1.0f) * lr;
  } else {
    st->t = lr;
  }
}

// ---- two-loop, first loop: step j handles logical index i = m-1-j -------------
__global__ __launch_bounds__(256) void lbfgs_loop1_kernel(const float* __restrict__ g, LbfgsState* st,
                                                          LbfgsWs w, size_t n, size_t nn, int hist, int j) {
  __shared__ float red[4];
  const int m = st->hist_len;
  if (st->skip || j >= m) return;
  const int i = m - 1 - j;
  const int head = st->head;
  float coef = 0.0f;
  const float* __restrict__ yprev = nullptr;
  if (j > 0) {
    const int sp = (head + i + 1) % (hist + 1);
    const float al = reduce_partials(w.part + (4 + ((j - 1) & 1)) * LB, red) * st->ro[sp];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->al[i + 1] = al;
    coef = -al;
    yprev = w.Y + (size_t)sp * nn;
  }
  const float* __restrict__ si = w.S + (size_t)((head + i) % (hist + 1)) * nn;
  float dot = 0.0f;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256) {
    float q = (j == 0) ? -g[k] : fmaf(coef, yprev[k], w.d[k]);
    w.d[k] = q;
    dot = fmaf(si[k], q, dot);
  }
  dot = block_sum_256(dot, red);
  if (threadIdx.x == 0) w.part[(4 + (j & 1)) * LB + blockIdx.x] = dot;
}

// ---- second loop: step j handles logical index i = j --------------------------
__global__ __launch_bounds__(256) void lbfgs_loop2_kernel(LbfgsState* st, LbfgsWs w, size_t n, size_t nn,
                                                          int hist, int j) {
  __shared__ float red[4];
  const int m = st->hist_len;
  if (st->skip || j >= m) return;
  const int head = st->head;
  const float* __restrict__ yi = w.Y + (size_t)((head + j) % (hist + 1)) * nn;
  float dot = 0.0f;
  if (j == 0) {
    // finish loop 1: q -= al[0]*y_0 ; r = q * H_diag
    const int s0 = head % (hist + 1);
    const float al0 = reduce_partials(w.part + (4 + ((m - 1) & 1)) * LB, red) * st->ro[s0];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->al[0] = al0;
    const float H = st->H_diag;
    const float* __restrict__ y0 = w.Y + (size_t)s0 * nn;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256) {
      const float r = fmaf(-al0, y0[k], w.d[k]) * H;
      w.d[k] = r;
      dot = fmaf(yi[k], r, dot);
    }
  } else {
    const int sp = (head + j - 1) % (hist + 1);
    const float be = reduce_partials(w.part + (6 + ((j - 1) & 1)) * LB, red) * st->ro[sp];
    // al[j-1] was written by an earlier launch (block 0), visible across the kernel boundary
    const float coef = st->al[j - 1] - be;
    const float* __restrict__ sprev = w.S + (size_t)sp * nn;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256) {
      const float r = fmaf(coef, sprev[k], w.d[k]);
      w.d[k] = r;
      dot = fmaf(yi[k], r, dot);
    }
  }
  dot = block_sum_256(dot, red);
  if (threadIdx.x == 0) w.part[(6 + (j & 1)) * LB + blockIdx.x] = dot;
}

// ---- finish the direction, g.d partials, prev_g <- g --------------------------
__global__ __launch_bounds__(256) void lbfgs_dir_kernel(const float* __restrict__ g, LbfgsState* st,
                                                        LbfgsWs w, size_t n, size_t nn, int hist) {
  __shared__ float red[4];
  if (st->skip) return;
  const int m = st->hist_len;
  float gtd = 0.0f;
  if (m == 0) {
    const float H = st->H_diag;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256) {
      const float gv = g[k];
      const float d = -gv * H;
      w.d[k] = d;
      w.prev_g[k] = gv;
      gtd = fmaf(gv, d, gtd);
    }
  } else {
    const int sp = (st->head + m - 1) % (hist + 1);
    const float be = reduce_partials(w.part + (6 + ((m - 1) & 1)) * LB, red) * st->ro[sp];
    const float coef = st->al[m - 1] - be;
    const float* __restrict__ sprev = w.S + (size_t)sp * nn;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256) {
      const float gv = g[k];
      const float d = fmaf(coef, sprev[k], w.d[k]);
      w.d[k] = d;
      w.prev_g[k] = gv;
      gtd = fmaf(gv, d, gtd);
    }
  }
  gtd = block_sum_256(gtd, red);
  if (threadIdx.x == 0) w.part[0 * LB + blockIdx.x] = gtd;
}

__global__ __launch_bounds__(256) void lbfgs_update_kernel(float* __restrict__ x, LbfgsState* st, LbfgsWs w,
                                                           size_t n, float tol_change) {
  __shared__ float red[4];
  if (st->skip) return;
  const float gtd = reduce_partials(w.part, red);
  const bool no_update = gtd > -tol_change;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->gtd = gtd;
    st->no_update = no_update ? 1 : 0;
  }
  if (no_update) return;
  const float t = st->t;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)LB * 256)
    x[k] = fmaf(t, w.d[k], x[k]);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float lr, float w1, float b2, float w2, float eps,
                                                   float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gv = g[i];
    const float mo = m[i];
    const float mn = mo + w1 * (gv - mo);                   // lerp_(grad, 1-beta1)
    const float vn = v[i] * b2 + w2 * gv * gv;              // mul_(beta2).addcmul_(g, g, 1-beta2)
    m[i] = mn;
    v[i] = vn;
    const float denom = sqrtf(vn) / bc2_sqrt + eps;
    x[i] = x[i] - step_size * (mn / denom);                 // addcdiv_(m, denom, -step_size)
  }
}

}  // namespace

extern "C" size_t stv_lbfgs_state_bytes(int history) {
  (void)history;
  return sizeof(LbfgsState);
}

extern "C" size_t stv_lbfgs_workspace_bytes(size_t n, int history) {
  const size_t nn = align_up(n, 64);
  return (nn * (2 + 2 * (size_t)(history + 1)) + 8 * (size_t)LB) * sizeof(float);
}

extern "C" int stv_lbfgs_step(float* x, const float* grad, void* state, void* workspace, size_t n,
                              int history, int m_max, float lr, float tol_grad, float tol_change,
                              void* stream) {
  if (!x || !grad || !state || !workspace || n == 0) return STV_ERR_ARG;
  if (history < 1 || history > MAX_HIST) return STV_ERR_ARG;
  if (m_max < 0) m_max = 0;
  if (m_max > history) m_max = history;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LbfgsState* s = static_cast<LbfgsState*>(state);
  const LbfgsWs w = carve(workspace, n, history);
  const size_t nn = align_up(n, 64);
  hipLaunchKernelGGL(lbfgs_pre_kernel, dim3(LB), dim3(256), 0, st, grad, s, w, n, nn, history);
  hipLaunchKernelGGL(lbfgs_decide_kernel, dim3(1), dim3(256), 0, st, s, w, history, lr, tol_grad);
  for (int j = 0; j < m_max; ++j)
    hipLaunchKernelGGL(lbfgs_loop1_kernel, dim3(LB), dim3(256), 0, st, grad, s, w, n, nn, history, j);
  for (int j = 0; j < m_max; ++j)
    hipLaunchKernelGGL(lbfgs_loop2_kernel, dim3(LB), dim3(256), 0, st, s, w, n, nn, history, j);
  hipLaunchKernelGGL(lbfgs_dir_kernel, dim3(LB), dim3(256), 0, st, grad, s, w, n, nn, history);
  hipLaunchKernelGGL(lbfgs_update_kernel, dim3(LB), dim3(256), 0, st, x, s, w, n, tol_change);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

extern "C" int stv_adam_step(float* x, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                             float lr, float one_minus_beta1, float beta2, float one_minus_beta2,
                             float eps, float bias_c1, float bias_c2_sqrt, void* stream) {
  if (!x || !grad || !exp_avg || !exp_avg_sq || n == 0) return STV_ERR_ARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     grad, exp_avg, exp_avg_sq, n, lr, one_minus_beta1, beta2, one_minus_beta2, eps, bias_c1,
                     bias_c2_sqrt);
  STV_CHECK_LAUNCH();
  return STV_OK;
}
