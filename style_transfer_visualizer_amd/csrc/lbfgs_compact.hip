// L-BFGS step in "dot-product space": the same update as torch.optim.LBFGS.step
// (max_iter = 1, no line search; see optim.hip for the operation-ordered form)
// but with the history read exactly twice per step instead of through 2m
// dependent vector passes.
//
// The two-loop recursion only ever needs inner products between the history
// vectors {s_i}, {y_i} and the gradient g.  With q = -g - sum_j alpha_j y_j:
//     alpha_i = rho_i * (s_i . q) = rho_i * ( -(s_i.g) - sum_{j>i} alpha_j (s_i.y_j) )
//     r = H q + sum_j (alpha_j - beta_j) s_j
//     beta_i = rho_i * (y_i . r)  -> needs (y_i.g), (y_i.y_j), (y_i.s_j)
// so we keep the S x S tables SY[i][j] = s_i.y_j and YY[i][j] = y_i.y_j across
// steps (a pushed pair installs one row/column), and per step
//   pass A : one sweep over the 2m history vectors + g computes every new dot
//            product (5 per pair + 7 scalars) as per-wave partial sums;
//   reduce : fixed-order summation of the partials (deterministic);
//   solve  : one workgroup runs torch's scalar control flow and the recursion on
//            coefficients (in double) and emits d = cg*g + sum cs_j s_j + cy_j y_j;
//   pass B : second sweep forms d, applies x += t*d and prev_g = g.
// HBM traffic per step: (4m + 8) vectors instead of ~(8m) with 2m+4 launches.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "stv_common.h"

#ifndef STV_LBFGS_PIPE_DEFAULT
#define STV_LBFGS_PIPE_DEFAULT 0      // sweep A with the next pair's loads in flight (STV_LBFGS_PIPE=1); set from the measurement
#endif

namespace {

constexpr int MAX_HIST = 128;
constexpr int MAX_S = MAX_HIST + 1;
constexpr int NSCAL = 8;   // gmax, |g|_1, g.g, g.s_c, g.y_c, s_c.y_c, y_c.y_c, (spare)

struct CState {
  int n_iter, hist_len, head, skip, no_update, pushed, steps_seen, pad0;
  float t, H_diag, gtd, gmax, ys, yy, cg, pad1;
  float ro[MAX_S];   // by ring slot
  float cs[MAX_S];   // direction coefficients by ring slot
  float cy[MAX_S];
};

struct CWs {
  float* d;
  float* prev_g;
  float* S;
  float* Y;
  double* SY;     // [S][S] by ring slot.  Double: an inner product of two fp32 vectors can exceed the fp32 range
  double* YY;     // [S][S]  (y.y of a step that overshot by 20 orders of magnitude is ~1e53 - torch's vector
                  // recursion never forms that number, the table form must be able to hold it)
  double* dots;   // [5*MAX_HIST + NSCAL]
  double* partd;  // [NSCAL][nparts]  per-wave partial sums of the step's scalars
  double* part;   // [5*hist][nparts] per-wave partial sums of the history products
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline int tile_floats(size_t n) {   // elements per workgroup tile: >= 256 workgroups, as fat as possible
  static const int forced = getenv("STV_LBFGS_TILE") ? atoi(getenv("STV_LBFGS_TILE")) : 0;   // tuning aid
  if (forced == 1024 || forced == 2048 || forced == 4096) return forced;
  // sweep A: the five wave reductions per history pair amortise over the tile, so it wants fat tiles and gets its
  // workgroup count from dealing a tile's pairs to several workgroups (step_geom: 512^2 = 192 tiles x 4 groups,
  // measured 124.6 us against 137.9 on 384 tiles of 2048)
  if (n >= (size_t)4096 * 192) return 4096;
  if (n >= (size_t)2048 * 256) return 2048;
  return 1024;
}

inline CWs carve(void* workspace, size_t n, int hist, int nparts) {
  const size_t nn = align_up(n, 4096);
  const int S = hist + 1;
  float* p = static_cast<float*>(workspace);
  CWs w;
  w.d = p; p += nn;
  w.prev_g = p; p += nn;
  w.S = p; p += nn * S;
  w.Y = p; p += nn * S;
  double* q = reinterpret_cast<double*>(p);         // (nn is a multiple of 4096 floats: 8-byte aligned)
  w.SY = q; q += align_up((size_t)S * S, 64);
  w.YY = q; q += align_up((size_t)S * S, 64);
  w.dots = q; q += align_up(5 * (size_t)MAX_HIST + NSCAL, 64);
  w.partd = q; q += align_up((size_t)NSCAL * nparts, 64);
  w.part = q;
  return w;
}

__device__ __forceinline__ f32x4 ld4_guard(const float* __restrict__ p, size_t idx, size_t n) {
  if (idx + 3 < n) return *reinterpret_cast<const f32x4*>(p + idx);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (idx < n) v[0] = p[idx];
  if (idx + 1 < n) v[1] = p[idx + 1];
  if (idx + 2 < n) v[2] = p[idx + 2];
  return v;
}
// History loads: NT = non-temporal (`global_load ... nt`).  The 2m history vectors are streamed once per sweep and not
// touched again before the next closure has turned every cache over; loading them non-temporally in BOTH sweeps measured
// -2.0 % step time at 512^2 (0.907 -> 0.888 ms, two alternating runs each on one box), within noise at 1024^2 (2.643 ->
// 2.627); either sweep alone: nothing (A) / -1 % (B).  Same values, same order: bit-identical results.
// STV_LBFGS_NT: bit 0 = sweep A, bit 1 = sweep B (default 3).
template <bool NT>
__device__ __forceinline__ f32x4 ldh4(const float* __restrict__ p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  else return *reinterpret_cast<const f32x4*>(p);
}
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
  return fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], a[3] * b[3])));
}

// ---- pass A ---------------------------------------------------------------------------------
template <int U, bool NT = false, bool PIPE = false>
__global__ __launch_bounds__(256) void pass_a_kernel(const float* __restrict__ g, const CState* st, CWs w,
                                                     size_t n, size_t nn, int hist, int nparts, int ntiles, int pgroups) {
  // blockIdx.x = pair group * ntiles + tile.  A small image has too few tiles to keep enough loads in flight
  // (256^2: 192 workgroups walking 100 pairs one after the other ran at 2.3 TB/s), so the PAIRS of a tile are
  // dealt to `pgroups` workgroups; every (pair, tile, wave) partial still has exactly one writer, and the
  // scalars / the new pair's own vectors belong to group 0.
  const int S = hist + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_idx = (int)blockIdx.x % ntiles, pgrp = (int)blockIdx.x / ntiles;
  const size_t base = (size_t)tile_idx * (256 * 4 * U);
  const float t = st->t;
  const int m = st->hist_len, head = st->head;
  const int cslot = (head + m) % S;
  float* __restrict__ yc = w.Y + (size_t)cslot * nn;
  float* __restrict__ sc = w.S + (size_t)cslot * nn;
  f32x4 gv[U], sv[U], yv[U];
  // The step's own scalars are summed in double from the start: g.g and y.y overflow fp32 when a step
  // overshoots (the reference's L-BFGS has no line search: 5.6e8 -> 8.3e29 in tests/golden/mini_clamp_lbfgs),
  // torch's vector recursion survives that (its y.y = inf only makes H_diag 0) and so must this one -
  // an inf here would turn into 0 * inf = NaN in the coefficient recursion.
  float gmax = 0.f;
  double gl1 = 0.0, gg = 0.0, gs = 0.0, gy = 0.0, sy = 0.0, yy = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t idx = base + (size_t)(u * 256 + tid) * 4;
    gv[u] = ld4_guard(g, idx, n);
    const f32x4 pg = *reinterpret_cast<const f32x4*>(w.prev_g + idx);   // nn-padded, zero tail
    const f32x4 dv = *reinterpret_cast<const f32x4*>(w.d + idx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      yv[u][e] = gv[u][e] - pg[e];
      sv[u][e] = dv[e] * t;
      gmax = fmaxf(gmax, fabsf(gv[u][e]));
    }
    if (pgrp == 0) {
      *reinterpret_cast<f32x4*>(yc + idx) = yv[u];
      *reinterpret_cast<f32x4*>(sc + idx) = sv[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double gd = (double)gv[u][e], sd = (double)sv[u][e], yd = (double)yv[u][e];
        gl1 += fabs(gd);
        gg = fma(gd, gd, gg);
        gs = fma(gd, sd, gs);
        gy = fma(gd, yd, gy);
        sy = fma(sd, yd, sy);
        yy = fma(yd, yd, yy);
      }
    }
  }
  const int p = tile_idx * 4 + wave;
  const int per = (m + pgroups - 1) / pgroups;
  const int j_end = (pgrp + 1) * per < m ? (pgrp + 1) * per : m;
  // Double accumulators: a product of two fp32 history / gradient vectors can leave the fp32 range (a step
  // that overshot leaves y ~ 1e25 in the history: y_j . y_c ~ 1e50), and torch's vector recursion - which
  // never forms these inner products - stays finite there.  A product of two floats is exact in double.
  // (the fourth product of a pair, y_j . s_c, fills the table entry s_c . y_j of a NEWER s with an OLDER y - an entry the
  //  recursion never reads (solve_kernel: only s_i . y_j with i older than j) - so it is not computed: its slot in the
  //  5-per-pair layout stays, as a zero)
  // PIPE: the next pair's loads are in flight while this pair is multiplied (two register sets; affordable since the
  // unused product went: 160 registers = the three waves per SIMD the grid gives anyway).
  auto load_pair = [&](int jj, f32x4 (&s4)[U], f32x4 (&y4)[U]) {
    const int slot = (head + jj) % S;
    const float* __restrict__ sj = w.S + (size_t)slot * nn;
    const float* __restrict__ yj = w.Y + (size_t)slot * nn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t idx = base + (size_t)(u * 256 + tid) * 4;
      s4[u] = ldh4<NT>(sj + idx);
      y4[u] = ldh4<NT>(yj + idx);
    }
  };
  auto dot_pair = [&](int jj, const f32x4 (&s4)[U], const f32x4 (&y4)[U]) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a4 = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double sd = (double)s4[u][e], yd = (double)y4[u][e];
        const double gd = (double)gv[u][e], cd = (double)yv[u][e];
        a0 = fma(sd, gd, a0);   // s_j . g
        a1 = fma(yd, gd, a1);   // y_j . g
        a2 = fma(sd, cd, a2);   // s_j . y_c
        a4 = fma(yd, cd, a4);   // y_j . y_c
      }
    a0 = wave_sum_d_dpp(a0); a1 = wave_sum_d_dpp(a1); a2 = wave_sum_d_dpp(a2); a4 = wave_sum_d_dpp(a4);
    if (lane == 0) {
      double* o = w.part + (size_t)(jj * 5) * nparts + p;
      o[0] = a0; o[(size_t)nparts] = a1; o[(size_t)2 * nparts] = a2; o[(size_t)3 * nparts] = 0.0;
      o[(size_t)4 * nparts] = a4;
    }
  };
  if constexpr (PIPE) {
    f32x4 sA[U], yA[U], sB[U], yB[U];
    int jj = pgrp * per;
    if (jj < j_end) load_pair(jj, sA, yA);
    for (; jj < j_end; jj += 2) {
      if (jj + 1 < j_end) load_pair(jj + 1, sB, yB);
      dot_pair(jj, sA, yA);
      if (jj + 1 < j_end) {
        if (jj + 2 < j_end) load_pair(jj + 2, sA, yA);
        dot_pair(jj + 1, sB, yB);
      }
    }
  } else {
    for (int jj = pgrp * per; jj < j_end; ++jj) {
      const int slot = (head + jj) % S;
      const float* __restrict__ sj = w.S + (size_t)slot * nn;
      const float* __restrict__ yj = w.Y + (size_t)slot * nn;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a4 = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t idx = base + (size_t)(u * 256 + tid) * 4;
        const f32x4 s4 = ldh4<NT>(sj + idx);
        const f32x4 y4 = ldh4<NT>(yj + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double sd = (double)s4[e], yd = (double)y4[e];
          const double gd = (double)gv[u][e], cd = (double)yv[u][e];
          a0 = fma(sd, gd, a0);   // s_j . g
          a1 = fma(yd, gd, a1);   // y_j . g
          a2 = fma(sd, cd, a2);   // s_j . y_c
          a4 = fma(yd, cd, a4);   // y_j . y_c
        }
      }
      a0 = wave_sum_d_dpp(a0); a1 = wave_sum_d_dpp(a1); a2 = wave_sum_d_dpp(a2); a4 = wave_sum_d_dpp(a4);
      if (lane == 0) {
        double* o = w.part + (size_t)(jj * 5) * nparts + p;
        o[0] = a0; o[(size_t)nparts] = a1; o[(size_t)2 * nparts] = a2; o[(size_t)3 * nparts] = 0.0;
        o[(size_t)4 * nparts] = a4;
      }
    }
  }
  if (pgrp != 0) return;
  gmax = wave_max(gmax);
  gl1 = wave_sum_d(gl1); gg = wave_sum_d(gg); gs = wave_sum_d(gs); gy = wave_sum_d(gy); sy = wave_sum_d(sy); yy = wave_sum_d(yy);
  if (lane == 0) {
    double* o = w.partd + p;
    o[0] = (double)gmax; o[(size_t)nparts] = gl1; o[(size_t)2 * nparts] = gg; o[(size_t)3 * nparts] = gs;
    o[(size_t)4 * nparts] = gy; o[(size_t)5 * nparts] = sy; o[(size_t)6 * nparts] = yy;
  }
}

// ---- fixed-order reduction of the partials: one workgroup per dot product ----------------------
__global__ __launch_bounds__(256) void reduce_kernel(const CState* st, CWs w, int hist, int nparts) {
  __shared__ double red[256];
  const int m = st->hist_len;
  int dot = blockIdx.x;                       // [0, 5*m) history dots, then NSCAL scalars
  const bool scalar = dot >= 5 * m;
  if (scalar) dot = 5 * hist + (dot - 5 * m);
  if (dot >= 5 * hist + NSCAL) return;
  const double* __restrict__ src = w.part + (size_t)dot * nparts;                       // history products
  const double* __restrict__ srcd = w.partd + (size_t)(scalar ? dot - 5 * hist : 0) * nparts;   // the step's scalars
  const bool is_max = dot == 5 * hist;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    const double v = scalar ? srcd[i] : src[i];
    acc = is_max ? fmax(acc, v) : acc + v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      red[threadIdx.x] = is_max ? fmax(red[threadIdx.x], red[threadIdx.x + s]) : red[threadIdx.x] + red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) w.dots[scalar ? 5 * MAX_HIST + (dot - 5 * hist) : dot] = red[0];
}

// ---- control flow + recursion on coefficients ---------------------------------------------------
// Diagnostic build (-DSTV_SOLVE_STAMPS): thread 0 leaves 100 MHz wall-clock stamps at the phase
// boundaries of the solve kernel; tools/solve_stamps.py reads them back.
#ifdef STV_SOLVE_STAMPS
__device__ unsigned long long g_solve_stamps[8];
#define SOLVE_STAMP(k) do { if (threadIdx.x == 0) g_solve_stamps[k] = wall_clock64(); } while (0)
#else
#define SOLVE_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256) void solve_kernel(CState* st, CWs w, int hist, float lr, float tol_grad,
                                                    float tol_change) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int S = hist + 1;
  // ONE [m][m] double table in logical order holds both products: T[i][j] = s_i.y_j for i < j (the recursion
  // only ever uses s_i.y_j of an OLDER s with a NEWER y) and T[i][j] = y_i.y_j for i >= j (symmetric: the
  // entry above the diagonal is read from its mirror image) - two double tables would not fit the 160 KB of
  // LDS at history 100.  The odd row pitch keeps row walks across lanes and column walks free of bank conflicts.
  const int P = hist | 1;
  double* sT = reinterpret_cast<double*>(smem_raw);
  __shared__ int sh_skip, sh_pushed, sh_m, sh_head, sh_cslot, sh_mold;
  __shared__ float sh_newro, sh_H;
  __shared__ double sh_gs[MAX_S], sh_gy[MAX_S], sh_ro[MAX_S];
  __shared__ double sh_bp[2][128];       // partial sums of the second walk (indices below / from 64)
  __shared__ double sh_cy[128];          // the final y-coefficients, for every thread of the second walk
  const int tid = threadIdx.x;
  const double* D = w.dots;
  const double* SC = w.dots + 5 * MAX_HIST;
  const double gmax = SC[0], gl1 = SC[1], gg = SC[2], gsc = SC[3], gyc = SC[4], ys = SC[5], yy = SC[6];
  SOLVE_STAMP(0);

  if (tid == 0) {
    // the 64-byte header is read and written once as a block: a chain of dependent global
    // read-modify-writes on its fields costs ~1 us each on an otherwise idle chip
    struct Hdr { int n_iter, hist_len, head, skip, no_update, pushed, steps_seen, pad0;
                 float t, H_diag, gtd, gmax, ys, yy, cg, pad1; };
    static_assert(sizeof(Hdr) == 64, "header layout");
    Hdr h = *reinterpret_cast<const Hdr*>(st);
    h.steps_seen += 1;
    h.gmax = (float)gmax;
    h.no_update = 0;
    h.pushed = 0;
    const int skip = ((float)gmax <= tol_grad) ? 1 : 0;   // opt_cond: leave every piece of state untouched
    h.skip = skip;
    int pushed = 0;
    const int m_old = h.hist_len;
    const int cslot = (h.head + m_old) % S;
    if (!skip) {
      h.n_iter += 1;
      if (h.n_iter == 1) {
        h.hist_len = 0;
        h.head = 0;
        h.H_diag = 1.0f;
      } else {
        h.ys = (float)ys;
        h.yy = (float)yy;
        if ((float)ys > 1e-10f) {
          pushed = 1;
          if (h.hist_len == hist) h.head = (h.head + 1) % S;
          else h.hist_len += 1;
          st->ro[cslot] = 1.0f / (float)ys;
          h.H_diag = (float)ys / (float)yy;
          sh_newro = 1.0f / (float)ys;
        }
      }
      if (h.n_iter == 1) {
        const float inv = 1.0f / (float)gl1;
        h.t = ((inv < 1.0f) ? inv : 1.0f) * lr;
      } else {
        h.t = lr;
      }
      h.pushed = pushed;
    }
    sh_skip = skip; sh_pushed = pushed; sh_m = h.hist_len; sh_head = h.head; sh_cslot = cslot;
    sh_mold = (h.n_iter == 1) ? 0 : m_old;
    sh_H = h.H_diag;
    // cg / gtd / no_update are filled in by the recursion below (lane 0 = this thread)
    *reinterpret_cast<Hdr*>(st) = h;
  }
  __syncthreads();
  SOLVE_STAMP(1);
  if (sh_skip) return;
  const int m = sh_m, head = sh_head, cslot = sh_cslot, m_old = sh_mold;
  const int old_head = (sh_pushed && m_old == hist) ? (head + S - 1) % S : head;

  // install the pushed pair's row/column: old logical index jj sat in slot (old_head + jj) % S
  if (sh_pushed) {
    for (int jj = tid; jj < m_old; jj += 256) {
      const int slot = (old_head + jj) % S;
      const double s_j_yc = D[jj * 5 + 2], y_j_sc = D[jj * 5 + 3], y_j_yc = D[jj * 5 + 4];
      w.SY[(size_t)slot * S + cslot] = s_j_yc;     // s_j . y_c
      w.SY[(size_t)cslot * S + slot] = y_j_sc;     // s_c . y_j
      w.YY[(size_t)slot * S + cslot] = y_j_yc;
      w.YY[(size_t)cslot * S + slot] = y_j_yc;
    }
    if (tid == 0) {
      w.SY[(size_t)cslot * S + cslot] = ys;
      w.YY[(size_t)cslot * S + cslot] = yy;
    }
  }
  // g-dots in (new) logical order; the pushed pair is the newest logical index m-1
  for (int i = tid; i < m; i += 256) {
    const int slot = (head + i) % S;
    sh_ro[i] = (sh_pushed && slot == cslot) ? (double)sh_newro : (double)st->ro[slot];
    if (sh_pushed && slot == cslot) {
      sh_gs[i] = gsc;
      sh_gy[i] = gyc;
    } else {
      const int jj = (slot - old_head + S) % S;     // its index during pass A
      sh_gs[i] = D[jj * 5 + 0];
      sh_gy[i] = D[jj * 5 + 1];
    }
  }
  __threadfence_block();
  __syncthreads();
  SOLVE_STAMP(2);
  {
    // table fill: thread = (row parity, column), a batch of rows of loads in flight per thread (walking the
    // rows one load at a time made this fill, not the recursion, the longest part of the kernel)
    const int j = tid & 127, i0 = tid >> 7;
    if (j < m) {
      int col = head + j;
      if (col >= S) col -= S;
      constexpr int FB = 50;                   // rows per batch: FB loads in flight, then the stores (history 100: ONE batch per thread)
      for (int ib = i0; ib < m; ib += 2 * FB) {
        double v[FB];
#pragma unroll
        for (int k = 0; k < FB; ++k) {
          int i = ib + 2 * k;
          if (i >= m) i = m - 1;               // clamped, not branched: keeps the batch one burst of loads
          int rs = head + i;
          if (rs >= S) rs -= S;
          const double* __restrict__ src = (i < j) ? w.SY : w.YY;
          v[k] = src[(size_t)rs * S + col];
        }
#pragma unroll
        for (int k = 0; k < FB; ++k) {
          const int i = ib + 2 * k;
          if (i < m) sT[i * P + j] = v[k];
        }
      }
    }
  }
  __syncthreads();
  SOLVE_STAMP(3);

  // ---- torch's two-loop on coefficients; a lane owns logical indices lane, lane+64 ----
  // All four waves run the first (serial) walk redundantly - each on its own SIMD, same registers,
  // same result - so that the second walk, a plain matrix-vector product, can be split over them
  // without a hand-over of state; the third walk and the tail are wave 0's alone.
  // Written as broadcast + rank-1 updates: when alpha_i (beta_i) becomes known its owner lane
  // broadcasts it with v_readlane and every other lane folds it into the running sum of the
  // indices it owns, so the serial chain per iteration is one FMA + one readlane instead of a
  // 6-step cross-lane reduction.
  const int lane = tid & 63, wave4 = tid >> 6;
  const int j0 = lane, j1 = lane + 64;
  auto bcast = [](double v, int src) -> double {          // src is wave-uniform
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xFFFFFFFFll), src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  const double gs0 = j0 < m ? sh_gs[j0] : 0.0, gs1 = j1 < m ? sh_gs[j1] : 0.0;
  const double gy0 = j0 < m ? sh_gy[j0] : 0.0, gy1 = j1 < m ? sh_gy[j1] : 0.0;
  const double ro0 = j0 < m ? sh_ro[j0] : 0.0, ro1 = j1 < m ? sh_ro[j1] : 0.0;
  double cy0 = 0.0, cy1 = 0.0, cs0 = 0.0, cs1 = 0.0, al0 = 0.0, al1 = 0.0;
  double a0 = 0.0, a1 = 0.0;                // sum_{j > own} cy_j * SY[own][j], built incrementally
  double cg = -1.0;
  // The three walks are serial in i; what each step needs from LDS does not depend on the chain,
  // so it is fetched four steps ahead (CH values per owned index) while the previous four run.
  // Each walk is split at index 64 (owner slot 0 / slot 1 of a lane) into two instantiations: with
  // the slot a compile-time constant a step is ~25 instructions; with run-time selects it was ~65,
  // and the kernel is bound by exactly that instruction count (one wave, one serial chain).
  constexpr int CH = 4;
  using hi_t = std::integral_constant<bool, true>;
  using lo_t = std::integral_constant<bool, false>;
  const int r0 = (j0 < m ? j0 : m - 1) * P, r1 = (j1 < m ? j1 : m - 1) * P;   // row bases (clamped: unused lanes)
  const int q0 = j0 < m ? j0 : m - 1, q1 = j1 < m ? j1 : m - 1;               // column indices, same clamp
  const float ro0f = (float)ro0, ro1f = (float)ro1;
  // walk 1, i = m-1 .. 0:  al_i = ro_i * (cg*(g.s_i) + a_i),  a_own -= al_i * SY[own][i]
  auto walk1 = [&](auto HI, int ifrom, int ito) {
    constexpr bool hi = decltype(HI)::value;
    if (ifrom < ito) return;
    double n0[CH], n1[CH];
    auto fetch = [&](int ib) {                 // column entries s_own . y_i, zero unless own < i
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int i = max(ib - k, 0);
        const double v0 = sT[r0 + i];
        if (hi) {                              // i >= 64: every slot-0 index is older than i
          n0[k] = v0;
          const double v1 = sT[r1 + i];
          n1[k] = (j1 < i) ? v1 : 0.0;
        } else {                               // below 64 no slot-1 index is older than i
          n0[k] = (j0 < i) ? v0 : 0.0;
        }
      }
    };
    fetch(ifrom);
    for (int ib = ifrom; ib >= ito; ib -= CH) {
      double c0[CH], c1[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) { c0[k] = n0[k]; c1[k] = hi ? n1[k] : 0.0; }
      if (ib - CH >= ito) fetch(ib - CH);
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int i = ib - k;
        if (i < ito) break;
        const int src = hi ? i - 64 : i;       // wave-uniform owner lane
        const double cand = (double)(float)((float)(cg * (hi ? gs1 : gs0) + (hi ? a1 : a0)) * (hi ? ro1f : ro0f));
        const double al = bcast(cand, src);    // fp32-rounded like torch's al[i]
        if (lane == src) {
          if (hi) { al1 = al; cy1 = -al; } else { al0 = al; cy0 = -al; }
        }
        a0 -= al * c0[k];
        if (hi) a1 -= al * c1[k];
      }
    }
  };
  if (m > 0) {
    walk1(hi_t{}, m - 1, 64);
    walk1(lo_t{}, m - 1 < 63 ? m - 1 : 63, 0);
  }
  SOLVE_STAMP(4);
  const double H = (double)sh_H;
  cg *= H; cy0 *= H; cy1 *= H;
  // walk 2: b_own = cg*(g.y_own) + sum_j cy_j * YY[own][j]  (no recursion: cy is final) - a plain symmetric matrix-vector
  // product.  Round 5: all 256 threads take part - thread (row = tid & 127, part = tid >> 7) sums its row over two
  // 32-index chunks with every LDS read independent of the others (the per-wave form walked its 32 indices with a
  // broadcast per index: 4.0 us of the kernel's 23.4 by the phase stamps, tools/solve_stamps.py).  Same chunks, same order
  // inside a chunk, same grouping of the four chunk sums as before: the coefficients are bit-identical.
  double b0 = cg * gy0, b1 = cg * gy1;
  {
    if (wave4 == 0) {
      sh_cy[j0] = cy0;
      sh_cy[j1] = cy1;
    }
    __syncthreads();
    const int row = tid & 127, part = tid >> 7;
    double chunk_sum[2] = {0.0, 0.0};
    if (row < m) {
      const int rb = row * P;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int jb = (2 * part + c) * 32;
        double acc = 0.0;
        // sixteen table entries and coefficients in flight, then their FMAs in index order: no branch inside (an index
        // past the history reads a valid entry against a coefficient that is zero: acc + 0 == acc)
#pragma unroll
        for (int kb = 0; kb < 32; kb += 16) {
          double cv[16], tv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int j = jb + kb + k;
            const int jc = j < m ? j : m - 1;
            cv[k] = sh_cy[j];
            tv[k] = sT[(jc <= row) ? rb + jc : jc * P + row];     // y_row . y_j: stored at [max][min]
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) acc = fma(cv[k], tv[k], acc);
        }
        chunk_sum[c] = acc;
      }
    }
    sh_bp[part][row] = chunk_sum[0] + chunk_sum[1];
    __syncthreads();
    if (tid >= 64) return;
    b0 = b0 + (sh_bp[0][lane] + sh_bp[1][lane]);
    b1 = b1 + (sh_bp[0][64 + lane] + sh_bp[1][64 + lane]);
  }
  SOLVE_STAMP(5);
  // walk 3, i = 0 .. m-1:  cs_i = al_i - ro_i * b_i,  b_own += cs_i * SY[i][own]  (own newer than i)
  auto walk3 = [&](auto HI, int ifrom, int ito) {       // i = ifrom .. ito-1
    constexpr bool hi = decltype(HI)::value;
    if (ifrom >= ito) return;
    double n0[CH], n1[CH];
    auto fetch = [&](int ib) {                 // row entries s_i . y_own, zero unless own > i
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int i = min(ib + k, m - 1);
        const double v1 = sT[i * P + q1];
        if (!hi) {                             // i < 64: every slot-1 index is newer than i
          const double v0 = sT[i * P + q0];
          n0[k] = (j0 > i) ? v0 : 0.0;
          n1[k] = v1;
        } else {                               // from 64 on no slot-0 index is newer than i
          n1[k] = (j1 > i) ? v1 : 0.0;
        }
      }
    };
    fetch(ifrom);
    for (int ib = ifrom; ib < ito; ib += CH) {
      double c0[CH], c1[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) { c0[k] = hi ? 0.0 : n0[k]; c1[k] = n1[k]; }
      if (ib + CH < ito) fetch(ib + CH);
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int i = ib + k;
        if (i >= ito) break;
        const int src = hi ? i - 64 : i;
        const double be = (double)(float)((float)(hi ? b1 : b0) * (hi ? ro1f : ro0f));
        const double cand = (double)(float)((float)(hi ? al1 : al0) - (float)be);
        const double csi = bcast(cand, src);
        if (lane == src) {
          if (hi) cs1 = csi; else cs0 = csi;
        }
        if (!hi) b0 += csi * c0[k];
        b1 += csi * c1[k];
      }
    }
  };
  walk3(lo_t{}, 0, m < 64 ? m : 64);
  walk3(hi_t{}, 64, m);
  SOLVE_STAMP(6);
  double gtd = 0.0;
  if (j0 < m) gtd += cy0 * sh_gy[j0] + cs0 * sh_gs[j0];
  if (j1 < m) gtd += cy1 * sh_gy[j1] + cs1 * sh_gs[j1];
  gtd = wave_sum_d(gtd) + cg * gg;
  if (j0 < m) { st->cs[(head + j0) % S] = (float)cs0; st->cy[(head + j0) % S] = (float)cy0; }
  if (j1 < m) { st->cs[(head + j1) % S] = (float)cs1; st->cy[(head + j1) % S] = (float)cy1; }
  if (lane == 0) {
    st->cg = (float)cg;
    st->gtd = (float)gtd;
    st->no_update = ((float)gtd > -tol_change) ? 1 : 0;
  }
  SOLVE_STAMP(7);
}

// ---- pass B: form the direction, move x, remember g ------------------------------------------------
// d = cg g + sum_j cy_j y_j + cs_j s_j is a sum of 2m + 1 terms that cancel to a result far smaller than its
// largest partial sums (the two-loop recursion deflates q step by step; here the terms arrive in whatever order
// the history is walked).  Accumulated in fp32 the direction was 7-10x further from the float64 update than
// torch's fp32 vector recursion once m > 60 (tests/test_gpu_lbfgs_long.py; CPU emulation of this algorithm:
// fp64 inner products change nothing, fp64 accumulation HERE brings it to the reference's own level), so the
// accumulators are double (ACC64): every product of two floats is exact in double, the sum is rounded to
// fp32 once.  The sweep stays bandwidth-bound: 2 conversions + 2 DP FMAs per element and pair beside 8 bytes.
template <int U, bool ACC64, bool NT = false>
__global__ __launch_bounds__(256) void pass_b_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                     const CState* st, CWs w, size_t n, size_t nn, int hist) {
  if (st->skip) return;
  const int S = hist + 1;
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * (256 * 4 * U);
  const int m = st->hist_len, head = st->head;
  const float cg = st->cg, t = st->t;
  const bool move = st->no_update == 0;
  using acc_t = typename std::conditional<ACC64, double, float>::type;
  acc_t acc[U][4];
  f32x4 gv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t idx = base + (size_t)(u * 256 + tid) * 4;
    gv[u] = ld4_guard(g, idx, n);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[u][e] = (acc_t)cg * (acc_t)gv[u][e];
  }
  // newest pair first: pass A walked the history oldest-to-newest just before, so its tail is what
  // the Infinity Cache still holds
  for (int jj = m - 1; jj >= 0; --jj) {
    const int slot = (head + jj) % S;
    const acc_t cs = (acc_t)st->cs[slot], cy = (acc_t)st->cy[slot];
    const float* __restrict__ sj = w.S + (size_t)slot * nn;
    const float* __restrict__ yj = w.Y + (size_t)slot * nn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t idx = base + (size_t)(u * 256 + tid) * 4;
      const f32x4 s4 = ldh4<NT>(sj + idx);
      const f32x4 y4 = ldh4<NT>(yj + idx);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ACC64) acc[u][e] = fma(cs, (double)s4[e], fma(cy, (double)y4[e], acc[u][e]));
        else acc[u][e] = fmaf(cs, s4[e], fmaf(cy, y4[e], acc[u][e]));
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t idx = base + (size_t)(u * 256 + tid) * 4;
    f32x4 dv;
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[e] = (float)acc[u][e];
    *reinterpret_cast<f32x4*>(w.d + idx) = dv;
    *reinterpret_cast<f32x4*>(w.prev_g + idx) = gv[u];
    if (move) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (idx + e < n) x[idx + e] = fmaf(t, dv[e], x[idx + e]);
    }
  }
}

}  // namespace

extern "C" size_t stv_lbfgsc_state_bytes(int history) {
  (void)history;
  return sizeof(CState);
}

extern "C" size_t stv_lbfgsc_workspace_bytes(size_t n, int history) {
  const size_t nn = align_up(n, 4096);
  const int S = history + 1;
  const int tile = tile_floats(n);
  const size_t nparts = (nn / tile) * 4;
  // (two floats per double: the two product tables, the inner-product block, the scalars' partial sums)
  size_t floats = nn * (2 + 2 * (size_t)S) + 2 * 2 * align_up((size_t)S * S, 64) +
                  2 * align_up(5 * (size_t)MAX_HIST + NSCAL, 64) + 2 * align_up((size_t)NSCAL * nparts, 64) +
                  2 * 5 * (size_t)history * nparts + 64;
  return floats * sizeof(float);
}

namespace {
// sweep B keeps no partial sums, so its tile is free of sweep A's: fat tiles amortise sweep A's per-pair
// reductions, sweep B only wants enough workgroups in flight
inline int tile_floats_b(size_t n) {
  static const int forced = getenv("STV_LBFGS_TILE_B") ? atoi(getenv("STV_LBFGS_TILE_B")) : 0;   // tuning aid
  if (forced == 1024 || forced == 2048 || forced == 4096) return forced;
  if (n >= (size_t)4096 * 256) return 4096;
  if (n >= (size_t)2048 * 256) return 2048;
  return 1024;
}
struct StepGeom { size_t nn; int tile, ntiles, nparts, pgroups; CWs w; };
inline StepGeom step_geom(void* workspace, size_t n, int history) {
  StepGeom g;
  g.nn = align_up(n, 4096);
  g.tile = tile_floats(n);
  g.ntiles = (int)(g.nn / g.tile);
  g.nparts = g.ntiles * 4;
  // sweep A: enough workgroups to cover the HBM latency (>= ~768 where the tile count alone does not give them)
  static const int forced_pg = getenv("STV_LBFGS_PGROUPS") ? atoi(getenv("STV_LBFGS_PGROUPS")) : 0;   // tuning aid
  // (measured: 256^2, 192 tiles -> 4 groups: step 0.549 -> 0.515 ms; 512^2, 384 tiles: 2 groups change nothing)
  g.pgroups = forced_pg > 0 ? forced_pg : (g.ntiles >= 256 ? 1 : (768 + g.ntiles - 1) / g.ntiles);
  if (g.pgroups > 4) g.pgroups = 4;
  if (g.pgroups < 1) g.pgroups = 1;
  g.w = carve(workspace, n, history, g.nparts);
  return g;
}
}  // namespace

// First half of a step: sweep A + the fixed-order reduction.  Leaves every inner product of the step
// (5 per history pair + NSCAL scalars) as doubles in the workspace, at stv_lbfgsc_dots_offset().
extern "C" int stv_lbfgsc_dots(const float* grad, void* state, void* workspace, size_t n, int history, int m_max,
                               void* stream) {
  if (!grad || !state || !workspace || n == 0) return STV_ERR_ARG;
  if (history < 1 || history > MAX_HIST) return STV_ERR_ARG;
  if (m_max < 0) m_max = 0;
  if (m_max > history) m_max = history;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CState* s = static_cast<CState*>(state);
  const StepGeom g = step_geom(workspace, n, history);
  static const int nt_mask = getenv("STV_LBFGS_NT") ? atoi(getenv("STV_LBFGS_NT")) : 3;
  static const int pipe_a = getenv("STV_LBFGS_PIPE") ? atoi(getenv("STV_LBFGS_PIPE")) : STV_LBFGS_PIPE_DEFAULT;
#define STV_LAUNCH_PASS_A(U_)                                                                                                      \
  do {                                                                                                                             \
    if ((nt_mask & 1) && pipe_a) hipLaunchKernelGGL((pass_a_kernel<U_, true, true>), dim3(g.ntiles * g.pgroups), dim3(256), 0, st, grad, s, g.w, n, g.nn, history, g.nparts, g.ntiles, g.pgroups); \
    else if (nt_mask & 1) hipLaunchKernelGGL((pass_a_kernel<U_, true>), dim3(g.ntiles * g.pgroups), dim3(256), 0, st, grad, s, g.w, n, g.nn, history, g.nparts, g.ntiles, g.pgroups); \
    else hipLaunchKernelGGL((pass_a_kernel<U_, false>), dim3(g.ntiles * g.pgroups), dim3(256), 0, st, grad, s, g.w, n, g.nn, history, g.nparts, g.ntiles, g.pgroups);          \
  } while (0)
  if (g.tile == 4096) STV_LAUNCH_PASS_A(4);
  else if (g.tile == 2048) STV_LAUNCH_PASS_A(2);
  else STV_LAUNCH_PASS_A(1);
#undef STV_LAUNCH_PASS_A
  hipLaunchKernelGGL(reduce_kernel, dim3(5 * m_max + NSCAL), dim3(256), 0, st, s, g.w, history, g.nparts);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

// Second half: torch's control flow + the recursion on coefficients (from the inner products), then
// sweep B: d, x += t*d, prev_g = g.
extern "C" int stv_lbfgsc_apply(float* x, const float* grad, void* state, void* workspace, size_t n, int history,
                                float lr, float tol_grad, float tol_change, void* stream) {
  if (!x || !grad || !state || !workspace || n == 0) return STV_ERR_ARG;
  if (history < 1 || history > MAX_HIST) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CState* s = static_cast<CState*>(state);
  const StepGeom g = step_geom(workspace, n, history);
  const size_t lds = (size_t)history * (size_t)(history | 1) * sizeof(double);
  if (stv_set_max_lds(reinterpret_cast<const void*>(&solve_kernel), MAX_HIST * (MAX_HIST | 1) * (int)sizeof(double)) != STV_OK)
    return STV_ERR_LAUNCH;
  hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(256), lds, st, s, g.w, history, lr, tol_grad, tol_change);
  // A/B aid: STV_LBFGS_ACC=f32 restores the fp32 accumulation of the direction (less accurate, see pass_b_kernel)
  static const bool acc64 = !(getenv("STV_LBFGS_ACC") && strcmp(getenv("STV_LBFGS_ACC"), "f32") == 0);
  const int tile_b = tile_floats_b(n);
  const int ntiles_b = (int)(g.nn / tile_b);
  static const int nt_mask_b = getenv("STV_LBFGS_NT") ? atoi(getenv("STV_LBFGS_NT")) : 3;
#define STV_LAUNCH_PASS_B(U_)                                                                                              \
  do {                                                                                                                     \
    if (acc64 && (nt_mask_b & 2)) hipLaunchKernelGGL((pass_b_kernel<U_, true, true>), dim3(ntiles_b), dim3(256), 0, st, x, grad, s, g.w, n, g.nn, history); \
    else if (acc64) hipLaunchKernelGGL((pass_b_kernel<U_, true>), dim3(ntiles_b), dim3(256), 0, st, x, grad, s, g.w, n, g.nn, history); \
    else hipLaunchKernelGGL((pass_b_kernel<U_, false>), dim3(ntiles_b), dim3(256), 0, st, x, grad, s, g.w, n, g.nn, history);      \
  } while (0)
  if (tile_b == 4096) STV_LAUNCH_PASS_B(4);
  else if (tile_b == 2048) STV_LAUNCH_PASS_B(2);
  else STV_LAUNCH_PASS_B(1);
#undef STV_LAUNCH_PASS_B
  STV_CHECK_LAUNCH();
  return STV_OK;
}

// Where the inner products live: byte offset into the workspace of `count` doubles; entry `max_index`
// is max|g| (combine across shards with MAX), every other entry is a sum (combine with SUM).
extern "C" size_t stv_lbfgsc_dots_offset(size_t n, int history, int* count, int* max_index) {
  const size_t nn = align_up(n, 4096);
  const int S = history + 1;
  if (count) *count = 5 * MAX_HIST + NSCAL;
  if (max_index) *max_index = 5 * MAX_HIST;
  return (nn * (2 + 2 * (size_t)S) + 2 * 2 * align_up((size_t)S * S, 64)) * sizeof(float);
}

extern "C" int stv_lbfgsc_step(float* x, const float* grad, void* state, void* workspace, size_t n, int history,
                               int m_max, float lr, float tol_grad, float tol_change, void* stream) {
  if (!x) return STV_ERR_ARG;
  const int rc = stv_lbfgsc_dots(grad, state, workspace, n, history, m_max, stream);
  if (rc != STV_OK) return rc;
  return stv_lbfgsc_apply(x, grad, state, workspace, n, history, lr, tol_grad, tol_change, stream);
}

#ifdef STV_SOLVE_STAMPS
extern "C" int stv_debug_solve_stamps(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_solve_stamps), sizeof(g_solve_stamps)) == hipSuccess ? 0 : 1;
}
#endif
