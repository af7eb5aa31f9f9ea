// Command-buffer executor: the host (plan.py) lowers a model's forward +
// backward schedule to a flat array of stv_op_t once; every optimisation step
// then costs one call that enqueues all kernels from C++, or - with use_graph -
// one hipGraphLaunch of the schedule captured on first use.
#include <stdlib.h>

#include <vector>

#include "stv_common.h"

struct stv_program {
  std::vector<stv_op_t> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t captured_on = nullptr;
  // side lane (ops flagged STV_LANE_SIDE): a second stream forked from / joined to the caller's
  // stream with events, so that small latency-bound chains overlap the main chain
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> events;
  // host-side tap arrays of GRAM_MULTI ops, owned here (the ops' p0 points into them)
  std::vector<std::vector<stv_gram_tap_t>> tap_tables;
};

namespace {

int run_op(const stv_op_t& op, void* st) {
  stv_op_t o = op;
  o.flags &= ~(STV_LANE_SIDE | STV_LANE_JOIN);      // scheduling hints, not kernel flags
  switch (o.op) {
    case STV_OP_CONV_FIRST_FWD:   // p3 (optional): weights packed by stv_conv_first_pack
      if (o.p3 && o.q1)    // the layer is a style tap: q1 = Gram slabs (stv_gram_partial's output, fused)
        return stv_conv_first_fwd_gram(static_cast<const float*>(o.p0), static_cast<const float*>(o.p3),
                                       static_cast<const float*>(o.p2), o.q0, static_cast<float*>(o.q1), o.H, o.W, o.cin,
                                       o.cout, o.dtype, st);
      if (o.p3)
        return stv_conv_first_fwd_packed(static_cast<const float*>(o.p0), static_cast<const float*>(o.p3),
                                         static_cast<const float*>(o.p2), o.q0, o.H, o.W, o.cin, o.cout, o.dtype, st);
      return stv_conv_first_fwd(static_cast<const float*>(o.p0), static_cast<const float*>(o.p1),
                                static_cast<const float*>(o.p2), o.q0, o.H, o.W, o.cin, o.cout, o.dtype, st);
    case STV_OP_CONV_FIRST_DGRAD:  // p2 (optional): packed weights
      if (o.p2)
        return stv_conv_first_dgrad_packed(o.p0, static_cast<const float*>(o.p2), static_cast<float*>(o.q0), o.H,
                                           o.W, o.cin, o.cout, o.dtype, st);
      return stv_conv_first_dgrad(o.p0, static_cast<const float*>(o.p1), static_cast<float*>(o.q0), o.H,
                                  o.W, o.cin, o.cout, o.dtype, st);
    case STV_OP_CONV:
      if (o.flags & STV_POOL_ROUTE)   // dgrad in front of a max-pool: p2 = arg-max byte map, q1 = full-resolution gradient
        return stv_conv_igemm_route(o.p0, o.p1, o.p2, o.q1, o.H, o.W, o.cin, o.cout, o.flags & ~STV_POOL_ROUTE, o.dtype, st);
      if (o.q2 && o.q3)   // dgrad with the Gram-backward 1x1 term of the same output fused in (q2 = x2, q3 = w2, n = cin2)
        return stv_conv_igemm_dual(o.p0, o.p1, o.q2, o.q3, o.p3, o.q0, o.H, o.W, o.cin, (int)o.n, o.cout,
                                   o.flags, o.dtype, st);
      if (o.q1)    // forward conv with the following MaxPool2d(2,2) fused into its epilogue
        return stv_conv_igemm_pool(o.p0, o.p1, static_cast<const float*>(o.p2), o.q0, o.q1, o.q2, o.H, o.W, o.cin,
                                   o.cout, o.flags, o.dtype, st);    // q2 (optional, q3 unset): arg-max byte map
      return stv_conv_igemm(o.p0, o.p1, static_cast<const float*>(o.p2), o.p3, o.q0, o.H, o.W, o.cin,
                            o.cout, o.taps, o.flags, o.dtype, st);
    case STV_OP_POOL_FWD:
      return stv_maxpool_fwd(o.p0, o.q0, o.H, o.W, o.cin, o.dtype, st);
    case STV_OP_POOL_BWD:
      return stv_maxpool_bwd(o.p0, o.p1, o.q0, o.H, o.W, o.cin, o.flags, o.dtype, st);
    case STV_OP_RELU_FWD:
      return stv_relu_fwd(o.p0, o.q0, (size_t)o.n, o.dtype, st);
    case STV_OP_RELU_BWD:
      return stv_relu_bwd(o.p0, o.p1, o.q0, (size_t)o.n, o.flags, o.dtype, st);
    case STV_OP_GRAM_PARTIAL:
      return stv_gram_partial(o.p0, static_cast<float*>(o.q0), (int)o.n, o.cin, o.dtype, st);
    case STV_OP_GRAM_FINISH:
      // p0 partials, p1 target, p2 coef_dev; q0 gram_out, q1 loss_part, q2 sgrad
      // f0 clamp_max, f1 norm, f2 coef
      return stv_gram_finish(static_cast<const float*>(o.p0), static_cast<const float*>(o.p1),
                             static_cast<float*>(o.q0), static_cast<float*>(o.q1), o.q2, (int)o.n, o.cin,
                             o.f0, o.f1, o.f2, static_cast<const float*>(o.p2), o.dtype, st);
    case STV_OP_CONTENT_LOSS:
      if (o.q1)   // loss + gradient in one pass (q1 = dF written, f0 = coefficient)
        return stv_content_loss_grad(o.p0, o.p1, static_cast<float*>(o.q0), o.q1, (size_t)o.n, o.f0, o.dtype, st);
      return stv_content_loss(o.p0, o.p1, static_cast<float*>(o.q0), (size_t)o.n, o.dtype, st);
    case STV_OP_CONTENT_GRAD:
      return stv_content_grad(o.p0, o.p1, o.q0, (size_t)o.n, o.f0, static_cast<const float*>(o.p2), o.flags,
                              o.dtype, st);
    case STV_OP_LOSS_COMBINE:
      // p0 parts, p1 table, p2 scale; q0 losses, q1 scores; cin = n_terms; f0 style_w, f1 content_w
      // q2, q3 (optional): history ring and its device counter, n = ring capacity; p3 (optional, written): host-visible
      // record count of a ring in host memory
      return stv_loss_combine_log(static_cast<const float*>(o.p0), static_cast<const int32_t*>(o.p1),
                                  static_cast<const float*>(o.p2), o.cin, o.f0, o.f1,
                                  static_cast<float*>(o.q0), static_cast<float*>(o.q1), static_cast<float*>(o.q2),
                                  (int)o.n, static_cast<uint32_t*>(o.q3), static_cast<uint32_t*>(const_cast<void*>(o.p3)), st);
    case STV_OP_GRAM_MULTI:
      return stv_gram_multi(static_cast<const stv_gram_tap_t*>(o.p0), (int)o.n, o.dtype, st);
    case STV_OP_LBFGS_STEP:
      return stv_lbfgsc_step(static_cast<float*>(o.q0), static_cast<const float*>(o.p0), o.q1, o.q2, (size_t)o.n, o.cin, o.cout,
                             o.f0, o.f1, o.f2, st);
    case STV_OP_MEMSET:
      if (hipMemsetAsync(o.q0, 0, (size_t)o.n, static_cast<hipStream_t>(st)) != hipSuccess)
        return STV_ERR_LAUNCH;
      return STV_OK;
    default:
      return STV_ERR_ARG;
  }
}

// A conv is told the weights of the next 3x3 conv and touches them on its way out (between main loop and
// epilogue): that launch then finds them on chip.  Deep layers at 512^2: -1.5...-3.9 us per launch, step
// 0.965 -> 0.936 ms.  STV_NEXT_W=0: off.  The caller clears the hint after the launch.
void hint_next_weights(const std::vector<stv_op_t>& ops, size_t i) {
  static const bool on = !(getenv("STV_NEXT_W") && atoi(getenv("STV_NEXT_W")) == 0);
  if (!on || ops[i].op != STV_OP_CONV) return;
  for (size_t j = i + 1; j < ops.size(); ++j) {
    const stv_op_t& nx = ops[j];
    if (nx.op == STV_OP_CONV && nx.taps == 9) {
      stv_conv_next_weights(nx.p1, (size_t)9 * nx.cin * nx.cout * (nx.dtype == STV_BF16 ? 2 : 4));
      return;
    }
  }
}

hipEvent_t next_event(stv_program* p, size_t& used) {
  if (used == p->events.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    p->events.push_back(e);
  }
  return p->events[used++];
}

// Ops run in program order on `st`, except ops flagged STV_LANE_SIDE: those go to the program's
// second stream, which waits for everything enqueued on `st` before the op (fork) and is waited for
// by the first later op flagged STV_LANE_JOIN, or by the end of the program (join).  The same
// calls build the dependency edges when `st` is being captured into a graph.
int run_all(stv_program* p, void* st, bool lanes) {
  hipStream_t main_s = static_cast<hipStream_t>(st);
  bool any_side = false;
  if (lanes)
    for (const stv_op_t& o : p->ops) any_side |= (o.flags & STV_LANE_SIDE) != 0;
  if (!any_side) {
    const size_t n_ops = p->ops.size();
    for (size_t i = 0; i < n_ops; ++i) {
      const stv_op_t& o = p->ops[i];
      hint_next_weights(p->ops, i);
      const int rc = run_op(o, st);
      stv_conv_next_weights(nullptr, 0);
      if (rc != STV_OK) return rc;
    }
    return STV_OK;
  }
  if (!p->side && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) return STV_ERR_ALLOC;
  size_t used = 0;
  bool side_busy = false;        // side lane holds work the main lane has not waited for
  bool main_moved = true;        // main lane enqueued work since the side lane last synchronised with it
  auto join = [&]() -> int {
    if (!side_busy) return STV_OK;
    hipEvent_t e = next_event(p, used);
    if (!e || hipEventRecord(e, p->side) != hipSuccess || hipStreamWaitEvent(main_s, e, 0) != hipSuccess)
      return STV_ERR_LAUNCH;
    side_busy = false;
    return STV_OK;
  };
  for (const stv_op_t& o : p->ops) {
    int rc = STV_OK;
    if (o.flags & STV_LANE_SIDE) {
      if (main_moved) {
        hipEvent_t e = next_event(p, used);
        if (!e || hipEventRecord(e, main_s) != hipSuccess || hipStreamWaitEvent(p->side, e, 0) != hipSuccess)
          return STV_ERR_LAUNCH;
        main_moved = false;
      }
      rc = run_op(o, p->side);
      side_busy = true;
    } else {
      if (o.flags & STV_LANE_JOIN) rc = join();
      if (rc == STV_OK) rc = run_op(o, st);
      main_moved = true;
    }
    if (rc != STV_OK) {
      (void)join();              // never leave a captured side stream dangling
      return rc;
    }
  }
  return join();
}

}  // namespace

extern "C" int stv_version(void) { return 100; }

extern "C" int stv_program_create(const stv_op_t* ops, int n_ops, stv_program** out) {
  if (!ops || n_ops <= 0 || !out) return STV_ERR_ARG;
  stv_program* p = new (std::nothrow) stv_program();
  if (!p) return STV_ERR_ALLOC;
  p->ops.assign(ops, ops + n_ops);
  for (stv_op_t& o : p->ops) {
    if (o.op != STV_OP_GRAM_MULTI) continue;
    if (!o.p0 || o.n <= 0 || o.n > 8) {
      delete p;
      return STV_ERR_ARG;
    }
    const stv_gram_tap_t* src = static_cast<const stv_gram_tap_t*>(o.p0);
    p->tap_tables.emplace_back(src, src + o.n);
  }
  size_t k = 0;                    // (pointers taken after the last emplace_back: the vectors no longer move)
  for (stv_op_t& o : p->ops)
    if (o.op == STV_OP_GRAM_MULTI) o.p0 = p->tap_tables[k++].data();
  *out = p;
  return STV_OK;
}

extern "C" int stv_program_run(stv_program* prog, int use_graph, void* stream) {
  if (!prog) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // The legacy default stream cannot be captured: run eagerly there.
  if (!use_graph || st == nullptr) return run_all(prog, stream, true);
  if (prog->exec && prog->captured_on != st) {
    (void)hipGraphExecDestroy(prog->exec);
    (void)hipGraphDestroy(prog->graph);
    prog->exec = nullptr;
    prog->graph = nullptr;
    // the fork/join events and the side stream took part in the destroyed capture: start the new one with fresh ones
    for (hipEvent_t e : prog->events) (void)hipEventDestroy(e);
    prog->events.clear();
    if (prog->side) {
      (void)hipStreamSynchronize(prog->side);
      (void)hipStreamDestroy(prog->side);
      prog->side = nullptr;
    }
  }
  if (!prog->exec) {
    // Eager warm-up first: one-time hipFuncSetAttribute calls must not land inside a capture.
    int rc = run_all(prog, stream, true);
    if (rc != STV_OK) return rc;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();  // do not leave a sticky error behind for the caller's runtime
      return STV_ERR_GRAPH;
    }
    rc = run_all(prog, stream, true);
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(st, &g) != hipSuccess || rc != STV_OK) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      return rc != STV_OK ? rc : STV_ERR_GRAPH;
    }
    hipGraphExec_t e = nullptr;
    if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      return STV_ERR_GRAPH;
    }
    prog->graph = g;
    prog->exec = e;
    prog->captured_on = st;
    return STV_OK;  // the warm-up run already produced this call's results
  }
  if (hipGraphLaunch(prog->exec, st) != hipSuccess) return STV_ERR_GRAPH;
  return STV_OK;
}

// Eager run with a HIP event pair around every op, recorded on the stream the
// kernels are launched on; ms_out[i] = device time of op i.  Synchronises at
// the end (measurement helper for bench.py, never used on the step path).
extern "C" int stv_program_profile(stv_program* prog, void* stream, float* ms_out, int n_out) {
  return stv_program_profile_reps(prog, stream, 1, ms_out, n_out);
}

// The same with every op launched `reps` times back to back inside its event pair and the time
// divided by reps: the event pair's own cost (a few microseconds, comparable to a short kernel)
// is amortised, so ms_out approaches the kernel durations a tracing profiler reports.  Ops that
// accumulate into their output do so `reps` times: the buffers are left meaningless - timing only.
extern "C" int stv_program_profile_reps(stv_program* prog, void* stream, int reps, float* ms_out, int n_out) {
  if (!prog || !ms_out || reps < 1 || n_out < (int)prog->ops.size()) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = prog->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return STV_ERR_ALLOC;
  int rc = STV_OK;
  (void)hipEventRecord(ev[0], st);
  for (size_t i = 0; i < n && rc == STV_OK; ++i) {
    for (int k = 0; k < reps && rc == STV_OK; ++k) {
      hint_next_weights(prog->ops, i);             // the same launches the replayed step makes
      rc = run_op(prog->ops[i], stream);
      stv_conv_next_weights(nullptr, 0);
    }
    (void)hipEventRecord(ev[i + 1], st);
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = STV_ERR_LAUNCH;
  if (rc == STV_OK)
    for (size_t i = 0; i < n; ++i) {
      if (hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]) != hipSuccess) rc = STV_ERR_LAUNCH;
      ms_out[i] /= (float)reps;
    }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

extern "C" int stv_program_op_count(const stv_program* prog) { return prog ? (int)prog->ops.size() : 0; }

extern "C" void stv_program_destroy(stv_program* prog) {
  if (!prog) return;
  if (prog->exec) (void)hipGraphExecDestroy(prog->exec);
  if (prog->graph) (void)hipGraphDestroy(prog->graph);
  for (hipEvent_t e : prog->events) (void)hipEventDestroy(e);
  if (prog->side) (void)hipStreamDestroy(prog->side);
  delete prog;
}
