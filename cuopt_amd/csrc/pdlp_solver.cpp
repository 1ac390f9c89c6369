// C++ host driver of the MI355X-native PDLP solver.  No HIP header is included here: the GPU is
// reached only through the thin C-ABI of include/cuopt_amd/pdlp_device.h.
//
// What lives here is the part of cuOpt's PDLP that is inherently scalar/host logic
// (reference: cuOpt 25.08, LP/ = cpp/src/linear_programming/):
//   - problem conversion (maximise -> minimise, explicit transpose)      mip/problem/problem.cu:53-93
//   - the outer loop and its major-iteration schedule                     LP/pdlp.cu:1081-1185
//   - termination verdicts and limits                                     LP/pdlp.cu:537-802,
//                                                                         LP/termination_strategy/termination_strategy.cu:116-250
//   - KKT restart decision + primal weight update                         LP/restart_strategy/pdlp_restart_strategy.cu:366-641,684-750
// The reference keeps these in 1x1 device kernels + blocking D2H reads; here they are plain C++
// working on the handful of scalars the fused device passes return, and -- unlike the reference --
// the PDHG step loop between two major iterations runs without any host round trip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "cuopt_amd/pdlp_solver.h"
#include "host_parallel.hpp"

namespace {

thread_local std::string g_error;
int fail(int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
#define DEV(expr)                                                          \
  do {                                                                     \
    int rc_ = (expr);                                                      \
    if (rc_ != 0) return fail(rc_, "%s: %s", #expr, pdlpdev_last_error()); \
  } while (0)

using clock_type = std::chrono::steady_clock;
double seconds_since(clock_type::time_point t0)
{
  return std::chrono::duration<double>(clock_type::now() - t0).count();
}

// status codes == CUOPT_TERIMINATION_STATUS_* (constants.h:65-74)
enum : int {
  kNoTermination  = 0,
  kOptimal        = 1,
  kPrimalInfeasible = 2,
  kDualInfeasible   = 3,
  kIterationLimit = 4,
  kTimeLimit      = 5,
  kNumericalError = 6,
  kPrimalFeasible = 7,
};

// convergence information of one iterate (convergence_information_t)
struct Convergence {
  double primal_objective = 0, dual_objective = 0, gap = 0, abs_objective = 0;
  double l2_primal_residual = 0, l2_dual_residual = 0, l2_x = 0, l2_y = 0;
  double linf_rel_primal_residual = 0, linf_rel_dual_residual = 0;
  double infeasibility[4] = {0, 0, 0, 0};  // max primal ray infeas., primal ray objective, max dual ray infeas., dual ray objective
};

}  // namespace

struct cuoptamd_solver {
  pdlpdev_ctx* dev = nullptr;
  // the set-up's analysis object, minus its workspace, when its release was put off to the solver's end (cuoptamd_solver_create)
  pdlpdev_analysis* spent_analysis = nullptr;
  cuoptamd_hyper H{};
  cuoptamd_settings S{};
  int32_t m_global = 0, n = 0, row_begin = 0, row_end = 0;
  int rank = 0, world = 1;
  bool empty_problem = false;  // n_constraints == 0 -> NumericalError (LP/solve.cu:355-359)
  double objective_scale = 1.0, objective_offset = 0.0;
  double norm_b = 0.0, norm_c = 0.0;
  double computed_step = 0.0, computed_weight = 0.0;  // compute_initial_step_size / _primal_weight, before overrides
  // loop state
  int32_t total_iterations = 0;  // total_pdlp_iterations_ (keeps counting across warm starts)
  int32_t iteration_offset = 0;  // total_pdlp_iterations_ - internal_solver_iterations_
  int32_t attempt_offset   = 0;
  int32_t major_done_at    = -1;
  bool step_error = false, need_aty = true, last_restart_was_average = false;
  double last_candidate_kkt = 0.0, last_restart_kkt = 0.0;
  // the reference leaves gap_reduction_ratio_last_trial_ uninitialised (pdlp_restart_strategy.cu:160);
  // 1.0 is the published algorithm's start value (same choice as the oracle)
  double gap_reduction_ratio_last_trial = 1.0;
  // save_best_primal_so_far: primal_quality_adapter_t of the stored point (convergence_information.hpp:110-124)
  struct Quality {
    bool feasible = false;
    double residual = std::numeric_limits<double>::infinity();
    double objective = 0.0;
    bool operator==(const Quality& o) const { return feasible == o.feasible && residual == o.residual && objective == o.objective; }
  } best_quality;
  bool have_best = false, maximize = false;
  cuoptamd_result best_result{};
  bool have_accepted = false;  // settings.accept_tolerance: first iterate that met the looser set (snapshot = best buffers)
  cuoptamd_result accepted_result{};
  std::string log_path;
  pdlpdev_ctl ctl{};
  Convergence conv_current, conv_average;
  int returned_which = PDLPDEV_CURRENT;
  bool finished = false, warm_started = false;
  cuoptamd_result result{};
  clock_type::time_point solve_start;
  bool started = false;
  // Set-up reordering (pdlpdev_analyze): the device works on P A Q -- row i of its matrix is row row_new2old[i] of the caller's,
  // column j is column col_new2old[j].  Everything that crosses this interface (bounds, initial iterates, solutions, warm-start
  // snapshots) is in the CALLER's order; empty maps = the matrix as given.
  std::vector<int32_t> row_new2old, col_new2old;
  int reorder_method = 0;
  int32_t analysis_info[10] = {0};
  // caller's order -> device order (nullptr stays nullptr; the identity passes the caller's pointer through)
  const double* cols_in(const double* v, std::vector<double>& tmp) const { return to_device_order(col_new2old, v, tmp); }
  const double* rows_in(const double* v, std::vector<double>& tmp) const { return to_device_order(row_new2old, v, tmp); }
  static const double* to_device_order(const std::vector<int32_t>& new2old, const double* v, std::vector<double>& tmp)
  {
    if (!v || new2old.empty()) return v;
    tmp.resize(new2old.size());
    for (size_t i = 0; i < new2old.size(); ++i) tmp[i] = v[new2old[i]];
    return tmp.data();
  }
  // device order -> caller's order for entries [first, last) of the device order (the rows a sharded rank owns; everything for columns)
  static void to_caller_order(const std::vector<int32_t>& new2old, const double* dev, double* user, size_t first, size_t last)
  {
    for (size_t i = first; i < last; ++i) user[new2old[i]] = dev[i];
  }
};

namespace {

// ---- presets: LP/solve.cu:64-199 (defaults = Stable2, LP/pdlp_hyper_params.cu:22-80) ----------
void preset_stable2(cuoptamd_hyper& h)
{
  h.initial_step_size_scaling                                  = 1.0;
  h.ruiz_iterations                                            = 10;
  h.do_pock_chambolle                                          = 1;
  h.do_ruiz                                                    = 1;
  h.alpha_pock_chambolle                                       = 1.0;
  h.artificial_restart_threshold                               = 0.36;
  h.compute_initial_step_size_before_scaling                   = 0;
  h.compute_initial_primal_weight_before_scaling               = 0;
  h.initial_primal_weight_c_scaling                            = 1.0;
  h.initial_primal_weight_b_scaling                            = 1.0;
  h.major_iteration                                            = 40;
  h.min_iteration_restart                                      = 10;
  h.restart_strategy                                           = 1;
  h.never_restart_to_average                                   = 0;
  h.reduction_exponent                                         = 0.3;
  h.growth_exponent                                            = 0.6;
  h.primal_weight_update_smoothing                             = 0.5;
  h.sufficient_reduction_for_restart                           = 0.2;
  h.necessary_reduction_for_restart                            = 0.8;
  h.primal_importance                                          = 1.0;
  h.primal_distance_smoothing                                  = 0.5;
  h.dual_distance_smoothing                                    = 0.5;
  h.compute_last_restart_before_new_primal_weight              = 1;
  h.artificial_restart_in_main_loop                            = 0;
  h.rescale_for_restart                                        = 1;
  h.update_primal_weight_on_initial_solution                   = 0;
  h.update_step_size_on_initial_solution                       = 0;
  h.handle_some_primal_gradients_on_finite_bounds_as_residuals = 0;
  h.project_initial_primal                                     = 1;
}

Convergence to_convergence(const cuoptamd_solver* s, const double* ev)
{
  Convergence c;
  double pobj = ev[PDLPDEV_EV_CX], dobj = ev[PDLPDEV_EV_DUAL_SUM];
  // compute_primal_objective / compute_dual_objective: scale and offset applied only when they are
  // not the identity (convergence_information.cu:261-284, 401-422)
  if (s->objective_scale != 1.0 || s->objective_offset != 0.0) {
    pobj = s->objective_scale * pobj + s->objective_offset;
    dobj = s->objective_scale * dobj + s->objective_offset;
  }
  c.primal_objective         = pobj;
  c.dual_objective           = dobj;
  c.gap                      = std::fabs(pobj - dobj);
  c.abs_objective            = std::fabs(pobj) + std::fabs(dobj);
  c.l2_primal_residual       = std::sqrt(ev[PDLPDEV_EV_PRES2]);
  c.l2_dual_residual         = std::sqrt(ev[PDLPDEV_EV_DRES2]);
  c.l2_x                     = std::sqrt(ev[PDLPDEV_EV_X2]);
  c.l2_y                     = std::sqrt(ev[PDLPDEV_EV_Y2]);
  c.linf_rel_primal_residual = ev[PDLPDEV_EV_LINF_PRES_REL];
  c.linf_rel_dual_residual   = ev[PDLPDEV_EV_LINF_DRES_REL];
  return c;
}

// check_termination_criteria_kernel (termination_strategy.cu:116-250): Optimal, PrimalFeasible, or
// "keep going" (which the reference encodes as NumericalError)
// Optimal by the looser acceptance tolerances (cuoptamd_settings::accept_tolerance)?  Same three inequalities.
bool accepted_by_looser(const cuoptamd_solver* s, const Convergence& c)
{
  const double* a = s->S.accept_tolerance;
  const bool gap_ok = c.gap <= a[0] + a[1] * c.abs_objective;
  bool primal_ok, dual_ok;
  if (s->S.per_constraint_residual) {
    primal_ok = c.linf_rel_primal_residual <= a[2];
    dual_ok   = c.linf_rel_dual_residual <= a[4];
  } else {
    primal_ok = c.l2_primal_residual <= a[2] + a[3] * s->norm_b;
    dual_ok   = c.l2_dual_residual <= a[4] + a[5] * s->norm_c;
  }
  return gap_ok && primal_ok && dual_ok;
}

int verdict(const cuoptamd_solver* s, const Convergence& c)
{
  const cuoptamd_settings& t = s->S;
  const bool gap_ok = c.gap <= t.absolute_gap_tolerance + t.relative_gap_tolerance * c.abs_objective;
  bool primal_ok, dual_ok;
  if (t.per_constraint_residual) {
    primal_ok = c.linf_rel_primal_residual <= t.absolute_primal_tolerance;
    dual_ok   = c.linf_rel_dual_residual <= t.absolute_dual_tolerance;
  } else {
    primal_ok = c.l2_primal_residual <= t.absolute_primal_tolerance + t.relative_primal_tolerance * s->norm_b;
    dual_ok   = c.l2_dual_residual <= t.absolute_dual_tolerance + t.relative_dual_tolerance * s->norm_c;
  }
  if (dual_ok && primal_ok && gap_ok) return kOptimal;
  if (primal_ok) {
    // (see cuoptamd_settings::unbounded_from_feasible_iterates: not the reference's PDLP, which stops looking here)
    if (t.detect_infeasibility && t.unbounded_from_feasible_iterates && !t.first_primal_feasible) {
      const double* f = c.infeasibility;
      if (f[1] < 0.0 && f[0] / -f[1] <= t.dual_infeasible_tolerance) return kDualInfeasible;
    }
    return kPrimalFeasible;
  }
  if (t.detect_infeasibility) {  // termination_strategy.cu:228-249
    const double* f = c.infeasibility;
    if (f[3] > 0.0 && f[2] / f[3] <= t.primal_infeasible_tolerance) return kPrimalInfeasible;
    if (f[1] < 0.0 && f[0] / -f[1] <= t.dual_infeasible_tolerance) return kDualInfeasible;
  }
  return kNumericalError;
}

// kernel_compute_kkt_score (pdlp_restart_strategy.cu:366-390)
double kkt_score(const Convergence& c, double w)
{
  const double w2 = w * w;
  return std::sqrt(w2 * c.l2_primal_residual * c.l2_primal_residual +
                   c.l2_dual_residual * c.l2_dual_residual / w2 + c.gap * c.gap);
}

void fill_result(cuoptamd_solver* s, int status, int which)
{
  const Convergence& c     = which == PDLPDEV_AVERAGE ? s->conv_average : s->conv_current;
  cuoptamd_result& r       = s->result;
  r.status                 = status;
  r.returned_average       = which == PDLPDEV_AVERAGE;
  r.steps_taken            = s->ctl.steps_taken;
  r.attempted_steps        = s->ctl.attempts;
  r.primal_objective       = c.primal_objective;
  r.dual_objective         = c.dual_objective;
  r.gap                    = c.gap;
  r.relative_gap           = c.gap / (1.0 + c.abs_objective);  // convergence_information.cu:474-492
  r.l2_primal_residual     = c.l2_primal_residual;
  r.l2_dual_residual       = c.l2_dual_residual;
  r.l2_relative_primal_residual = c.l2_primal_residual / (1.0 + s->norm_b);
  r.l2_relative_dual_residual   = c.l2_dual_residual / (1.0 + s->norm_c);
  r.step_size              = s->ctl.step_size;
  r.primal_weight          = s->ctl.primal_weight;
  r.max_primal_ray_infeasibility = c.infeasibility[0];
  r.primal_ray_linear_objective  = c.infeasibility[1];
  r.max_dual_ray_infeasibility   = c.infeasibility[2];
  r.dual_ray_linear_objective    = c.infeasibility[3];
  s->returned_which        = which;
}


void log_line(const cuoptamd_solver* s, const char* fmt, ...)
{
  if (!s->S.log_to_console && s->log_path.empty()) return;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s->S.log_to_console) {
    std::fputs(buf, stdout);
    std::fflush(stdout);
  }
  if (!s->log_path.empty()) {
    if (FILE* f = std::fopen(s->log_path.c_str(), "a")) {
      std::fputs(buf, f);
      std::fclose(f);
    }
  }
}
// one row of the iteration table (print_termination_criteria, termination_strategy.cu:380-390)
void log_iteration(const cuoptamd_solver* s, const Convergence& c)
{
  log_line(s, "%7d %+.8e %+.8e  %8.2e   %8.2e     %8.2e   %.3fs\n", s->total_iterations, c.primal_objective,
           c.dual_objective, c.gap, c.l2_primal_residual, c.l2_dual_residual, seconds_since(s->solve_start));
}

// get_best_quality (pdlp.cu:344-378): feasibility first, then objective (sense aware), else least residual
const cuoptamd_solver::Quality& best_of(const cuoptamd_solver* s, const cuoptamd_solver::Quality& a,
                                        const cuoptamd_solver::Quality& b)
{
  if (a.feasible && !b.feasible) return a;
  if (!a.feasible && b.feasible) return b;
  if (a.feasible && b.feasible) {
    const bool a_lower = a.objective < b.objective;
    return ((!s->maximize && a_lower) || (s->maximize && !a_lower)) ? a : b;
  }
  return a.residual < b.residual ? a : b;
}

// One major iteration: averages, the two convergence evaluations, termination, limits, restart.
// Sets *terminated when a solution must be returned.
struct HostRange {  // roctx range of a host-side phase (LP/pdlp.cu:541,1227 are NVTX ranges in the reference)
  explicit HostRange(const char* name) { pdlpdev_range_push(name); }
  ~HostRange() { pdlpdev_range_pop(); }
};

// The head of a major iteration asks the device for: pending average + average iterate + both convergence evaluations
// (pdlpdev_major_eval; one launch for K LPs of a small-LP batch) ...
pdlpdev_small_eval major_eval_request(const cuoptamd_solver* s)
{
  // pdlp.cu:1110-1122: with 0 or 1 steps the average IS the iterate (avoids a*x/x != x);
  // right after a restart the sums are empty and the reference yields zeros.
  int mode = 2;
  if (s->total_iterations - s->iteration_offset <= 1 && !s->warm_started)  // internal_solver_iterations_ <= 1
    mode = 0;
  else if (s->ctl.its_since_restart == 0)
    mode = 1;
  pdlpdev_small_eval r;
  r.mode        = mode;
  r.rule_finite = s->H.handle_some_primal_gradients_on_finite_bounds_as_residuals == 0;
  // the l-infinity residuals are only consumed by the per-constraint verdict (termination_strategy.cu:189-205)
  r.eps_p = s->S.per_constraint_residual ? s->S.relative_primal_tolerance : -1.0;
  r.eps_d = s->S.per_constraint_residual ? s->S.relative_dual_tolerance : -1.0;
  return r;
}
// ... and, when the KKT rule decides to restart, for pdlpdev_restart(which, unscaled) -- whose distances then give the new primal
// weight (major_restart_done).  The plan is what the head hands back to whoever talks to the device.
struct MajorPlan {
  bool restart = false;
  int which = PDLPDEV_CURRENT, unscaled = 0;
  bool really_average = false;
  double candidate = 0.0;
};
int major_head(cuoptamd_solver* s, const double* ev, const double* ev_avg, bool* terminated, MajorPlan* plan);
double major_restart_done(cuoptamd_solver* s, const MajorPlan& plan, const double dist2[2]);

int major_iteration(cuoptamd_solver* s, bool* terminated)
{
  HostRange range("pdlp: major iteration (termination + restart logic)");
  pdlpdev_ctx* dev = s->dev;
  const pdlpdev_small_eval rq = major_eval_request(s);
  double ev[PDLPDEV_EV_COUNT], ev_avg[PDLPDEV_EV_COUNT];
  DEV(pdlpdev_major_eval(dev, rq.mode, rq.rule_finite, rq.eps_p, rq.eps_d, ev, ev_avg));
  MajorPlan plan;
  int rc = major_head(s, ev, ev_avg, terminated, &plan);
  if (rc != 0 || *terminated || !plan.restart) return rc;
  double dist2[2];
  DEV(pdlpdev_restart(dev, plan.which, plan.unscaled, dist2));
  const double nw = major_restart_done(s, plan, dist2);
  if (nw > 0.0) DEV(pdlpdev_set_step(dev, -1.0, nw));
  return 0;
}

// compute_new_primal_weight (pdlp_restart_strategy.cu:684-750, safe guard pdlp_constants.hpp:34-35) + the bookkeeping behind a KKT
// restart; returns the new primal weight (<= 0: unchanged) for pdlpdev_set_step
double major_restart_done(cuoptamd_solver* s, const MajorPlan& plan, const double dist2[2])
{
  const double w = s->ctl.primal_weight;
  double nw      = -1.0;
  s->last_restart_was_average = plan.really_average;
  if (plan.really_average) s->need_aty = true;  // pdhg.cu:183-184
  s->ctl.its_since_restart = 0;
  s->ctl.sum_weights       = 0.0;
  const double pd = std::sqrt(dist2[0]), dd = std::sqrt(dist2[1]);
  const double guard = 1.0e-10;
  if (!(pd < guard || pd >= 1.0 / guard || dd < guard || dd >= 1.0 / guard)) {
    const double theta = s->H.primal_weight_update_smoothing;
    nw                 = std::exp(theta * std::log(dd / pd) + (1.0 - theta) * std::log(w));
    s->ctl.primal_weight = nw;
  }
  s->last_restart_kkt = plan.candidate;
  return nw;
}

int major_head(cuoptamd_solver* s, const double* ev, const double* ev_avg, bool* terminated, MajorPlan* plan)
{
  const cuoptamd_hyper& H = s->H;
  pdlpdev_ctx* dev        = s->dev;
  *terminated             = false;
  *plan                   = MajorPlan();
  s->result.num_major_iterations += 1;
  const int rule_finite = H.handle_some_primal_gradients_on_finite_bounds_as_residuals == 0;
  s->conv_current = to_convergence(s, ev);
  s->conv_average = to_convergence(s, ev_avg);
  if (s->S.detect_infeasibility) {
    DEV(pdlpdev_eval_infeasibility(dev, PDLPDEV_CURRENT, rule_finite, s->conv_current.infeasibility));
    DEV(pdlpdev_eval_infeasibility(dev, PDLPDEV_AVERAGE, rule_finite, s->conv_average.infeasibility));
  }
  const int t_cur = verdict(s, s->conv_current), t_avg = verdict(s, s->conv_average);
  const double w  = s->ctl.primal_weight;

  // ---- check_termination (pdlp.cu:537-802) ----
  bool done = false;
  int status = kNumericalError, which = PDLPDEV_CURRENT;
  if (s->total_iterations > 1) {  // :580-583: only limits during the first two iterations
    if (s->S.first_primal_feasible) {  // :587-633
      if (t_avg == kPrimalFeasible && t_cur == kPrimalFeasible) {
        done = true, status = kPrimalFeasible;
        which = s->conv_current.l2_primal_residual < s->conv_average.l2_primal_residual ? PDLPDEV_CURRENT : PDLPDEV_AVERAGE;
      } else if (t_cur == kPrimalFeasible) {
        done = true, status = kPrimalFeasible, which = PDLPDEV_CURRENT;
      } else if (t_avg == kPrimalFeasible) {
        done = true, status = kPrimalFeasible, which = PDLPDEV_AVERAGE;
      }
    }
    if (!done && t_avg == kOptimal && t_cur == kOptimal) {  // :636-682, ties -> average
      done = true, status = kOptimal;
      which = kkt_score(s->conv_current, w) < kkt_score(s->conv_average, w) ? PDLPDEV_CURRENT : PDLPDEV_AVERAGE;
    }
    if (!done && t_avg == kOptimal) done = true, status = kOptimal, which = PDLPDEV_AVERAGE;   // :685-700
    if (!done && t_cur == kOptimal) done = true, status = kOptimal, which = PDLPDEV_CURRENT;   // :701-716
    if (!done && s->S.detect_infeasibility) {  // :718-776: strict -> one iterate suffices, else both must agree
      const bool cur_inf = t_cur == kPrimalInfeasible || t_cur == kDualInfeasible;
      const bool avg_inf = t_avg == kPrimalInfeasible || t_avg == kDualInfeasible;
      if (s->S.strict_infeasibility) {
        if (cur_inf)
          done = true, status = t_cur, which = PDLPDEV_CURRENT;
        else if (avg_inf)
          done = true, status = t_avg, which = PDLPDEV_AVERAGE;
      } else if (cur_inf && t_cur == t_avg) {
        done = true, status = t_cur, which = PDLPDEV_CURRENT;
      }
    }
    if (!done && s->step_error) {  // :780-789: numerical error, empty solution
      fill_result(s, kNumericalError, PDLPDEV_CURRENT);
      *terminated = true;
      return 0;
    }
  }
  if (!done && s->S.save_best_primal_so_far) {  // record_best_primal_so_far, pdlp.cu:390-466
    const cuoptamd_solver::Quality qc{t_cur == kPrimalFeasible, s->conv_current.l2_primal_residual, s->conv_current.primal_objective};
    const cuoptamd_solver::Quality qa{t_avg == kPrimalFeasible, s->conv_average.l2_primal_residual, s->conv_average.primal_objective};
    const cuoptamd_solver::Quality& cand = best_of(s, qc, qa);
    const cuoptamd_solver::Quality& over = best_of(s, cand, s->best_quality);
    if (!(over == s->best_quality) || !s->have_best) {
      const int bw = (&cand == &qc) ? PDLPDEV_CURRENT : PDLPDEV_AVERAGE;
      s->best_quality = over;
      DEV(pdlpdev_save_best(dev, bw));
      fill_result(s, kTimeLimit, bw);
      s->best_result = s->result;
      s->have_best   = true;
    }
  }
  if (!done && s->S.accept_enabled && !s->S.save_best_primal_so_far && !s->have_accepted && s->total_iterations > 1) {
    const bool ac = accepted_by_looser(s, s->conv_current), aa = accepted_by_looser(s, s->conv_average);
    if (ac || aa) {
      const int aw = (ac && aa) ? (kkt_score(s->conv_current, w) < kkt_score(s->conv_average, w) ? PDLPDEV_CURRENT : PDLPDEV_AVERAGE)
                                : (aa ? PDLPDEV_AVERAGE : PDLPDEV_CURRENT);
      DEV(pdlpdev_save_best(dev, aw));
      fill_result(s, kOptimal, aw);
      s->accepted_result = s->result;
      s->have_accepted   = true;
    }
  }
  if (s->total_iterations % 1000 == 0) log_iteration(s, s->conv_current);  // pdlp.cu:798
  if (!done) {  // check_limits, pdlp.cu:264-331 (time first, then iterations)
    const double tl = s->S.time_limit;
    double elapsed  = std::isfinite(tl) ? seconds_since(s->solve_start) : 0.0;
    if (std::isfinite(tl) && s->world > 1) DEV(pdlpdev_agree_max(dev, &elapsed));  // every rank must stop together
    if (std::isfinite(tl) && elapsed * 1000.0 >= tl * 1000.0)
      done = true, status = kTimeLimit, which = PDLPDEV_CURRENT;
    else if (s->total_iterations - s->iteration_offset >= s->S.iteration_limit)  // internal_solver_iterations_
      done = true, status = kIterationLimit, which = PDLPDEV_CURRENT;
  }
  if (done) {
    if ((status == kTimeLimit || status == kIterationLimit) && s->S.save_best_primal_so_far && s->have_best) {
      const int nr = s->result.num_restarts, nm = s->result.num_major_iterations;
      s->result                      = s->best_result;  // the stored point and its statistics
      s->result.status               = status;
      s->result.num_restarts         = nr;
      s->result.num_major_iterations = nm;
      s->returned_which              = PDLPDEV_BEST;
    } else if ((status == kTimeLimit || status == kIterationLimit) && s->have_accepted) {
      const int nr = s->result.num_restarts, nm = s->result.num_major_iterations;
      const int steps = s->ctl.steps_taken, attempts = s->ctl.attempts;
      s->result                      = s->accepted_result;  // Optimal at the tolerances the caller asked for
      s->result.num_restarts         = nr;
      s->result.num_major_iterations = nm;
      s->result.steps_taken          = steps;  // the work that was done, not the moment of acceptance
      s->result.attempted_steps      = attempts;
      s->result.accepted_at_looser_tolerances = 1;
      s->returned_which              = PDLPDEV_BEST;
    } else {
      fill_result(s, status, which);
    }
    *terminated = true;
    return 0;
  }

  // ---- run_kkt_restart (pdlp_restart_strategy.cu:467-641) ----
  if (H.restart_strategy == 1) {
    const double cur_score = kkt_score(s->conv_current, w);
    if (s->ctl.its_since_restart == 0) {  // :507-514
      s->last_candidate_kkt = cur_score;
      s->last_restart_kkt   = cur_score;
    } else {
      const double avg_score = kkt_score(s->conv_average, w);
      const bool to_average  = !(cur_score < avg_score);  // ties go to the average, :524-530
      const double candidate = to_average ? avg_score : cur_score;
      // kkt_restart_conditions :431-437 = artificial (:939-961) || kkt_decay (:407-429)
      bool restart = s->ctl.its_since_restart >= H.artificial_restart_threshold * s->total_iterations;
      if (!restart) {
        if (candidate < H.sufficient_reduction_for_restart * s->last_restart_kkt)
          restart = true;
        else if (candidate < H.necessary_reduction_for_restart * s->last_restart_kkt && candidate > s->last_candidate_kkt)
          restart = true;
      }
      if (restart) {  // the device part (pdlpdev_restart, then the new weight) is the caller's: major_iteration / the small-LP batch
        s->result.num_restarts += 1;
        plan->restart        = true;
        plan->really_average = to_average && !H.never_restart_to_average;
        plan->which          = plan->really_average ? PDLPDEV_AVERAGE : PDLPDEV_CURRENT;
        plan->unscaled       = !H.rescale_for_restart;
        plan->candidate      = candidate;
      }
      s->last_candidate_kkt = candidate;
    }
  }
  // ---- run_trust_region_restart (pdlp_restart_strategy.cu:277-364), Methodical1 ----
  if (H.restart_strategy == 2 && s->ctl.its_since_restart != 0) {
    const double tau = s->ctl.tau, sigma = s->ctl.sigma;
    const double wp = tau == 0.0 ? 0.0 : 1.0 / tau, wd = sigma == 0.0 ? 0.0 : 1.0 / sigma;  // norm weights :301-310
    const double pds = H.primal_distance_smoothing, dds = H.dual_distance_smoothing;
    bool restart = s->ctl.its_since_restart >= H.artificial_restart_threshold * s->total_iterations;
    // compute_localized_duality_gaps :982-1030 (A x / A^T y of both iterates are those of the evaluations above)
    double avg[6], cur[6];
    // rescale_for_restart: scaled iterates meet the unscaled problem the reference's restart strategy is built on (pdlp.cu:99-103)
    const int scaled = H.rescale_for_restart ? 1 : 0;
    DEV(pdlpdev_trust_region_bounds(dev, PDLPDEV_AVERAGE, wp, wd, pds, dds, w, -1.0, scaled, avg));
    DEV(pdlpdev_trust_region_bounds(dev, PDLPDEV_CURRENT, wp, wd, pds, dds, w, -1.0, scaled, cur));
    const double ng_avg = (avg[5] - avg[4]) / avg[2], ng_cur = (cur[5] - cur[4]) / cur[2];
    const bool to_average = ng_cur / cur[2] >= ng_avg / avg[2];  // pick_restart_candidate_kernel :841-856
    const double* cand    = to_average ? avg : cur;
    const double ng_cand  = to_average ? ng_avg : ng_cur;
    if (!restart) {  // should_do_adaptive_restart_normalized_duality_gap :905-937
      double ev_lr[PDLPDEV_EV_COUNT], lr[6];
      if (!scaled) DEV(pdlpdev_eval(dev, PDLPDEV_LAST_RESTART, rule_finite, -1.0, -1.0, ev_lr));
      DEV(pdlpdev_trust_region_bounds(dev, PDLPDEV_LAST_RESTART, wp, wd, pds, dds, w, cand[2], scaled, lr));
      const double ng_lr = (lr[5] - lr[4]) / cand[2];
      const double ratio = ng_cand / ng_lr;  // adaptive_restart_triggered :876-903
      if (ratio < H.necessary_reduction_for_restart &&
          (ratio < H.sufficient_reduction_for_restart || ratio > s->gap_reduction_ratio_last_trial))
        restart = true;
      s->gap_reduction_ratio_last_trial = ratio;
    }
    if (restart) {
      s->result.num_restarts += 1;
      const bool really_average = to_average && !H.never_restart_to_average;
      double dist2[2];
      DEV(pdlpdev_restart(dev, really_average ? PDLPDEV_AVERAGE : PDLPDEV_CURRENT, scaled ? 0 : 1, dist2));
      // the reference measures the distances of the PICKED candidate even when never_restart_to_average
      // redirects the copy; only Fast1 sets that flag and it uses the KKT restart
      s->last_restart_was_average = really_average;
      if (really_average) s->need_aty = true;
      s->ctl.its_since_restart = 0;
      s->ctl.sum_weights       = 0.0;
      const double pd = std::sqrt(cand[0]), dd = std::sqrt(cand[1]);
      const double guard = 1.0e-10;
      if (!(pd < guard || pd >= 1.0 / guard || dd < guard || dd >= 1.0 / guard)) {
        const double theta = H.primal_weight_update_smoothing;
        const double nw    = std::exp(theta * std::log(dd / pd) + (1.0 - theta) * std::log(w));
        DEV(pdlpdev_set_step(dev, -1.0, nw));
        s->ctl.primal_weight = nw;
      }
    }
  }
  return 0;
}

}  // namespace

extern "C" {

const char* cuoptamd_last_error(void) { return g_error.c_str(); }

void cuoptamd_hyper_preset(int mode, cuoptamd_hyper* h)
{
  preset_stable2(*h);
  if (mode == 0) {  // Stable1, LP/solve.cu:66-96
    h->initial_step_size_scaling                                  = 1.6;
    h->ruiz_iterations                                            = 1;
    h->alpha_pock_chambolle                                       = 1.3;
    h->artificial_restart_threshold                               = 0.5;
    h->compute_initial_primal_weight_before_scaling               = 1;
    h->initial_primal_weight_c_scaling                            = 2.2;
    h->initial_primal_weight_b_scaling                            = 4.6;
    h->major_iteration                                            = 52;
    h->min_iteration_restart                                      = 0;
    h->reduction_exponent                                         = 0.5;
    h->growth_exponent                                            = 0.9;
    h->primal_weight_update_smoothing                             = 0.3;
    h->necessary_reduction_for_restart                            = 0.5;
    h->primal_importance                                          = 1.8;
    h->primal_distance_smoothing                                  = 0.6;
    h->dual_distance_smoothing                                    = 0.2;
    h->compute_last_restart_before_new_primal_weight              = 0;
    h->rescale_for_restart                                        = 0;
    h->handle_some_primal_gradients_on_finite_bounds_as_residuals = 1;
    h->project_initial_primal                                     = 0;
  } else if (mode == 2) {  // Methodical1, LP/solve.cu:133-163
    h->ruiz_iterations                                            = 5;
    h->artificial_restart_threshold                               = 0.5;
    h->major_iteration                                            = 64;
    h->min_iteration_restart                                      = 0;
    h->restart_strategy                                           = 2;
    h->sufficient_reduction_for_restart                           = 0.1;
    h->necessary_reduction_for_restart                            = 0.9;
    h->rescale_for_restart                                        = 0;
    h->handle_some_primal_gradients_on_finite_bounds_as_residuals = 1;
    h->project_initial_primal                                     = 0;
  } else if (mode == 3) {  // Fast1, LP/solve.cu:167-197
    h->initial_step_size_scaling                                  = 0.8;
    h->ruiz_iterations                                            = 6;
    h->do_ruiz                                                    = 0;
    h->alpha_pock_chambolle                                       = 2.0;
    h->artificial_restart_threshold                               = 0.3;
    h->compute_initial_primal_weight_before_scaling               = 1;
    h->initial_primal_weight_c_scaling                            = 1.2;
    h->initial_primal_weight_b_scaling                            = 1.2;
    h->major_iteration                                            = 76;
    h->min_iteration_restart                                      = 6;
    h->never_restart_to_average                                   = 1;
    h->reduction_exponent                                         = 0.4;
    h->sufficient_reduction_for_restart                           = 0.3;
    h->necessary_reduction_for_restart                            = 0.9;
    h->primal_importance                                          = 0.8;
    h->primal_distance_smoothing                                  = 0.8;
    h->dual_distance_smoothing                                    = 0.3;
    h->artificial_restart_in_main_loop                            = 1;
    h->handle_some_primal_gradients_on_finite_bounds_as_residuals = 1;
    h->project_initial_primal                                     = 0;
  }
}

void cuoptamd_default_settings(cuoptamd_settings* s)
{
  // tolerances_t, pdlp/solver_settings.hpp:179-188
  s->absolute_gap_tolerance = s->relative_gap_tolerance = 1e-4;
  s->absolute_primal_tolerance = s->relative_primal_tolerance = 1e-4;
  s->absolute_dual_tolerance = s->relative_dual_tolerance = 1e-4;
  s->iteration_limit         = std::numeric_limits<int32_t>::max();
  s->time_limit              = std::numeric_limits<double>::infinity();
  s->per_constraint_residual = 0;
  s->first_primal_feasible   = 0;
  s->initial_step_size       = -1.0;
  s->initial_primal_weight   = -1.0;
  s->initial_k               = -1;
  s->use_graph               = 1;
  s->detect_infeasibility    = 0;
  s->strict_infeasibility    = 0;
  s->primal_infeasible_tolerance = 1e-8;
  s->dual_infeasible_tolerance   = 1e-8;
  s->save_best_primal_so_far     = 0;
  s->log_to_console              = 0;
  s->log_file                    = nullptr;
  s->unbounded_from_feasible_iterates = 0;
  s->accept_enabled                   = 0;
  for (double& t : s->accept_tolerance) t = 1e-4;
  s->relative_primal_tolerance_factor = s->relative_dual_tolerance_factor = -1.0;
}

// Plain parallel counting sort by column (small matrices, and the fallback of the blocked one below).  Thread t owns a
// contiguous chunk of rows; because the chunks are ordered and every thread scans its rows in order, the row indices come out
// ascending inside every column -- what cusparseCsr2cscEx2 produces for the reference (problem.cu:277-309) -- for any thread count.
static void transpose_direct(int32_t m, int32_t n, const int32_t* offsets, const int32_t* indices, const double* values,
                             int32_t* t_offsets, int32_t* t_indices, double* t_values, int T)
{
  const int64_t nnz = offsets[m];
  std::vector<int32_t> row_cut(T + 1);
  for (int t = 0; t <= T; ++t) {  // chunks balanced by nonzeros
    const int64_t want = nnz * t / T;
    row_cut[t]         = (int32_t)(std::lower_bound(offsets, offsets + m + 1, (int32_t)want) - offsets);
  }
  row_cut[0] = 0, row_cut[T] = m;
  // hist[t][j] = entries of column j inside chunk t, later turned into chunk t's write cursor for column j
  std::vector<std::vector<int32_t>> hist(T);
  cuopt_amd::parallel_tasks(T, [&](int t) {
    hist[t].assign((size_t)n, 0);
    int32_t* h = hist[t].data();
    for (int64_t k = offsets[row_cut[t]]; k < offsets[row_cut[t + 1]]; ++k) h[indices[k]] += 1;
  });
  // column totals -> offsets; per (column, chunk) start positions
  const int CT = T;  // columns are split over the same number of workers for the prefix pass
  std::vector<int64_t> col_block_sum(CT + 1, 0);
  std::vector<int32_t> col_cut(CT + 1);
  for (int t = 0; t <= CT; ++t) col_cut[t] = (int32_t)((int64_t)n * t / CT);
  cuopt_amd::parallel_tasks(CT, [&](int b) {
    int64_t s = 0;
    for (int32_t j = col_cut[b]; j < col_cut[b + 1]; ++j)
      for (int t = 0; t < T; ++t) s += hist[t][j];
    col_block_sum[b + 1] = s;
  });
  for (int b = 0; b < CT; ++b) col_block_sum[b + 1] += col_block_sum[b];
  cuopt_amd::parallel_tasks(CT, [&](int b) {
    int64_t pos = col_block_sum[b];
    for (int32_t j = col_cut[b]; j < col_cut[b + 1]; ++j) {
      t_offsets[j] = (int32_t)pos;
      for (int t = 0; t < T; ++t) {
        const int32_t c = hist[t][j];
        hist[t][j]      = (int32_t)pos;
        pos += c;
      }
    }
  });
  t_offsets[n] = (int32_t)nnz;
  cuopt_amd::parallel_tasks(T, [&](int t) {
    int32_t* cursor = hist[t].data();
    for (int32_t i = row_cut[t]; i < row_cut[t + 1]; ++i)
      for (int32_t k = offsets[i]; k < offsets[i + 1]; ++k) {
        const int32_t q = cursor[indices[k]]++;
        t_indices[q]    = i;
        t_values[q]     = values[k];
      }
  });
}

void cuoptamd_csr_transpose(int32_t m, int32_t n, const int32_t* offsets, const int32_t* indices,
                            const double* values, int32_t* t_offsets, int32_t* t_indices,
                            double* t_values)
{
  const int64_t nnz = offsets[m];
  const int T       = nnz < (1 << 18) ? 1 : cuopt_amd::host_threads();
  // Large matrices: the direct scatter writes every entry to a random place of a 120 MB array (a cache line read and written
  // per 12 bytes).  Blocked instead: (1) each thread deals its rows' entries into per-(thread, column block) buckets -- 256
  // sequential write streams; (2) each column block is sorted by column from the buckets of all threads IN THREAD ORDER (rows
  // stay ascending inside a column: the same output as the direct version, for any thread count) into its own contiguous piece
  // of the result, which the cache holds.
  constexpr int kBlocks = 256;
  if (T == 1 || nnz < (1 << 22) || n < 16 * kBlocks || cuopt_amd::tune_int("transpose_direct", 0)) {
    transpose_direct(m, n, offsets, indices, values, t_offsets, t_indices, t_values, T);
    return;
  }
  const int32_t bw = (n + kBlocks - 1) / kBlocks;  // columns per block
  std::vector<int32_t> row_cut(T + 1);
  for (int t = 0; t <= T; ++t) {
    const int64_t want = nnz * t / T;
    row_cut[t]         = (int32_t)(std::lower_bound(offsets, offsets + m + 1, (int32_t)want) - offsets);
  }
  row_cut[0] = 0, row_cut[T] = m;
  // bucket sizes: cnt[t][b]
  std::vector<std::vector<int32_t>> cnt(T, std::vector<int32_t>(kBlocks + 1, 0));
  cuopt_amd::parallel_tasks(T, [&](int t) {
    int32_t* c = cnt[t].data();
    for (int64_t k = offsets[row_cut[t]]; k < offsets[row_cut[t + 1]]; ++k) c[indices[k] / bw + 1] += 1;
    for (int b = 0; b < kBlocks; ++b) c[b + 1] += c[b];  // -> the thread's bucket starts
  });
  struct Item {
    int32_t col, row;
    double val;
  };
  std::vector<cuopt_amd::PoolArray<Item>> bucket(T);
  cuopt_amd::parallel_tasks(T, [&](int t) {
    const int64_t mine = (int64_t)offsets[row_cut[t + 1]] - offsets[row_cut[t]];
    bucket[t].reset((size_t)std::max<int64_t>(mine, 1));
    Item* out = bucket[t].get();
    std::vector<int32_t> at(cnt[t].begin(), cnt[t].end() - 1);
    for (int32_t i = row_cut[t]; i < row_cut[t + 1]; ++i)
      for (int32_t k = offsets[i]; k < offsets[i + 1]; ++k) out[at[indices[k] / bw]++] = Item{indices[k], i, values[k]};
  });
  // where each column block starts in the result
  std::vector<int64_t> block_start(kBlocks + 1, 0);
  for (int b = 0; b < kBlocks; ++b) {
    int64_t s = 0;
    for (int t = 0; t < T; ++t) s += cnt[t][b + 1] - cnt[t][b];
    block_start[b + 1] = block_start[b] + s;
  }
  cuopt_amd::parallel_tasks(kBlocks, [&](int b) {
    const int32_t j0 = (int32_t)std::min<int64_t>((int64_t)b * bw, n), j1 = (int32_t)std::min<int64_t>((int64_t)(b + 1) * bw, n);
    std::vector<int32_t> cur((size_t)(j1 - j0) + 1, 0);
    for (int t = 0; t < T; ++t) {
      const Item* it = bucket[t].get();
      for (int32_t e = cnt[t][b]; e < cnt[t][b + 1]; ++e) cur[it[e].col - j0 + 1] += 1;
    }
    int64_t pos = block_start[b];
    for (int32_t j = j0; j < j1; ++j) {
      const int32_t c = cur[j - j0 + 1];
      t_offsets[j]    = (int32_t)pos;
      cur[j - j0]     = (int32_t)pos;  // (cursor of column j; slot j - j0 + 1 is read before it is overwritten in the next round)
      pos += c;
    }
    for (int t = 0; t < T; ++t) {
      const Item* it = bucket[t].get();
      for (int32_t e = cnt[t][b]; e < cnt[t][b + 1]; ++e) {
        const int32_t q = cur[it[e].col - j0]++;
        t_indices[q]    = it[e].row;
        t_values[q]     = it[e].val;
      }
    }
  });
  t_offsets[n] = (int32_t)nnz;
}

void cuoptamd_partition_rows(int32_t m, const int32_t* offsets, int world, int32_t* bounds)
{
  // contiguous blocks, boundary g at the first row whose prefix nnz reaches g/world of the total
  const int64_t nnz = offsets[m];
  bounds[0] = 0;
  for (int g = 1; g < world; ++g) {
    const int64_t want = (nnz * g) / world;
    const int32_t* it  = std::lower_bound(offsets, offsets + m + 1, (int32_t)want);
    int32_t row        = (int32_t)(it - offsets);
    row                = std::max(row, bounds[g - 1]);
    bounds[g]          = std::min(row, m);
  }
  bounds[world] = m;
}

// tail of the [init] block of run_solver (pdlp.cu:1014-1056): overrides, step/weight/k on the device, optional
// initial iterate, projection.  Shared by cuoptamd_solver_create and cuoptamd_solver_reset.
static int start_run(cuoptamd_solver* s, const double* init_x, const double* init_y)
{
  const cuoptamd_hyper* hyper = &s->H;
  const cuoptamd_settings* settings = &s->S;
  // set_relative_{primal,dual}_tolerance_factor (pdlp.cu:209-231): the caller's ||b|| / ||c|| for the termination rule
  if (settings->relative_primal_tolerance_factor >= 0.0) s->norm_b = settings->relative_primal_tolerance_factor;
  if (settings->relative_dual_tolerance_factor >= 0.0) s->norm_c = settings->relative_dual_tolerance_factor;
  double step = s->computed_step, weight = s->computed_weight;
  if (settings->initial_step_size >= 0.0) step = settings->initial_step_size;  // pdlp.cu:1014-1021
  if (settings->initial_primal_weight >= 0.0) weight = settings->initial_primal_weight;
  s->result.initial_step_size     = step;
  s->result.initial_primal_weight = weight;
  DEV(pdlpdev_set_step(s->dev, step, weight));
  if (settings->initial_k >= 0) DEV(pdlpdev_set_k(s->dev, settings->initial_k));
  std::vector<double> init_x_tmp, init_y_tmp;
  init_x = s->cols_in(init_x, init_x_tmp), init_y = s->rows_in(init_y, init_y_tmp);
  if (init_x || init_y) {
    DEV(pdlpdev_set_initial(s->dev, init_x, init_y ? init_y + s->row_begin : nullptr));
    // update_{step_size,primal_weight}_on_initial_solution (pdlp.cu:878-979; off in every preset, toggled by the
    // reference's initial_solution_test, pdlp_test.cu:245-523: no change without BOTH iterates or with an all-zero one)
    if (hyper->update_step_size_on_initial_solution && init_x && init_y) {
      // compute_initial_step_size_before_scaling: the caller's vectors as they came, against the scaled matrix (pdlp.cu:929-947)
      const bool before = hyper->compute_initial_step_size_before_scaling != 0;
      double st[5];
      DEV(pdlpdev_initial_solution_stats(s->dev, st, before ? init_x : nullptr, before ? init_y + s->row_begin : nullptr));
      if (st[3] != 0.0 && st[4] != 0.0) {
        // one compute_step_sizes with delta = the initial iterate and A^T y = 0 (adaptive_step_size_strategy.cu:91-188);
        // the kernel's own increment of the device iteration counter is not undone
        const double movement = hyper->primal_distance_smoothing * weight * st[1] + (hyper->dual_distance_smoothing / weight) * st[2];
        if (movement > 0.0 && movement < 1.0e100) {
          const int k_after  = (settings->initial_k >= 0 ? settings->initial_k : 0) + 1;
          const double coef = (double)k_after, inter = std::fabs(st[0]);
          const double limit = inter > 0.0 ? movement / inter : std::numeric_limits<double>::infinity();
          const double s1 = (1.0 - std::pow(coef + 1.0, -hyper->reduction_exponent)) * limit;
          const double s2 = (1.0 + std::pow(coef + 1.0, -hyper->growth_exponent)) * step;
          step = std::min(s1, s2);
          DEV(pdlpdev_set_step(s->dev, step, weight));
          DEV(pdlpdev_set_k(s->dev, k_after));
        } else {
          s->step_error = true;  // valid_step_size = -1: the first loop trip is a major iteration that reports it
        }
      }
      if (!before) s->need_aty = false;  // A^T y0 (scaled) was just computed
    }
    if (hyper->update_primal_weight_on_initial_solution) {
      // update_distance (pdlp_restart_strategy.cu:440-465): distance to the (zero) anchors, anchors <- iterate, new weight;
      // on the unscaled iterate when the initial weight is computed before scaling (pdlp.cu:950-979)
      double dist2[2];
      DEV(pdlpdev_restart(s->dev, PDLPDEV_CURRENT, hyper->compute_initial_primal_weight_before_scaling ? 1 : 0, dist2));
      const double pd = std::sqrt(dist2[0]), dd = std::sqrt(dist2[1]);
      const double guard = 1.0e-10;
      if (!(pd < guard || pd >= 1.0 / guard || dd < guard || dd >= 1.0 / guard)) {
        const double theta = hyper->primal_weight_update_smoothing;
        weight = std::exp(theta * std::log(dd / pd) + (1.0 - theta) * std::log(weight));
        DEV(pdlpdev_set_step(s->dev, -1.0, weight));
      }
    }
    s->result.initial_step_size     = step;
    s->result.initial_primal_weight = weight;
  }
  if (hyper->project_initial_primal) DEV(pdlpdev_project_primal(s->dev));  // pdlp.cu:1041-1056
  DEV(pdlpdev_get_ctl(s->dev, &s->ctl));
  s->result.norm_b = s->norm_b, s->result.norm_c = s->norm_c;
  s->result.step_size = step, s->result.primal_weight = weight;
  return 0;
}

// cuoptamd_solve_sharded runs one host thread per rank.  A rank whose set-up fails before the communicator exists (out of
// memory, a layout that cannot be built ...) would leave the others waiting in ncclCommInitRank for ever: the threads
// therefore agree on their status right before it (this hook, set per thread; max of the ranks' error codes), and a rank
// that fails later aborts the communicators (pdlpdev_comm_abort).
static thread_local std::function<int(int)> t_agree_before_comm;
// test hook: CUOPT_AMD_TUNE="fault_inject=<rank>:create" | "...=<rank>:advance" makes that rank of a sharded solve fail there
static bool fault_injected(int rank, int world, const char* where)
{
  std::string env;
  if (world <= 1 || !cuopt_amd::tune_get("fault_inject", &env)) return false;
  const std::string want = std::to_string(rank) + ":" + where;
  return want == env;
}

int cuoptamd_solver_create(cuoptamd_solver** out, const cuoptamd_lp* lp, const cuoptamd_hyper* hyper,
                           const cuoptamd_settings* settings, const double* init_x,
                           const double* init_y, int device, int rank, int world,
                           const uint8_t* comm_id)
{
  HostRange range("pdlp: solver set-up (partition, transpose, scaling, initial step)");
  if (!out || !lp || !hyper || !settings) return fail(-1, "cuoptamd_solver_create: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(-1, "cuoptamd_solver_create: bad rank/world");
  const auto t0 = clock_type::now();
  const bool timing = std::getenv("CUOPT_AMD_TIMING") != nullptr;
  auto lap = [&, last = t0](const char* what) mutable {
    if (!timing) return;
    const auto now = clock_type::now();
    std::fprintf(stderr, "[cuopt_amd setup] %-28s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - last).count());
    last = now;
  };
  cuoptamd_solver* s = new cuoptamd_solver();
  *out               = s;
  s->H = *hyper, s->S = *settings;
  if (settings->log_file && settings->log_file[0]) s->log_path = settings->log_file;
  s->S.log_file = nullptr;  // the caller's string is not kept
  s->m_global = lp->m, s->n = lp->n, s->rank = rank, s->world = world;
  s->objective_offset = lp->objective_offset;
  const int32_t m = lp->m, n = lp->n;
  if (m == 0) {  // run_pdlp_solver: no constraints -> NumericalError (LP/solve.cu:355-359)
    s->empty_problem = true;
    return 0;
  }
  // maximise -> minimise (problem_helpers.cuh:126-141)
  std::vector<double> c(lp->c, lp->c + n);
  if (lp->maximize) {
    for (double& v : c) v = -v;
    s->objective_scale = -1.0;
  }
  s->maximize               = lp->maximize != 0;
  s->best_quality.objective = s->maximize ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();  // pdlp.cu:183-185
  // ---- device-side set-up (round 5): ONE upload of A; A^T, the analysis pass that looks for structure the matrix arrives without
  // and the panels are built on the GPU (pdlp_device.h "device-side set-up").  CUOPT_AMD_TUNE: device_setup=0 keeps the host
  // transposition + four uploads (the tests' reference construction), reorder=0 skips the ordering search,
  // device_setup_min_nnz=... moves the size from which it is used (small LPs: the resident path, nothing to gain).
  const int64_t nnz_g  = lp->offsets[m];
  const bool sharded   = world > 1 || comm_id != nullptr;
  const bool dev_setup = nnz_g >= cuopt_amd::tune_int("device_setup_min_nnz", 200000) && cuopt_amd::tune_int("device_setup", 1) != 0;
  const bool reorder   = dev_setup && cuopt_amd::tune_int("reorder", 1) != 0;
  struct AnalysisGuard {
    pdlpdev_analysis* an = nullptr;
    void drop() { if (an) pdlpdev_analysis_destroy(an); an = nullptr; }
    ~AnalysisGuard() { drop(); }
  } ag;
  cuoptamd_lp plp = *lp;  // the LP in the order the device works in (the caller's, or P A Q with permuted vectors)
  plp.c           = c.data();
  cuopt_amd::PoolArray<double> pc, plo, phi, plb, pub;  // (pooled, not zero-filled: five 8 MB vectors at 1e6 x 1e6 were 5 ms of first touches)
  cuopt_amd::PoolArray<int32_t> poff, pidx;
  cuopt_amd::PoolArray<double> pval;
  int setup_rc = 0;
  if (dev_setup && (!sharded || reorder)) {
    // (single GPU: the vectors travel while the analysis runs; a sharded rank uploads its own slices later)
    setup_rc = sharded ? pdlpdev_analyze(&ag.an, device, m, n, lp->offsets, lp->indices, lp->values, reorder ? 1 : 0)
                       : pdlpdev_analyze_with_vectors(&ag.an, device, m, n, lp->offsets, lp->indices, lp->values, reorder ? 1 : 0, c.data(), lp->lo,
                                                      lp->hi, lp->lb, lp->ub);
    if (setup_rc != 0) fail(setup_rc, "pdlpdev_analyze: %s", pdlpdev_last_error());
    if (setup_rc == 0) {
      s->row_new2old.resize((size_t)m), s->col_new2old.resize((size_t)n);
      const int permuted = pdlpdev_analysis_maps(ag.an, s->row_new2old.data(), s->col_new2old.data());
      (void)pdlpdev_analysis_info(ag.an, s->analysis_info);
      s->reorder_method = permuted == 1 ? s->analysis_info[1] : 0;
      if (permuted != 1) {
        s->row_new2old.clear(), s->col_new2old.clear();
      } else if (!sharded && pdlpdev_analysis_vectors_in_order(ag.an) == 1) {
        // (the vectors that travelled ahead were gathered into the new order on the device: the context takes them from there -- it is
        //  handed the SAME host pointers the analysis was -- and the host touches none of them)
      } else {
        auto gather = [](const std::vector<int32_t>& new2old, const double* src, cuopt_amd::PoolArray<double>& dst) {
          dst.reset(new2old.size());
          double* out = dst.get();
          const size_t count = new2old.size(), chunk = (count + 15) / 16;
          cuopt_amd::parallel_tasks(16, [&](int t) {
            for (size_t i = (size_t)t * chunk; i < std::min(count, ((size_t)t + 1) * chunk); ++i) out[i] = src[new2old[i]];
          }, (int64_t)count * 16);
          return (const double*)out;
        };
        plp.c = gather(s->col_new2old, c.data(), pc), plp.lb = gather(s->col_new2old, lp->lb, plb), plp.ub = gather(s->col_new2old, lp->ub, pub);
        plp.lo = gather(s->row_new2old, lp->lo, plo), plp.hi = gather(s->row_new2old, lp->hi, phi);
        if (sharded) {  // the ranks slice the permuted matrix on the host (every rank found the same order: the search is deterministic)
          poff.reset((size_t)m + 1), pidx.reset((size_t)std::max<int64_t>(nnz_g, 1)), pval.reset((size_t)std::max<int64_t>(nnz_g, 1));
          setup_rc = pdlpdev_analysis_download(ag.an, 0, poff.get(), pidx.get(), pval.get());
          if (setup_rc != 0) fail(setup_rc, "pdlpdev_analysis_download: %s", pdlpdev_last_error());
          plp.offsets = poff.get(), plp.indices = pidx.get(), plp.values = pval.get();
        }
      }
      if (timing)
        std::fprintf(stderr, "[cuopt_amd setup] ordering: method %d (estimates x1e4: natural %d/%d, levels %d/%d over %d levels, cells %d/%d in %d rounds)\n",
                     s->reorder_method, s->analysis_info[2], s->analysis_info[3], s->analysis_info[4], s->analysis_info[5], s->analysis_info[8],
                     s->analysis_info[6], s->analysis_info[7], s->analysis_info[9]);
    }
    if (sharded) ag.drop();  // (a sharded rank uploads its slice below; the full matrix leaves the device first)
    lap("device analysis");
  }
  const cuoptamd_lp* L = &plp;
  // this rank's row block
  std::vector<int32_t> bounds(world + 1);
  cuoptamd_partition_rows(m, L->offsets, world, bounds.data());
  s->row_begin = bounds[rank], s->row_end = bounds[rank + 1];
  const int32_t ml = s->row_end - s->row_begin;
  if (ag.an && setup_rc == 0) {
    // single GPU: the analysis' device arrays become the context's
    pdlpdev_create_hint(0);
    int rc = pdlpdev_create_from_analysis(&s->dev, ag.an, L->c, L->lo, L->hi, L->lb, L->ub);
    // the spent analysis: its workspace goes back at once (a one-slot cache: the next analysis, of this or of another solver, takes it
    // from there), the hipFree calls of the rest -- each a device synchronisation, ~3 ms of a 25 ms set-up -- wait for the solver's end
    // (CUOPT_AMD_TUNE=keep_analysis_max_nnz: the size up to which they do)
    if (rc == 0 && nnz_g <= cuopt_amd::tune_int("keep_analysis_max_nnz", 50000000)) {
      pdlpdev_analysis_release_workspace(ag.an);
      s->spent_analysis = ag.an, ag.an = nullptr;
    } else {
      ag.drop();
    }
    if (rc != 0) return fail(rc, "pdlpdev_create_from_analysis: %s", pdlpdev_last_error());
  } else {
  const int32_t k0 = L->offsets[s->row_begin];
  std::vector<int32_t> off(ml + 1);
  for (int32_t i = 0; i <= ml; ++i) off[i] = L->offsets[s->row_begin + i] - k0;
  const int64_t nnz_l = off[ml];
  const int32_t* idx  = L->indices + k0;
  const double* val   = L->values + k0;
  std::vector<int32_t> t_off(n + 1);
  cuopt_amd::PoolArray<int32_t> t_idx((size_t)std::max<int64_t>(nnz_l, 1));  // no zero fill of 120 MB, pooled
  cuopt_amd::PoolArray<double> t_val((size_t)std::max<int64_t>(nnz_l, 1));
  lap("partition + slice");
  // The transpose (host threads) runs while the device layer uploads A and builds A's panels; it is joined by the
  // callback right before A^T is needed, and by the guard on every other path.
  struct TransposeJob {
    std::thread worker;
    static void wait(void* self) { static_cast<TransposeJob*>(self)->join(); }
    void join() { if (worker.joinable()) worker.join(); }
    ~TransposeJob() { join(); }
  } job;
  if (setup_rc == 0) {
    const int32_t* off_p = off.data();
    int32_t* t_off_p     = t_off.data();
    int32_t* t_idx_p     = t_idx.get();
    double* t_val_p      = t_val.get();
    job.worker = std::thread([=] { cuoptamd_csr_transpose(ml, n, off_p, idx, val, t_off_p, t_idx_p, t_val_p); });
  }
  {
    pdlpdev_create_hint(world > 1 || comm_id != nullptr);
    int rc = setup_rc;
    if (rc == 0)
      rc = pdlpdev_create_overlapped(&s->dev, device, ml, n, off.data(), idx, val, t_off.data(), t_idx.get(), t_val.get(),
                                     &TransposeJob::wait, &job, L->c, L->lo + s->row_begin, L->hi + s->row_begin, L->lb, L->ub);
    if (rc == 0 && fault_injected(rank, world, "create")) rc = -6;
    job.join();
    if (rc != 0 && rc != setup_rc) fail(rc, "pdlpdev_create: %s", rc == -6 ? "injected fault (CUOPT_AMD_TUNE=fault_inject)" : pdlpdev_last_error());
    if (t_agree_before_comm) {
      const int all = t_agree_before_comm(rc);
      if (rc == 0 && all != 0) return fail(all, "another rank of the sharded solve failed during set-up");
    }
    if (rc != 0) return rc;
  }
  }
  lap("pdlpdev_create (upload+panels)");
  if (comm_id) DEV(pdlpdev_comm_init(s->dev, rank, world, comm_id));
  else if (world > 1) return fail(-1, "cuoptamd_solver_create: world > 1 needs a communicator id");
  DEV(pdlpdev_set_graph_mode(s->dev, settings->use_graph));
  DEV(pdlpdev_problem_norms(s->dev, &s->norm_c, &s->norm_b));
  pdlpdev_step_params sp{hyper->reduction_exponent, hyper->growth_exponent,
                         hyper->primal_distance_smoothing, hyper->dual_distance_smoothing};
  DEV(pdlpdev_set_step_params(s->dev, &sp));

  // ---- [init] block of run_solver (pdlp.cu:984-1075) ----
  double step = hyper->initial_step_size_scaling, weight = hyper->primal_importance;
  auto initial_step_size = [&]() -> int {  // pdlp.cu:1224-1258 (x / 0 -> 0)
    double nr[3];
    DEV(pdlpdev_init_norms(s->dev, nr));
    step = nr[0] == 0.0 ? 0.0 : hyper->initial_step_size_scaling / nr[0];
    return 0;
  };
  auto initial_primal_weight = [&]() -> int {  // pdlp.cu:1260-1309, weighted norms utils.cuh:365-383
    double nr[3];
    DEV(pdlpdev_init_norms(s->dev, nr));
    const double cn = std::sqrt(hyper->initial_primal_weight_c_scaling * nr[1]);
    const double bn = std::sqrt(hyper->initial_primal_weight_b_scaling * nr[2]);
    weight = (bn > 0.0 && cn > 0.0) ? hyper->primal_importance * (cn / bn) : hyper->primal_importance;
    return 0;
  };
  if (hyper->compute_initial_step_size_before_scaling) { int rc = initial_step_size(); if (rc) return rc; }
  if (hyper->compute_initial_primal_weight_before_scaling) { int rc = initial_primal_weight(); if (rc) return rc; }
  lap("norms + params");
  DEV(pdlpdev_scaling_compute(s->dev, hyper->do_ruiz, hyper->ruiz_iterations, hyper->do_pock_chambolle, hyper->alpha_pock_chambolle));
  lap("scaling_compute");
  DEV(pdlpdev_scale_problem(s->dev));
  lap("scale_problem");
  if (comm_id && pdlpdev_shard_dataflow(s->dev) == 3) {  // (also with ONE rank behind a communicator: bench.py --force-comm)
    // owner-computes dataflow: this rank's columns of A over ALL rows (rows of the global A^T), cut from the caller's CSR
    int32_t cb = 0, nc = 0;
    DEV(pdlpdev_owner_slice(s->dev, &cb, &nc));
    std::vector<int32_t> coff((size_t)nc + 1, 0);
    const int64_t nnz_g = L->offsets[m];
    for (int64_t k = 0; k < nnz_g; ++k) {
      const int32_t j = L->indices[k] - cb;
      if (j >= 0 && j < nc) ++coff[j + 1];
    }
    for (int32_t j = 0; j < nc; ++j) coff[j + 1] += coff[j];
    std::vector<int32_t> cidx((size_t)std::max<int32_t>(coff[nc], 1)), cur(coff.begin(), coff.end() - 1);
    std::vector<double> cval((size_t)std::max<int32_t>(coff[nc], 1));
    for (int32_t i = 0; i < m; ++i)  // rows ascending: every column lists its rows in the order an unsharded solve sums them
      for (int32_t k = L->offsets[i]; k < L->offsets[i + 1]; ++k) {
        const int32_t j = L->indices[k] - cb;
        if (j >= 0 && j < nc) {
          const int32_t p = cur[j]++;
          cidx[p] = i, cval[p] = L->values[k];
        }
      }
    DEV(pdlpdev_owner_setup(s->dev, coff.data(), cidx.data(), cval.data(), bounds.data()));
    lap("owner-computes column block");
  }
  if (!hyper->compute_initial_step_size_before_scaling) { int rc = initial_step_size(); if (rc) return rc; }
  if (!hyper->compute_initial_primal_weight_before_scaling) { int rc = initial_primal_weight(); if (rc) return rc; }
  s->computed_step = step, s->computed_weight = weight;
  { int rc = start_run(s, init_x, init_y); if (rc) return rc; }
  lap("initial step/weight/iterate");
  s->result.setup_seconds = seconds_since(t0);
  return 0;
}

void cuoptamd_solver_destroy(cuoptamd_solver* s)
{
  if (!s) return;
  if (s->dev) pdlpdev_destroy(s->dev);
  if (s->spent_analysis) pdlpdev_analysis_destroy(s->spent_analysis);
  delete s;
}

// loop state as freshly constructed (cuoptamd_solver_reset, cuoptamd_batch_reset)
static void reset_host_state(cuoptamd_solver* s)
{
  const cuoptamd_result blank{};
  s->total_iterations = 0, s->iteration_offset = 0, s->attempt_offset = 0, s->major_done_at = -1;
  s->step_error = false, s->need_aty = true, s->last_restart_was_average = false;
  s->last_candidate_kkt = 0.0, s->last_restart_kkt = 0.0, s->gap_reduction_ratio_last_trial = 1.0;
  s->best_quality = cuoptamd_solver::Quality{};
  s->best_quality.objective = s->maximize ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
  s->have_best = false, s->best_result = blank;
  s->have_accepted = false, s->accepted_result = blank;
  s->conv_current = Convergence{}, s->conv_average = Convergence{};
  s->returned_which = PDLPDEV_CURRENT, s->finished = false, s->warm_started = false, s->started = false;
  s->result = blank;
}

int cuoptamd_solver_reset(cuoptamd_solver* s, const double* lb, const double* ub, const double* lo, const double* hi,
                          const cuoptamd_settings* settings, const double* init_x, const double* init_y)
{
  if (!s) return fail(-1, "cuoptamd_solver_reset: null solver");
  const auto t0 = clock_type::now();
  if (settings) {
    s->S = *settings;
    if (settings->log_file && settings->log_file[0]) s->log_path = settings->log_file;
    s->S.log_file = nullptr;
  }
  reset_host_state(s);
  if (s->empty_problem) return 0;
  {
    std::vector<double> tlb, tub, tlo, thi;  // the new bounds in the device's order
    lb = s->cols_in(lb, tlb), ub = s->cols_in(ub, tub), lo = s->rows_in(lo, tlo), hi = s->rows_in(hi, thi);
    DEV(pdlpdev_reset(s->dev, lb, ub, lo ? lo + s->row_begin : nullptr, hi ? hi + s->row_begin : nullptr));
  }
  const bool timing = std::getenv("CUOPT_AMD_TIMING") != nullptr;
  const double t_reset = seconds_since(t0);
  // ||b||, ||c|| of the termination rule: from the problem again (the previous settings may have overridden them)
  DEV(pdlpdev_problem_norms(s->dev, &s->norm_c, &s->norm_b));
  if (lo || hi) {
    // the initial primal weight depends on the row bounds (pdlp.cu:1260-1309)
    double nr[2];
    DEV(pdlpdev_weight_norms(s->dev, s->H.compute_initial_primal_weight_before_scaling, nr));
    const double cn = std::sqrt(s->H.initial_primal_weight_c_scaling * nr[0]);
    const double bn = std::sqrt(s->H.initial_primal_weight_b_scaling * nr[1]);
    s->computed_weight = (bn > 0.0 && cn > 0.0) ? s->H.primal_importance * (cn / bn) : s->H.primal_importance;
  }
  DEV(pdlpdev_set_graph_mode(s->dev, s->S.use_graph));
  const double t_norms = seconds_since(t0);
  { int rc = start_run(s, init_x, init_y); if (rc) return rc; }
  s->result.setup_seconds = seconds_since(t0);
  if (timing)
    fprintf(stderr, "[cuopt_amd setup] reset: bounds to the device %.2f ms, norms + weight %.2f ms, initial iterate / step %.2f ms\n", 1e3 * t_reset,
            1e3 * (t_norms - t_reset), 1e3 * (s->result.setup_seconds - t_norms));
  return 0;
}

// One trip of the main loop up to the attempts (pdlp.cu:1099-1222): the major iteration when one is due (termination checks, restart,
// primal weight), the iteration budget, then what the attempts need (a cleared step error, A^T y of a fresh iterate).  *stop: the
// solve is over or the budget is used up (s->result says which); otherwise *target = the accepted-step count (the device's) the
// attempts run to.  cuoptamd_solver_advance runs one LP through it, cuoptamd_batch_advance K of them in lockstep.
static bool major_due(const cuoptamd_solver* s)
{
  const cuoptamd_hyper& H = s->H;
  const int32_t it = s->total_iterations;
  const bool major = (it % H.major_iteration == 0 && it > 0) || it <= H.min_iteration_restart;
  // should_do_artificial_restart (pdlp_restart_strategy.cu:939-961), Fast1 only
  const bool artificial = H.artificial_restart_in_main_loop && s->ctl.its_since_restart >= H.artificial_restart_threshold * it;
  return (major || artificial || s->step_error) && s->major_done_at != it;
}
static void major_was_done(cuoptamd_solver* s, bool terminated)
{
  s->major_done_at = s->total_iterations;
  if (!terminated) return;
  s->finished = true;
  log_line(s, "%7d %+.8e %+.8e  %8.2e   %8.2e     %8.2e   %.3fs\n", s->result.steps_taken, s->result.primal_objective,
           s->result.dual_objective, s->result.gap, s->result.l2_primal_residual, s->result.l2_dual_residual,
           seconds_since(s->solve_start));
  log_line(s, "PDLP finished: status %d, %d iterations, %d restarts\n", s->result.status, s->result.steps_taken,
           s->result.num_restarts);
}
// the iteration budget, then the accepted-step count (the device's) the next attempts run to; false: the budget is used up
static bool next_target(cuoptamd_solver* s, int32_t budget_end, int32_t* target_out)
{
  const cuoptamd_hyper& H = s->H;
  const int32_t it = s->total_iterations;
  if (it >= budget_end) {
    s->result.status          = kNoTermination;
    s->result.steps_taken     = s->ctl.steps_taken;
    s->result.attempted_steps = s->ctl.attempts;
    s->result.step_size       = s->ctl.step_size;
    s->result.primal_weight   = s->ctl.primal_weight;
    return false;
  }
  // ---- take_step(s) up to the next major iteration (pdlp.cu:1187-1222), no host round trips ----
  int32_t next_major;
  if (it + 1 <= H.min_iteration_restart || H.artificial_restart_in_main_loop)
    next_major = it + 1;
  else
    next_major = (it / H.major_iteration + 1) * H.major_iteration;
  *target_out = std::min(next_major, budget_end) - s->iteration_offset;
  return true;
}
static int advance_to_attempts(cuoptamd_solver* s, int32_t budget_end, int32_t* target_out, bool* stop)
{
  *stop = false;
  if (s->total_iterations >= s->H.major_iteration && fault_injected(s->rank, s->world, "advance")) return fail(-6, "injected fault (CUOPT_AMD_TUNE=fault_inject)");
  if (major_due(s)) {
    bool terminated = false;
    int rc          = major_iteration(s, &terminated);
    if (rc != 0) return rc;
    major_was_done(s, terminated);
    if (terminated) {
      *stop = true;
      return 0;
    }
  }
  if (!next_target(s, budget_end, target_out)) {
    *stop = true;
    return 0;
  }
  if (s->step_error) {  // take_step re-arms valid_step_size = 0 (pdlp.cu:1190)
    s->step_error = false;
    int rc = pdlpdev_clear_error(s->dev);
    if (rc != 0) return fail(rc, "pdlpdev_clear_error: %s", pdlpdev_last_error());
  }
  if (s->need_aty) {
    int rc = pdlpdev_compute_aty(s->dev);
    if (rc != 0) return fail(rc, "pdlpdev_compute_aty: %s", pdlpdev_last_error());
    s->need_aty = false;
  }
  return 0;
}
static void advance_after_attempts(cuoptamd_solver* s)  // (s->ctl: the control block the attempts left)
{
  s->total_iterations = s->iteration_offset + s->ctl.steps_taken;
  if (s->ctl.error) s->step_error = true;
}
static void advance_begin(cuoptamd_solver* s, clock_type::time_point t0)
{
  if (s->started) return;
  s->started     = true;
  s->solve_start = t0;
  log_line(s, "PDLP on gfx950: %d constraints, %d variables\n", s->m_global, s->n);
  log_line(s, "   Iter    Primal Obj.      Dual Obj.    Gap        Primal Res.  Dual Res.   Time\n");  // pdlp.cu:1077-1080
}
static int32_t advance_budget_end(const cuoptamd_solver* s, int32_t max_new_iterations)
{
  const int64_t budget_end64 = (int64_t)s->total_iterations + std::max<int64_t>(max_new_iterations, 0);
  return (int32_t)std::min<int64_t>(budget_end64, std::numeric_limits<int32_t>::max());
}

int cuoptamd_solver_advance(cuoptamd_solver* s, int32_t max_new_iterations, cuoptamd_result* result)
{
  if (!s) return fail(-1, "cuoptamd_solver_advance: null solver");
  const auto t0 = clock_type::now();
  advance_begin(s, t0);
  auto leave = [&](int rc) {
    s->result.gpus = s->world;
    s->result.loop_seconds += seconds_since(t0);
    if (result) *result = s->result;
    return rc;
  };
  if (s->empty_problem) {
    s->result.status = kNumericalError;
    s->finished      = true;
    return leave(0);
  }
  if (s->finished) return leave(0);
  const int32_t budget_end = advance_budget_end(s, max_new_iterations);
  for (;;) {
    int32_t target = 0;
    bool stop      = false;
    int rc         = advance_to_attempts(s, budget_end, &target, &stop);
    if (rc != 0 || stop) return leave(rc);
    rc = pdlpdev_run(s->dev, target, &s->ctl);
    if (rc != 0) return leave(fail(rc, "pdlpdev_run: %s", pdlpdev_last_error()));
    advance_after_attempts(s);
  }
}

// ---- K LPs over ONE matrix and objective in lockstep (kernels_batch.hip) --------------------------------------------------------
struct cuoptamd_batch {
  int K = 0;
  std::vector<cuoptamd_solver*> s;
  pdlpdev_batch* dev = nullptr;          // K <= 16 LPs over ONE matrix in lockstep (kernels_batch.hip), or ...
  pdlpdev_small_batch* small = nullptr;  // ... K resident small LPs, one workgroup each (kernels_resident.hip)
};

int cuoptamd_solver_clone(cuoptamd_solver* parent, const double* lb, const double* ub, const double* lo, const double* hi,
                          const cuoptamd_settings* settings, cuoptamd_solver** out)
{
  if (!parent || !out) return fail(-1, "cuoptamd_solver_clone: null argument");
  if (parent->empty_problem || !parent->dev || parent->world != 1) return fail(-7, "cuoptamd_solver_clone: not for empty or sharded solvers");
  const bool timing = std::getenv("CUOPT_AMD_TIMING") != nullptr;
  const auto t0     = clock_type::now();
  pdlpdev_ctx* dev = nullptr;
  int rc           = pdlpdev_clone_shared(&dev, parent->dev);
  if (rc != 0) {
    if (dev) pdlpdev_destroy(dev);
    return fail(rc, "pdlpdev_clone_shared: %s", pdlpdev_last_error());
  }
  const double t_dev = seconds_since(t0);
  cuoptamd_solver* s = new cuoptamd_solver(*parent);
  s->spent_analysis = nullptr;  // (the parent's to release)
  s->dev             = dev;
  rc = cuoptamd_solver_reset(s, lb, ub, lo, hi, settings, nullptr, nullptr);
  if (timing) fprintf(stderr, "[cuopt_amd setup] clone: device context %.2f ms, reset to its bounds %.2f ms\n", 1e3 * t_dev, 1e3 * (seconds_since(t0) - t_dev));
  if (rc != 0) {
    cuoptamd_solver_destroy(s);
    return rc;
  }
  *out = s;
  return 0;
}

int cuoptamd_batch_create(cuoptamd_solver** solvers, int K, cuoptamd_batch** out)
{
  if (!solvers || !out || K < 1) return fail(-1, "cuoptamd_batch_create: null argument");
  std::vector<pdlpdev_ctx*> ctx(K);
  for (int l = 0; l < K; ++l) {
    if (!solvers[l] || !solvers[l]->dev) return fail(-1, "cuoptamd_batch_create: null solver");
    ctx[l] = solvers[l]->dev;
  }
  // small LPs first: any number of them, any matrices, a workgroup each
  pdlpdev_small_batch* small = nullptr;
  int rc                     = pdlpdev_small_batch_create(&small, ctx.data(), K);
  pdlpdev_batch* dev         = nullptr;
  if (rc == -7) {  // not the resident path: 2, 4, 8 or 16 LPs over one matrix in lockstep
    if (K > 16) return fail(-7, "cuoptamd_batch_create: more than 16 LPs need the resident small-LP path (%s)", pdlpdev_last_error());
    rc = pdlpdev_batch_create(&dev, ctx.data(), K);
  }
  if (rc != 0) {
    if (dev) pdlpdev_batch_destroy(dev);
    return fail(rc, "%s", pdlpdev_last_error());
  }
  cuoptamd_batch* b = new cuoptamd_batch();
  b->K = K, b->dev = dev, b->small = small;
  b->s.assign(solvers, solvers + K);
  *out = b;
  return 0;
}

pdlpdev_batch* cuoptamd_batch_device(cuoptamd_batch* b) { return b ? b->dev : nullptr; }

void cuoptamd_batch_destroy(cuoptamd_batch* b)
{
  if (!b) return;
  if (b->dev) pdlpdev_batch_destroy(b->dev);
  if (b->small) pdlpdev_small_batch_destroy(b->small);
  delete b;
}

// K resident small LPs: every phase of the loop that touches the device is ONE launch over the LPs that are in it
// (pdlpdev_small_batch_*), the scalar logic in between runs per LP on the host exactly as in cuoptamd_solver_advance
static int small_batch_advance(cuoptamd_batch* b, const std::vector<int32_t>& budget_end, std::vector<char>& done)
{
  const int K = b->K;
  std::vector<pdlpdev_small_eval> req(K), ahead(K);
  std::vector<double> ev((size_t)K * PDLPDEV_EV_COUNT), ev_avg((size_t)K * PDLPDEV_EV_COUNT), dist2(2 * (size_t)K), weight(K);
  std::vector<int32_t> which(K), unscaled(K), target(K), clear(K), aty(K), evaluated(K, 0);
  std::vector<char> due(K);
  std::vector<MajorPlan> plan(K);
  std::vector<pdlpdev_ctl> ctl(K);
  auto same_request = [](const pdlpdev_small_eval& a, const pdlpdev_small_eval& c) {
    return a.mode == c.mode && a.rule_finite == c.rule_finite && a.eps_p == c.eps_p && a.eps_d == c.eps_d;
  };
  for (;;) {
    // ---- major iterations that are due: one evaluation launch (unless the evaluation already ran behind the attempts), the heads on
    // the host, one restart launch
    bool any = false, launch = false;
    for (int l = 0; l < K; ++l) {
      req[l].mode = -1, due[l] = 0;
      if (done[l] || !major_due(b->s[l])) continue;
      due[l] = 1, any = true;
      const pdlpdev_small_eval rq = major_eval_request(b->s[l]);
      if (evaluated[l] && same_request(rq, ahead[l])) continue;  // ev / ev_avg of LP l are in place
      req[l] = rq, launch = true;
    }
    std::fill(evaluated.begin(), evaluated.end(), 0);
    std::fill(weight.begin(), weight.end(), -1.0);
    if (any) {
      HostRange range("pdlp: major iterations of a small-LP batch");
      int rc = launch ? pdlpdev_small_batch_major_eval(b->small, req.data(), ev.data(), ev_avg.data()) : 0;
      if (rc != 0) return fail(rc, "pdlpdev_small_batch_major_eval: %s", pdlpdev_last_error());
      bool any_restart = false;
      for (int l = 0; l < K; ++l) {
        which[l] = -1, unscaled[l] = 0;
        if (!due[l]) continue;
        bool terminated = false;
        rc = major_head(b->s[l], &ev[(size_t)l * PDLPDEV_EV_COUNT], &ev_avg[(size_t)l * PDLPDEV_EV_COUNT], &terminated, &plan[l]);
        if (rc != 0) return rc;
        major_was_done(b->s[l], terminated);
        if (terminated) done[l] = 1;
        else if (plan[l].restart) which[l] = plan[l].which, unscaled[l] = plan[l].unscaled, any_restart = true;
      }
      if (any_restart) {
        rc = pdlpdev_small_batch_restart(b->small, which.data(), unscaled.data(), dist2.data());
        if (rc != 0) return fail(rc, "pdlpdev_small_batch_restart: %s", pdlpdev_last_error());
        for (int l = 0; l < K; ++l)
          if (which[l] >= 0) weight[l] = major_restart_done(b->s[l], plan[l], &dist2[2 * (size_t)l]);
      }
    }
    // ---- budgets and targets; what the attempts need first (new weights, a cleared step error, A^T y of a fresh iterate)
    any = false;
    bool any_prepare = false;
    for (int l = 0; l < K; ++l) {
      target[l] = 0, clear[l] = 0, aty[l] = 0, ahead[l].mode = -1;
      if (weight[l] > 0.0) any_prepare = true;  // (also for an LP whose budget ends here: the device keeps what a later call continues from)
      if (done[l]) continue;
      cuoptamd_solver* s = b->s[l];
      if (!next_target(s, budget_end[l], &target[l])) {
        done[l] = 1, target[l] = 0;
        continue;
      }
      any = true;
      if (s->step_error) s->step_error = false, clear[l] = 1, any_prepare = true;
      if (s->need_aty) s->need_aty = false, aty[l] = 1, any_prepare = true;
      // the major iteration these attempts end in (when the target is a boundary of the schedule, the usual case): its evaluation is
      // enqueued right behind them.  The request is what major_eval_request will say once the attempts are done -- verified then.
      const cuoptamd_hyper& H = s->H;
      const int32_t it_after = target[l] + s->iteration_offset;
      if ((it_after % H.major_iteration == 0 && it_after > 0) || it_after <= H.min_iteration_restart) {
        ahead[l]      = major_eval_request(s);
        ahead[l].mode = (target[l] <= 1 && !s->warm_started) ? 0 : 2;
      }
    }
    if (any_prepare) {
      int rc = pdlpdev_small_batch_prepare(b->small, clear.data(), weight.data(), aty.data());
      if (rc != 0) return fail(rc, "pdlpdev_small_batch_prepare: %s", pdlpdev_last_error());
    }
    if (!any) return 0;
    int rc = pdlpdev_small_batch_run(b->small, target.data(), ctl.data(), ahead.data(), ev.data(), ev_avg.data(), evaluated.data());
    if (rc != 0) return fail(rc, "pdlpdev_small_batch_run: %s", pdlpdev_last_error());
    for (int l = 0; l < K; ++l)
      if (target[l] > 0) {
        b->s[l]->ctl = ctl[l];
        advance_after_attempts(b->s[l]);
      }
  }
}


// every LP of the batch up to max_new_iterations further iterations (or to its verdict); results[l] as cuoptamd_solver_advance's.
// An LP that finishes rests while the others go on; each LP's trajectory is the one its own cuoptamd_solver_advance would take.
int cuoptamd_batch_advance(cuoptamd_batch* b, int32_t max_new_iterations, cuoptamd_result* results)
{
  if (!b) return fail(-1, "cuoptamd_batch_advance: null batch");
  const auto t0 = clock_type::now();
  const int K   = b->K;
  std::vector<int32_t> budget_end(K), target(K);
  std::vector<char> done(K);
  for (int l = 0; l < K; ++l) {
    cuoptamd_solver* s = b->s[l];
    advance_begin(s, t0);
    if (s->empty_problem && !s->finished) s->result.status = kNumericalError, s->finished = true;
    done[l]       = s->finished;
    budget_end[l] = advance_budget_end(s, max_new_iterations);
  }
  auto leave = [&](int rc) {
    const double dt = seconds_since(t0);
    for (int l = 0; l < K; ++l) {
      b->s[l]->result.gpus = 1;
      b->s[l]->result.loop_seconds += dt;  // (wall time of the batch: the LPs ran side by side)
      if (results) results[l] = b->s[l]->result;
    }
    return rc;
  };
  if (b->small) return leave(small_batch_advance(b, budget_end, done));
  std::vector<pdlpdev_ctl> ctl(K);
  for (;;) {
    bool any = false;
    for (int l = 0; l < K; ++l) {
      target[l] = 0;
      if (done[l]) continue;
      bool stop = false;
      int rc    = advance_to_attempts(b->s[l], budget_end[l], &target[l], &stop);
      if (rc != 0) return leave(rc);
      if (stop) done[l] = true, target[l] = 0;
      else any = true;
    }
    if (!any) return leave(0);
    int rc = pdlpdev_batch_run(b->dev, target.data(), ctl.data());
    if (rc != 0) return leave(fail(rc, "pdlpdev_batch_run: %s", pdlpdev_last_error()));
    for (int l = 0; l < K; ++l)
      if (target[l] > 0) {
        b->s[l]->ctl = ctl[l];
        advance_after_attempts(b->s[l]);
      }
  }
}

// cuoptamd_solver_reset(lb[l], ub[l], NULL, NULL, NULL, init_x[l], init_y[l]) for every solver of a small-LP batch in ONE launch (the MIP
// heuristics' re-solve: other variable bounds, the previous primal / dual as the start; relaxed_lp.cu:74-108).  Arrays and entries may
// be NULL (all four entries NULL: that solver is reset to a cold start under its current bounds).  -7: not a small-LP batch, or a
// solver whose preset updates step size / primal weight from the initial iterate (none does): use cuoptamd_solver_reset per solver.
static int batch_reset_impl(cuoptamd_batch* b, const double* const* lb, const double* const* ub, const double* const* init_x, const double* const* init_y,
                            const int32_t* var, const double* var_lb, const double* var_ub, bool warm_from_own)
{
  if (!b) return fail(-1, "cuoptamd_batch_reset: null batch");
  if (!b->small) return fail(-7, "cuoptamd_batch_reset: a batch of resident small LPs only");
  const auto t0 = clock_type::now();
  const int K   = b->K;
  std::vector<int32_t> take(K, 1), k(K, -1), warm(K, -1);
  std::vector<double> step(K), weight(K);
  std::vector<pdlpdev_ctl> ctl(K);
  int project = -1;
  for (int l = 0; l < K; ++l) {
    cuoptamd_solver* s = b->s[l];
    if (s->empty_problem) { take[l] = 0; continue; }
    if (s->H.update_step_size_on_initial_solution || s->H.update_primal_weight_on_initial_solution || !s->row_new2old.empty())
      return fail(-7, "cuoptamd_batch_reset: solver %d needs cuoptamd_solver_reset", l);
    if (project >= 0 && project != (s->H.project_initial_primal ? 1 : 0)) return fail(-7, "cuoptamd_batch_reset: the solvers' presets differ");
    project   = s->H.project_initial_primal ? 1 : 0;
    step[l]   = s->S.initial_step_size >= 0.0 ? s->S.initial_step_size : s->computed_step;      // start_run
    weight[l] = s->S.initial_primal_weight >= 0.0 ? s->S.initial_primal_weight : s->computed_weight;
    k[l]      = s->S.initial_k;
    if (warm_from_own) warm[l] = s->returned_which;  // what cuoptamd_solver_get_solution would have handed the caller
  }
  for (int l = 0; l < K; ++l) reset_host_state(b->s[l]);
  int rc = pdlpdev_small_batch_reset(b->small, take.data(), lb, ub, init_x, init_y, step.data(), weight.data(), k.data(), std::max(project, 0), ctl.data(), var,
                                     var_lb, var_ub, warm_from_own ? warm.data() : nullptr);
  if (rc != 0) return fail(rc, "pdlpdev_small_batch_reset: %s", pdlpdev_last_error());
  const double dt = seconds_since(t0);
  for (int l = 0; l < K; ++l) {
    cuoptamd_solver* s = b->s[l];
    if (!take[l]) continue;
    if (s->S.relative_primal_tolerance_factor >= 0.0) s->norm_b = s->S.relative_primal_tolerance_factor;
    if (s->S.relative_dual_tolerance_factor >= 0.0) s->norm_c = s->S.relative_dual_tolerance_factor;
    s->ctl = ctl[l];
    s->result.initial_step_size = step[l], s->result.initial_primal_weight = weight[l];
    s->result.norm_b = s->norm_b, s->result.norm_c = s->norm_c;
    s->result.step_size = step[l], s->result.primal_weight = weight[l];
    s->result.setup_seconds = dt;
  }
  return 0;
}

int cuoptamd_batch_reset(cuoptamd_batch* b, const double* const* lb, const double* const* ub, const double* const* init_x, const double* const* init_y)
{
  return batch_reset_impl(b, lb, ub, init_x, init_y, nullptr, nullptr, nullptr, false);
}

// One branch-and-bound step in every node of the batch, one launch, three scalars per node across PCIe: the bounds of variable var[l]
// of solver l become [lb[l], ub[l]] (var[l] < 0: unchanged), and the solver starts again from the primal / dual its LAST solve
// returned -- exactly cuoptamd_solver_reset(full lb, full ub, NULL, NULL, NULL, x, y) with (x, y) = cuoptamd_solver_get_solution,
// without the two trips through the host (relaxed_lp.cu:74-108 keeps bounds and lp_state on the device in the same way).
int cuoptamd_batch_branch(cuoptamd_batch* b, const int32_t* var, const double* lb, const double* ub)
{
  return batch_reset_impl(b, nullptr, nullptr, nullptr, nullptr, var, lb, ub, true);
}

// cuoptamd_solver_get_solution for every solver of a small-LP batch in one launch (arrays and entries may be NULL)
int cuoptamd_batch_get_solutions(cuoptamd_batch* b, double* const* x, double* const* y, double* const* rc)
{
  if (!b) return fail(-1, "cuoptamd_batch_get_solutions: null batch");
  if (!b->small) {
    for (int l = 0; l < b->K; ++l) {
      int rc_ = cuoptamd_solver_get_solution(b->s[l], x ? x[l] : nullptr, y ? y[l] : nullptr, rc ? rc[l] : nullptr);
      if (rc_ != 0) return rc_;
    }
    return 0;
  }
  std::vector<int32_t> which(b->K);
  for (int l = 0; l < b->K; ++l) {
    cuoptamd_solver* s = b->s[l];
    which[l] = -1;
    if (s->empty_problem || !s->dev) {
      if (x && x[l]) std::fill(x[l], x[l] + s->n, 0.0);
      if (rc && rc[l]) std::fill(rc[l], rc[l] + s->n, 0.0);
    } else {
      which[l] = s->returned_which;
    }
  }
  int rc_ = pdlpdev_small_batch_get_solutions(b->small, which.data(), x, y, rc);
  if (rc_ != 0) return fail(rc_, "pdlpdev_small_batch_get_solutions: %s", pdlpdev_last_error());
  return 0;
}

// the same without the copies: x[l] / y[l] / rc[l] receive pointers INTO the batch's pinned staging block (any of the three arrays may
// be NULL), valid until the next cuoptamd_batch_reset / _branch / _get_solutions / _solution_views of this batch.  Small-LP batches only.
int cuoptamd_batch_solution_views(cuoptamd_batch* b, const double** x, const double** y, const double** rc)
{
  if (!b) return fail(-1, "cuoptamd_batch_solution_views: null batch");
  if (!b->small) return fail(-7, "cuoptamd_batch_solution_views: a batch of resident small LPs only");
  std::vector<int32_t> which(b->K);
  for (int l = 0; l < b->K; ++l) which[l] = (b->s[l]->empty_problem || !b->s[l]->dev) ? -1 : b->s[l]->returned_which;
  int rc_ = pdlpdev_small_batch_solution_views(b->small, which.data(), x, y, rc);
  if (rc_ != 0) return fail(rc_, "pdlpdev_small_batch_solution_views: %s", pdlpdev_last_error());
  return 0;
}

int cuoptamd_solver_get_solution(cuoptamd_solver* s, double* x, double* y, double* rc)
{
  if (!s) return fail(-1, "null solver");
  if (s->empty_problem || !s->dev) {
    if (x) std::fill(x, x + s->n, 0.0);
    if (rc) std::fill(rc, rc + s->n, 0.0);
    return 0;
  }
  if (y && s->world > 1) std::fill(y, y + s->m_global, 0.0);  // other ranks' rows stay 0
  if (s->row_new2old.empty()) {
    DEV(pdlpdev_get_solution(s->dev, s->returned_which, x, y ? y + s->row_begin : nullptr, rc));
    return 0;
  }
  // the device's order -> the caller's
  std::vector<double> tx(x ? (size_t)s->n : 0), ty(y ? (size_t)s->m_global : 0), trc(rc ? (size_t)s->n : 0);
  DEV(pdlpdev_get_solution(s->dev, s->returned_which, x ? tx.data() : nullptr, y ? ty.data() + s->row_begin : nullptr, rc ? trc.data() : nullptr));
  if (x) cuoptamd_solver::to_caller_order(s->col_new2old, tx.data(), x, 0, (size_t)s->n);
  if (rc) cuoptamd_solver::to_caller_order(s->col_new2old, trc.data(), rc, 0, (size_t)s->n);
  if (y) cuoptamd_solver::to_caller_order(s->row_new2old, ty.data(), y, (size_t)s->row_begin, (size_t)s->row_end);
  return 0;
}

int cuoptamd_solve_sharded(const cuoptamd_lp* lp, const cuoptamd_hyper* hyper, const cuoptamd_settings* settings,
                           int gpus, int soft_communicator, cuoptamd_result* result, double* x, double* y, double* rc)
{
  if (!lp || !hyper || !settings || !result) return fail(-1, "cuoptamd_solve_sharded: null argument");
  if (gpus < 1 || gpus > 16) return fail(-1, "cuoptamd_solve_sharded: 1..16 row blocks");
  if (!soft_communicator && pdlpdev_device_count() < gpus)
    return fail(-5, "cuoptamd_solve_sharded: %d GPUs requested, %d visible (there is no CPU fallback)", gpus,
                pdlpdev_device_count());
  uint8_t id[128];
  if (soft_communicator) {
    if (pdlpdev_softcomm_create(gpus, id) != 0) return fail(-6, "%s", pdlpdev_last_error());
  } else {
    if (pdlpdev_comm_unique_id(id) != 0) return fail(-6, "%s", pdlpdev_last_error());
  }
  struct Rank {
    int rc = 0;
    std::string error;
    cuoptamd_result res{};
  };
  std::vector<Rank> ranks(gpus);
  if (y) std::fill(y, y + lp->m, 0.0);
  struct Agreement {  // one-shot barrier of the rank threads carrying the worst error code
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, worst = 0;
  } agreement;
  std::vector<std::thread> pool;
  for (int g = 0; g < gpus; ++g)
    pool.emplace_back([&, g] {
      Rank& me = ranks[g];
      cuoptamd_settings st = *settings;
      if (g != 0) st.log_to_console = 0, st.log_file = nullptr;  // one voice
      cuoptamd_solver* solver = nullptr;
      t_agree_before_comm = [&](int rc) {
        std::unique_lock<std::mutex> lk(agreement.mu);
        if (rc != 0 && agreement.worst == 0) agreement.worst = rc;
        if (++agreement.arrived == gpus) agreement.cv.notify_all();
        else agreement.cv.wait(lk, [&] { return agreement.arrived == gpus; });
        return agreement.worst;
      };
      me.rc = cuoptamd_solver_create(&solver, lp, hyper, &st, nullptr, nullptr, soft_communicator ? 0 : g, g, gpus, id);
      t_agree_before_comm = nullptr;
      if (me.rc == 0) {
        me.rc = cuoptamd_solver_advance(solver, std::numeric_limits<int32_t>::max(), &me.res);
      }
      if (me.rc != 0) {
        me.error = cuoptamd_last_error();  // thread-local message: carry it to the caller's thread
        (void)pdlpdev_comm_abort(id);      // nobody waits for this rank in a collective
      }
      if (me.rc == 0) {
        // every rank holds the same x and reduced costs (replicated primal side) and ITS rows of y
        std::vector<double> yl(y ? (size_t)lp->m : 0);
        me.rc = cuoptamd_solver_get_solution(solver, g == 0 ? x : nullptr, y ? yl.data() : nullptr, g == 0 ? rc : nullptr);
        if (me.rc == 0 && y) {
          // a rank's rows are a contiguous block of the DEVICE's order -- scattered in the caller's when the set-up reordered the
          // LP; what the rank does not own is exactly zero, and no two ranks own the same row
          static std::mutex y_mutex;
          std::lock_guard<std::mutex> lock(y_mutex);
          for (int32_t i = 0; i < lp->m; ++i)
            if (yl[i] != 0.0) y[i] = yl[i];
        }
      }
      if (me.rc != 0 && me.error.empty()) me.error = cuoptamd_last_error();
      cuoptamd_solver_destroy(solver);
    });
  for (auto& t : pool) t.join();
  // report the rank that failed first-hand, not one that merely noticed ("another rank failed", an aborted collective)
  for (int pass = 0; pass < 2; ++pass)
    for (int g = 0; g < gpus; ++g)
      if (ranks[g].rc != 0 && (pass == 1 || (ranks[g].error.find("another rank") == std::string::npos && ranks[g].error.find("abandoned") == std::string::npos)))
        return fail(ranks[g].rc, "rank %d of %d: %s", g, gpus, ranks[g].error.c_str());
  *result      = ranks[0].res;
  result->gpus = gpus;
  return 0;
}

int cuoptamd_solver_get_warm_start(cuoptamd_solver* s, cuoptamd_warm_start* ws)
{
  if (!s || !ws || !s->dev) return fail(-1, "cuoptamd_solver_get_warm_start: null argument");
  // Sharded solver: the primal-side vectors are replicated, of the dual-side (m-sized) vectors this rank owns rows
  // [row_begin, row_end): they land at their global positions, the other rows are zeroed -- the snapshots of all ranks
  // add up to the full one (pdlp.cu:468-489 copies whole vectors).
  ws->n_variables = s->n, ws->n_constraints = s->m_global;
  const int64_t n = s->n, ml = s->row_end - s->row_begin;
  auto own_rows = [&](double* v) -> double* {
    if (!v) return nullptr;
    if (s->world > 1) std::fill(v, v + s->m_global, 0.0);
    return v + s->row_begin;
  };
  if (!s->row_new2old.empty()) {
    // A reordered LP: every vector leaves in the CALLER's order (a snapshot can be restored into any solver of the same LP, whatever
    // order its own set-up found).  Device order -> temporaries -> scatter.
    std::vector<double> tn((size_t)n), tm((size_t)s->m_global);
    auto cols = [&](double* user) { if (user) cuoptamd_solver::to_caller_order(s->col_new2old, tn.data(), user, 0, (size_t)n); };
    auto rows = [&](double* user) {
      if (!user) return;
      if (s->world > 1) std::fill(user, user + s->m_global, 0.0);
      cuoptamd_solver::to_caller_order(s->row_new2old, tm.data(), user, (size_t)s->row_begin, (size_t)s->row_end);
    };
    for (int which : {PDLPDEV_CURRENT, PDLPDEV_AVERAGE}) {
      double* ux = which == PDLPDEV_CURRENT ? ws->current_primal_solution : ws->initial_primal_average;
      double* uy = which == PDLPDEV_CURRENT ? ws->current_dual_solution : ws->initial_dual_average;
      DEV(pdlpdev_get_solution(s->dev, which, ux ? tn.data() : nullptr, uy ? tm.data() + s->row_begin : nullptr, nullptr));
      cols(ux), rows(uy);
    }
    auto get_n = [&](int id, double* user) -> int {
      if (!user) return 0;
      if (pdlpdev_download(s->dev, id, tn.data(), n) != n) return fail(-2, "download failed: %s", pdlpdev_last_error());
      cols(user);
      return 0;
    };
    auto get_m = [&](int id, double* user) -> int {
      if (!user) return 0;
      if (pdlpdev_download(s->dev, id, tm.data() + s->row_begin, ml) != ml) return fail(-2, "download failed: %s", pdlpdev_last_error());
      rows(user);
      return 0;
    };
    int rc;
    if ((rc = get_n(PDLPDEV_BUF_X, ws->current_primal_solution_scaled))) return rc;
    if ((rc = get_m(PDLPDEV_BUF_Y, ws->current_dual_solution_scaled))) return rc;
    if ((rc = get_n(PDLPDEV_BUF_ATY, ws->current_ATY))) return rc;
    if ((rc = get_n(PDLPDEV_BUF_SUM_X, ws->sum_primal_solutions))) return rc;
    if ((rc = get_m(PDLPDEV_BUF_SUM_Y, ws->sum_dual_solutions))) return rc;
    if ((rc = get_n(PDLPDEV_BUF_LAST_RESTART_X, ws->last_restart_duality_gap_primal_solution))) return rc;
    if ((rc = get_m(PDLPDEV_BUF_LAST_RESTART_Y, ws->last_restart_duality_gap_dual_solution))) return rc;
  } else {
  DEV(pdlpdev_get_solution(s->dev, PDLPDEV_CURRENT, ws->current_primal_solution, own_rows(ws->current_dual_solution), nullptr));
  DEV(pdlpdev_get_solution(s->dev, PDLPDEV_AVERAGE, ws->initial_primal_average, own_rows(ws->initial_dual_average), nullptr));
  auto get = [&](int id, double* dst, int64_t count) -> int {
    if (!dst) return 0;
    if (pdlpdev_download(s->dev, id, dst, count) != count) return fail(-2, "download failed: %s", pdlpdev_last_error());
    return 0;
  };
  int rc;
  if ((rc = get(PDLPDEV_BUF_X, ws->current_primal_solution_scaled, n))) return rc;
  if ((rc = get(PDLPDEV_BUF_Y, own_rows(ws->current_dual_solution_scaled), ml))) return rc;
  if ((rc = get(PDLPDEV_BUF_ATY, ws->current_ATY, n))) return rc;
  if ((rc = get(PDLPDEV_BUF_SUM_X, ws->sum_primal_solutions, n))) return rc;
  if ((rc = get(PDLPDEV_BUF_SUM_Y, own_rows(ws->sum_dual_solutions), ml))) return rc;
  if ((rc = get(PDLPDEV_BUF_LAST_RESTART_X, ws->last_restart_duality_gap_primal_solution, n))) return rc;
  if ((rc = get(PDLPDEV_BUF_LAST_RESTART_Y, own_rows(ws->last_restart_duality_gap_dual_solution), ml))) return rc;
  }
  DEV(pdlpdev_get_ctl(s->dev, &s->ctl));
  ws->initial_primal_weight         = s->ctl.primal_weight;
  ws->initial_step_size             = s->ctl.step_size;
  ws->total_pdlp_iterations         = s->total_iterations;
  ws->total_pdhg_iterations         = s->ctl.k;
  ws->last_candidate_kkt_score      = s->last_candidate_kkt;
  ws->last_restart_kkt_score        = s->last_restart_kkt;
  ws->sum_solution_weight           = s->ctl.sum_weights;
  ws->iterations_since_last_restart = s->ctl.its_since_restart;
  return 0;
}

int cuoptamd_solver_set_warm_start(cuoptamd_solver* s, const cuoptamd_warm_start* ws)
{
  if (!s || !ws || !s->dev) return fail(-1, "cuoptamd_solver_set_warm_start: null argument");
  if (s->started) return fail(-1, "cuoptamd_solver_set_warm_start: the solver has already been advanced");
  if ((ws->n_variables != 0 && ws->n_variables != s->n) || (ws->n_constraints != 0 && ws->n_constraints != s->m_global))
    return fail(-1, "cuoptamd_solver_set_warm_start: the snapshot belongs to a %d x %d problem, this one is %d x %d",
                ws->n_constraints, ws->n_variables, s->m_global, s->n);
  // iterate: unscaled in the snapshot -> scale_solutions (initial_scaling.cu:410-427), then the usual projection
  // (a sharded solver takes the FULL snapshot and keeps its own rows of the m-sized vectors)
  auto own_rows = [&](const double* v) -> const double* { return v ? v + s->row_begin : nullptr; };
  auto put = [&](int id, const double* src, int64_t count) -> int {
    if (!src) return 0;
    if (pdlpdev_upload(s->dev, id, src, count) != count) return fail(-2, "upload failed: %s", pdlpdev_last_error());
    return 0;
  };
  const int64_t n = s->n, ml = s->row_end - s->row_begin;
  // (a reordered LP: the snapshot is in the caller's order; every vector goes through the maps on its way in)
  std::vector<double> t[9];
  DEV(pdlpdev_set_initial(s->dev, s->cols_in(ws->current_primal_solution, t[0]), own_rows(s->rows_in(ws->current_dual_solution, t[1]))));
  if (s->H.project_initial_primal) DEV(pdlpdev_project_primal(s->dev));
  int rc;
  if ((rc = put(PDLPDEV_BUF_X, s->cols_in(ws->current_primal_solution_scaled, t[2]), n))) return rc;  // bit-exact iterate if given
  if ((rc = put(PDLPDEV_BUF_Y, own_rows(s->rows_in(ws->current_dual_solution_scaled, t[3])), ml))) return rc;
  if ((rc = put(PDLPDEV_BUF_ATY, s->cols_in(ws->current_ATY, t[4]), n))) return rc;
  if ((rc = put(PDLPDEV_BUF_SUM_X, s->cols_in(ws->sum_primal_solutions, t[5]), n))) return rc;
  if ((rc = put(PDLPDEV_BUF_SUM_Y, own_rows(s->rows_in(ws->sum_dual_solutions, t[6])), ml))) return rc;
  if ((rc = put(PDLPDEV_BUF_LAST_RESTART_X, s->cols_in(ws->last_restart_duality_gap_primal_solution, t[7]), n))) return rc;
  if ((rc = put(PDLPDEV_BUF_LAST_RESTART_Y, own_rows(s->rows_in(ws->last_restart_duality_gap_dual_solution, t[8])), ml))) return rc;
  DEV(pdlpdev_set_step(s->dev, ws->initial_step_size, ws->initial_primal_weight));
  DEV(pdlpdev_set_loop_state(s->dev, ws->sum_solution_weight, ws->iterations_since_last_restart, ws->total_pdhg_iterations));
  DEV(pdlpdev_get_ctl(s->dev, &s->ctl));
  s->iteration_offset   = ws->total_pdlp_iterations;
  s->total_iterations   = ws->total_pdlp_iterations;
  s->last_candidate_kkt = ws->last_candidate_kkt_score;
  s->last_restart_kkt   = ws->last_restart_kkt_score;
  // A^T y travels with the snapshot (pdlp.cu:160-163) -- and is trusted when the snapshot also carries the scaled iterate it
  // belongs to (the bit-exact resume).  A snapshot without it (the reference's nine vectors: y is re-scaled on restore; or one
  // carried to another problem by cuoptamd_warm_start_remap, whose A^T y is zero-padded / permuted) gets A^T y recomputed from
  // the restored y: one SpMV, and the step-size rule never sees an interaction term that does not vanish with the step
  // (which ends in an endless series of rejected steps).
  s->need_aty           = ws->current_ATY == nullptr || ws->current_dual_solution_scaled == nullptr;
  s->warm_started       = true;
  s->result.initial_step_size     = ws->initial_step_size;
  s->result.initial_primal_weight = ws->initial_primal_weight;
  return 0;
}

// pdlp_solver_settings_t::set_pdlp_warm_start_data with var_mapping / constraint_mapping (LP/solver_settings.cu:92-240),
// as its two tests pin it (unit_tests/solver_settings_test.cu:84-181 smaller, :183-280 bigger).  The reference scatters IN
// PLACE over the old vector's whole range with a map of the new, shorter length (it reads past the map); what the tests
// pin -- and what is restated here, out of place -- is  new[map[i]] = old[i]  for i < new size, then truncation; a longer
// map only says "pad with zeros"; an empty map or one of the same length leaves that side untouched.
int cuoptamd_warm_start_remap(const cuoptamd_warm_start* in, const int32_t* var_mapping, int32_t n_mapping,
                              const int32_t* constraint_mapping, int32_t m_mapping, cuoptamd_warm_start* out)
{
  if (!in || !out) return fail(-1, "cuoptamd_warm_start_remap: null argument");
  if (n_mapping < 0 || m_mapping < 0 || (n_mapping > 0 && !var_mapping) || (m_mapping > 0 && !constraint_mapping))
    return fail(-1, "cuoptamd_warm_start_remap: a mapping is missing");
  const int32_t n_old = in->n_variables, m_old = in->n_constraints;
  if (n_old <= 0 || m_old <= 0) return fail(-1, "cuoptamd_warm_start_remap: the snapshot does not carry its sizes");
  const int32_t n_new = n_mapping ? n_mapping : n_old, m_new = m_mapping ? m_mapping : m_old;
  auto check = [](const int32_t* map, int32_t len, int32_t old_len) {
    if (len >= old_len) return true;
    std::vector<char> seen((size_t)len, 0);
    for (int32_t i = 0; i < len; ++i) {
      if (map[i] < 0 || map[i] >= len || seen[(size_t)map[i]]) return false;
      seen[(size_t)map[i]] = 1;
    }
    return true;
  };
  if (!check(var_mapping, n_mapping, n_old) || !check(constraint_mapping, m_mapping, m_old))
    return fail(-1, "cuoptamd_warm_start_remap: a shrinking mapping must be a permutation of 0..new size-1");
  auto move = [](const double* src, double* dst, const int32_t* map, int32_t len_map, int32_t len_old, int32_t len_new) -> bool {
    if (!src) return dst == nullptr;
    if (!dst) return false;
    if (len_map != 0 && len_new < len_old) {
      for (int32_t i = 0; i < len_new; ++i) dst[map[i]] = src[i];
    } else {
      const int32_t keep = std::min(len_old, len_new);
      std::copy(src, src + keep, dst);
      std::fill(dst + keep, dst + len_new, 0.0);
    }
    return true;
  };
  bool ok = true;
  ok &= move(in->current_primal_solution, out->current_primal_solution, var_mapping, n_mapping, n_old, n_new);
  ok &= move(in->initial_primal_average, out->initial_primal_average, var_mapping, n_mapping, n_old, n_new);
  ok &= move(in->current_ATY, out->current_ATY, var_mapping, n_mapping, n_old, n_new);
  ok &= move(in->sum_primal_solutions, out->sum_primal_solutions, var_mapping, n_mapping, n_old, n_new);
  ok &= move(in->last_restart_duality_gap_primal_solution, out->last_restart_duality_gap_primal_solution, var_mapping, n_mapping, n_old, n_new);
  ok &= move(in->current_dual_solution, out->current_dual_solution, constraint_mapping, m_mapping, m_old, m_new);
  ok &= move(in->initial_dual_average, out->initial_dual_average, constraint_mapping, m_mapping, m_old, m_new);
  ok &= move(in->sum_dual_solutions, out->sum_dual_solutions, constraint_mapping, m_mapping, m_old, m_new);
  ok &= move(in->last_restart_duality_gap_dual_solution, out->last_restart_duality_gap_dual_solution, constraint_mapping, m_mapping, m_old, m_new);
  if (!ok) return fail(-1, "cuoptamd_warm_start_remap: the output snapshot must provide every vector the input has");
  // the scaled iterate belongs to the old problem's scaling: it does not survive a change of the problem
  if (n_new != n_old || m_new != m_old) {
    out->current_primal_solution_scaled = nullptr;
    out->current_dual_solution_scaled   = nullptr;
  } else {
    if (in->current_primal_solution_scaled && out->current_primal_solution_scaled)
      std::copy(in->current_primal_solution_scaled, in->current_primal_solution_scaled + n_old, out->current_primal_solution_scaled);
    else
      out->current_primal_solution_scaled = nullptr;
    if (in->current_dual_solution_scaled && out->current_dual_solution_scaled)
      std::copy(in->current_dual_solution_scaled, in->current_dual_solution_scaled + m_old, out->current_dual_solution_scaled);
    else
      out->current_dual_solution_scaled = nullptr;
  }
  out->initial_primal_weight         = in->initial_primal_weight;
  out->initial_step_size             = in->initial_step_size;
  out->total_pdlp_iterations         = in->total_pdlp_iterations;
  out->total_pdhg_iterations         = in->total_pdhg_iterations;
  out->last_candidate_kkt_score      = in->last_candidate_kkt_score;
  out->last_restart_kkt_score        = in->last_restart_kkt_score;
  out->sum_solution_weight           = in->sum_solution_weight;
  out->iterations_since_last_restart = in->iterations_since_last_restart;
  out->n_variables                   = n_new;
  out->n_constraints                 = m_new;
  return 0;
}

}  // extern "C"

// cuoptamd_batch_solve's path for LPs over one matrix and objective.  kNotShared: they are not (or the layouts are not the batch's):
// nothing was done, the caller solves them independently.
static constexpr int kNotShared = 12345;
static int shared_matrix_batch_solve(int32_t count, const cuoptamd_lp* lps, const cuoptamd_hyper* hyper, const cuoptamd_settings* settings, int device,
                                     cuoptamd_result* results, double** x, double** y, double** rc)
{
  const cuoptamd_lp& L0 = lps[0];
  if (L0.m <= 0 || L0.n <= 0 || !L0.offsets || !L0.lb || !L0.ub || !L0.lo || !L0.hi) return kNotShared;
  const size_t nnz = (size_t)L0.offsets[L0.m];
  for (int i = 1; i < count; ++i) {
    const cuoptamd_lp& L = lps[i];
    if (L.m != L0.m || L.n != L0.n || L.maximize != L0.maximize || L.objective_offset != L0.objective_offset) return kNotShared;
    auto same = [](const void* a, const void* b, size_t bytes) { return a == b || (a && b && memcmp(a, b, bytes) == 0); };
    if (!same(L.offsets, L0.offsets, ((size_t)L0.m + 1) * sizeof(int32_t)) || !same(L.indices, L0.indices, nnz * sizeof(int32_t)) ||
        !same(L.values, L0.values, nnz * sizeof(double)) || !same(L.c, L0.c, (size_t)L0.n * sizeof(double)))
      return kNotShared;
  }
  struct Slots {  // slot 0: the parent (created on LP 0), slots 1..15: clones of it -- destroyed before it
    std::vector<cuoptamd_solver*> s;
    ~Slots() { for (size_t i = s.size(); i-- > 0;) cuoptamd_solver_destroy(s[i]); }
  } slots;
  {
    cuoptamd_solver* parent = nullptr;
    int rc_ = cuoptamd_solver_create(&parent, &L0, hyper, settings, nullptr, nullptr, device, 0, 1, nullptr);
    if (parent) slots.s.push_back(parent);
    if (rc_ != 0) return rc_;
  }
  // the solver of slot q takes LP i: the parent as created (i = 0), a new clone, or an existing solver reset to LP i's bounds
  auto take = [&](int q, int i) -> int {
    const cuoptamd_lp& L = lps[i];
    if (i == 0) return 0;
    if (q < (int)slots.s.size()) return cuoptamd_solver_reset(slots.s[q], L.lb, L.ub, L.lo, L.hi, settings, nullptr, nullptr);
    cuoptamd_solver* s = nullptr;
    int rc_ = cuoptamd_solver_clone(slots.s[0], L.lb, L.ub, L.lo, L.hi, settings, &s);
    if (rc_ == 0) slots.s.push_back(s);
    return rc_;
  };
  auto solution = [&](int q, int i) { return cuoptamd_solver_get_solution(slots.s[q], x ? x[i] : nullptr, y ? y[i] : nullptr, rc ? rc[i] : nullptr); };
  int done = 0;
  bool lockstep = true;
  while (done < count) {
    const int left = count - done;
    const int K    = lockstep && left >= 16 ? 16 : lockstep && left >= 8 ? 8 : lockstep && left >= 4 ? 4 : 1;
    for (int q = 0; q < K; ++q) {
      int rc_ = take(q, done + q);
      if (rc_ == -7 && done == 0) return kNotShared;  // (a resident small-LP solver has no clones: nothing is lost, solve independently)
      if (rc_ != 0) return rc_;
    }
    cuoptamd_batch* b = nullptr;
    if (K > 1) {
      int rc_ = cuoptamd_batch_create(slots.s.data(), K, &b);
      if (rc_ == -7) lockstep = false;  // the layouts are not the batch's: the clones still save the set-ups, one LP after the other
      else if (rc_ != 0) return rc_;
    }
    if (b) {
      int rc_ = cuoptamd_batch_advance(b, std::numeric_limits<int32_t>::max(), &results[done]);
      cuoptamd_batch_destroy(b);
      if (rc_ != 0) return rc_;
    } else {
      for (int q = 0; q < K; ++q) {
        int rc_ = cuoptamd_solver_advance(slots.s[q], std::numeric_limits<int32_t>::max(), &results[done + q]);
        if (rc_ != 0) return rc_;
      }
    }
    for (int q = 0; q < K; ++q) {
      int rc_ = solution(q, done + q);
      if (rc_ != 0) return rc_;
    }
    done += K;
  }
  return 0;
}

// cuoptamd_batch_solve's path for LPs of resident size, whatever their matrices: every LP gets a solver (created on the worker threads,
// the solvers of one worker share a stream), then ALL of them advance as one cuoptamd_batch -- a workgroup per LP, one launch per
// phase of the loop (kernels_resident.hip) -- instead of one host thread per LP driving its own launches.  kNotShared: not this path.
static int small_lp_batch_solve(int32_t count, const cuoptamd_lp* lps, const cuoptamd_hyper* hyper, const cuoptamd_settings* settings, int device,
                                int max_threads, cuoptamd_result* results, double** x, double** y, double** rc)
{
  for (int i = 0; i < count; ++i) {
    const cuoptamd_lp& L = lps[i];
    if (L.m <= 0 || L.n <= 0 || !L.offsets || !pdlpdev_resident_size(L.m, L.n, L.offsets[L.m])) return kNotShared;
  }
  const int nt = std::max(1, std::min<int>(std::min<int>(count, 16), max_threads > 0 ? max_threads : cuopt_amd::host_threads()));
  std::vector<cuoptamd_solver*> sv(count, nullptr);
  std::vector<int> codes(count, 0);
  std::vector<std::string> messages(count);
  auto on_workers = [&](const std::function<void(int, int)>& body) {  // body(worker, LP) for LP = worker, worker + nt, ...
    std::vector<std::thread> pool;
    for (int w = 0; w < nt; ++w)
      pool.emplace_back([&, w] {
        for (int i = w; i < count; i += nt) body(w, i);
      });
    for (auto& t : pool) t.join();
  };
  auto destroy_all = [&]() {
    for (int i = count; i-- > 0;) cuoptamd_solver_destroy(sv[i]);  // (a worker's first solver lent its stream to the later ones)
  };
  auto first_error = [&]() -> int {
    for (int i = 0; i < count; ++i)
      if (codes[i] != 0) return fail(codes[i], "LP %d of the batch: %s", i, messages[i].c_str());
    return 0;
  };
  on_workers([&](int w, int i) {
    if (i != w && sv[w]) pdlpdev_create_share_stream(sv[w]->dev);
    codes[i] = cuoptamd_solver_create(&sv[i], &lps[i], hyper, settings, nullptr, nullptr, device, 0, 1, nullptr);
    pdlpdev_create_share_stream(nullptr);
    if (codes[i] != 0) messages[i] = cuoptamd_last_error();
  });
  int rc_ = first_error();
  if (rc_ == 0) {
    cuoptamd_batch* b = nullptr;
    rc_ = cuoptamd_batch_create(sv.data(), count, &b);
    if (rc_ == 0 && !b->small) rc_ = -7;  // (a lockstep batch of big LPs is shared_matrix_batch_solve's business)
    if (rc_ == 0) rc_ = cuoptamd_batch_advance(b, std::numeric_limits<int32_t>::max(), results);
    if (rc_ == 0) rc_ = cuoptamd_batch_get_solutions(b, x, y, rc);
    cuoptamd_batch_destroy(b);
    if (rc_ == -7) {  // e.g. CUOPT_AMD_SMALL=0: the solvers are not on the resident path
      destroy_all();
      return kNotShared;
    }
  }
  destroy_all();
  return rc_;
}

extern "C" {

int cuoptamd_batch_solve(int32_t count, const cuoptamd_lp* lps, const cuoptamd_hyper* hyper,
                         const cuoptamd_settings* settings, int device, int max_threads,
                         cuoptamd_result* results, double** x, double** y, double** rc)
{
  if (count < 0 || (count > 0 && (!lps || !hyper || !settings || !results)))
    return fail(-1, "cuoptamd_batch_solve: null argument");
  // small LPs (the size of MIP relaxations): all of them at once, a workgroup each (round 6)
  if (count >= 2 && cuopt_amd::tune_int("small_batch", 1) != 0) {
    int rc_ = small_lp_batch_solve(count, lps, hyper, settings, device, max_threads, results, x, y, rc);
    if (rc_ != kNotShared) return rc_;
  }
  // LPs that share matrix and objective (the MIP heuristics' re-solves: the same A and c under other bounds) go through ONE set-up and
  // advance in lockstep, sixteen, eight or four at a time (cuoptamd_batch_*): each gets, bit for bit, the answer of its own solve
  if (count >= 4 && cuopt_amd::tune_int("shared_batch", 1) != 0) {
    int rc_ = shared_matrix_batch_solve(count, lps, hyper, settings, device, results, x, y, rc);
    if (rc_ != kNotShared) return rc_;
  }
  const int nt = std::max(1, std::min<int>(count, max_threads > 0 ? max_threads : cuopt_amd::host_threads()));
  std::vector<int> codes(count, 0);
  std::vector<std::string> messages(count);
  std::vector<std::thread> pool;
  for (int w = 0; w < nt; ++w)
    pool.emplace_back([&, w] {
      for (int i = w; i < count; i += nt) {
        cuoptamd_solver* s = nullptr;
        int rc_ = cuoptamd_solver_create(&s, &lps[i], hyper, settings, nullptr, nullptr, device, 0, 1, nullptr);
        if (rc_ == 0) rc_ = cuoptamd_solver_advance(s, std::numeric_limits<int32_t>::max(), &results[i]);
        if (rc_ == 0) rc_ = cuoptamd_solver_get_solution(s, x ? x[i] : nullptr, y ? y[i] : nullptr, rc ? rc[i] : nullptr);
        if (rc_ != 0) messages[i] = cuoptamd_last_error();  // thread-local message of this worker
        codes[i] = rc_;
        cuoptamd_solver_destroy(s);
      }
    });
  for (auto& t : pool) t.join();
  for (int i = 0; i < count; ++i)
    if (codes[i] != 0) return fail(codes[i], "LP %d of the batch: %s", i, messages[i].c_str());
  return 0;
}

pdlpdev_ctx* cuoptamd_solver_device(cuoptamd_solver* s) { return s ? s->dev : nullptr; }

int cuoptamd_solver_reorder_info(cuoptamd_solver* s, int32_t info[10], int32_t* row_new2old, int32_t* col_new2old)
{
  if (!s) return fail(-1, "cuoptamd_solver_reorder_info: null solver");
  if (info) std::copy(s->analysis_info, s->analysis_info + 10, info);
  if (s->row_new2old.empty()) return 0;
  if (row_new2old) std::copy(s->row_new2old.begin(), s->row_new2old.end(), row_new2old);
  if (col_new2old) std::copy(s->col_new2old.begin(), s->col_new2old.end(), col_new2old);
  return 1;
}

int cuoptamd_solver_row_range(cuoptamd_solver* s, int32_t* row_begin, int32_t* row_end)
{
  if (!s) return fail(-1, "null solver");
  if (row_begin) *row_begin = s->row_begin;
  if (row_end) *row_end = s->row_end;
  return 0;
}

}  // extern "C"
