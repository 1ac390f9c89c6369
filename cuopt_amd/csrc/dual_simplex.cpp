// A bounded dual simplex for SMALL LPs: this library's second engine behind CUOPT_METHOD_DUAL_SIMPLEX and the
// Concurrent method (the reference runs its CPU dual simplex there: LP/solve.cu:295-347 run_dual_simplex, :383-443
// run_concurrent; cpp/src/dual_simplex/).  Own implementation, nothing of the reference's simplex is linked or restated:
// the textbook bounded dual simplex (Dantzig pricing on the primal infeasibilities, Harris' two-pass dual ratio test) on
//     min c.x   s.t.  A x - s = 0,   lb <= x <= ub,   lo <= s <= hi
// with a DENSE explicit basis inverse (rank-one updates, refactorisation from scratch every 400 pivots), which is what
// an LP of a few thousand rows needs and no more.  Infinite bounds are boxed (+-BIG) so that the slack basis is dual
// feasible from the start; a solution that leans on a box bound is solved again with a 1000 times larger box, and if
// it still does, the LP is unbounded.  Host code: the reference's simplex is CPU code too; PDLP on the GPU stays the
// engine for everything that is not small.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "cuopt_amd/pdlp_solver.h"

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

struct Simplex {
  int m = 0, n = 0, N = 0;
  // columns of A (CSC) -- column j of M = [A | -I] is A's column j for j < n, -e_(j-n) behind
  std::vector<int32_t> cp, ci;
  std::vector<double> cv;
  std::vector<double> g, L, U;      // cost, bounds of z = (x, s)
  std::vector<char> boxedL, boxedU; // the bound is an artificial box bound
  std::vector<int> basic, pos;      // basic[r] = variable, pos[j] = row or -1
  std::vector<char> atU;            // nonbasic at its upper bound
  std::vector<double> z, d, y;      // primal values, reduced costs, duals
  std::vector<double> Binv;         // m x m, row major
  int iterations = 0;

  double col_dot(const double* row, int j) const  // row . M_j
  {
    if (j >= n) return -row[j - n];
    double s = 0.0;
    for (int k = cp[j]; k < cp[j + 1]; ++k) s += row[ci[k]] * cv[k];
    return s;
  }
  // Binv from scratch (Gauss-Jordan with partial pivoting on the basis matrix); false: singular
  bool refactor()
  {
    std::vector<double> Bm((size_t)m * m, 0.0);
    for (int r = 0; r < m; ++r) {
      const int j = basic[r];
      if (j >= n) Bm[(size_t)(j - n) * m + r] = -1.0;
      else
        for (int k = cp[j]; k < cp[j + 1]; ++k) Bm[(size_t)ci[k] * m + r] = cv[k];
    }
    Binv.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) Binv[(size_t)i * m + i] = 1.0;
    for (int c = 0; c < m; ++c) {
      int piv = c;
      double best = std::fabs(Bm[(size_t)c * m + c]);
      for (int i = c + 1; i < m; ++i)
        if (std::fabs(Bm[(size_t)i * m + c]) > best) best = std::fabs(Bm[(size_t)i * m + c]), piv = i;
      if (best < 1e-11) return false;
      if (piv != c) {
        for (int k = 0; k < m; ++k) std::swap(Bm[(size_t)piv * m + k], Bm[(size_t)c * m + k]), std::swap(Binv[(size_t)piv * m + k], Binv[(size_t)c * m + k]);
      }
      const double inv = 1.0 / Bm[(size_t)c * m + c];
      for (int k = 0; k < m; ++k) Bm[(size_t)c * m + k] *= inv, Binv[(size_t)c * m + k] *= inv;
      for (int i = 0; i < m; ++i) {
        if (i == c) continue;
        const double f = Bm[(size_t)i * m + c];
        if (f == 0.0) continue;
        double* bi       = &Bm[(size_t)i * m];
        double* vi       = &Binv[(size_t)i * m];
        const double* bc = &Bm[(size_t)c * m];
        const double* vc = &Binv[(size_t)c * m];
        for (int k = 0; k < m; ++k) bi[k] -= f * bc[k], vi[k] -= f * vc[k];
      }
    }
    return true;
  }
  // z_B, y, d from the nonbasic values and the current inverse
  void recompute()
  {
    std::vector<double> rhs(m, 0.0);  // -N z_N
    for (int j = 0; j < N; ++j) {
      if (pos[j] >= 0) continue;
      const double v = z[j];
      if (v == 0.0) continue;
      if (j >= n) rhs[j - n] += v;
      else
        for (int k = cp[j]; k < cp[j + 1]; ++k) rhs[ci[k]] -= cv[k] * v;
    }
    for (int r = 0; r < m; ++r) {
      double s         = 0.0;
      const double* br = &Binv[(size_t)r * m];
      for (int i = 0; i < m; ++i) s += br[i] * rhs[i];
      z[basic[r]] = s;
    }
    std::fill(y.begin(), y.end(), 0.0);
    for (int r = 0; r < m; ++r) {
      const double gb = g[basic[r]];
      if (gb == 0.0) continue;
      const double* br = &Binv[(size_t)r * m];
      for (int i = 0; i < m; ++i) y[i] += gb * br[i];
    }
    for (int j = 0; j < N; ++j) d[j] = pos[j] >= 0 ? 0.0 : g[j] - col_dot(y.data(), j);
  }
};

// status: 1 optimal, 2 primal infeasible, 5 iteration limit, 6 time limit, 7 numerical trouble
int run(Simplex& S, int iteration_limit, double time_limit, const std::chrono::steady_clock::time_point& t0, const volatile int32_t* cancel)
{
  const int m = S.m, n = S.n, N = S.N;
  // slack basis: B = -I, dual feasible by the choice of the nonbasic bounds (every bound is finite after boxing)
  S.basic.resize(m), S.pos.assign(N, -1), S.atU.assign(N, 0);
  for (int r = 0; r < m; ++r) S.basic[r] = n + r, S.pos[n + r] = r;
  S.z.assign(N, 0.0), S.d.assign(N, 0.0), S.y.assign(m, 0.0);
  for (int j = 0; j < n; ++j) {
    S.atU[j] = S.g[j] < 0.0;
    S.z[j]   = S.atU[j] ? S.U[j] : S.L[j];
  }
  S.Binv.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) S.Binv[(size_t)i * m + i] = -1.0;
  S.recompute();
  std::vector<double> alpha(N), w(m);
  std::vector<int> passed(m, 0);  // iteration (+1) at which the row was passed over as "violated by rounding only"
  const double tol_d = 1e-9;
  int since_refactor = 0;
  for (;;) {
    if (S.iterations >= iteration_limit) return 5;
    if (cancel && *cancel) return 9;  // the other engine of a Concurrent solve has finished
    if ((S.iterations & 63) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_limit) return 6;
    // leaving row: the largest primal infeasibility (primal tolerance 1e-7 relative to the bound; the reference's simplex: 1e-6
    // absolute).  A row whose violation is within 1e-6 and that has no entering candidate is rounding, not a proof of
    // infeasibility (the box bounds put values of 1e6 into the basis): it is passed over.
    int r = -1;
    double worst = 0.0;
    for (int i = 0; i < m; ++i) {
      if (passed[i] == S.iterations + 1) continue;
      const int b     = S.basic[i];
      const double v  = S.z[b];
      const double lo = S.L[b] - v, up = v - S.U[b];
      const double inf = std::max(lo, up);
      const double tol = 1e-7 * (1.0 + std::fabs(lo > up ? S.L[b] : S.U[b]));
      if (inf > tol && inf > worst) worst = inf, r = i;
    }
    if (r < 0) return 1;
    const int p        = S.basic[r];
    const bool to_low  = S.z[p] < S.L[p];
    const double delta = to_low ? S.L[p] - S.z[p] : S.U[p] - S.z[p];  // change the leaving variable needs
    const double sigma = to_low ? 1.0 : -1.0;
    // row r of the tableau
    const double* br = &S.Binv[(size_t)r * m];
    double amax      = 0.0;
    for (int j = 0; j < N; ++j) {
      alpha[j] = 0.0;
      if (S.pos[j] >= 0 || S.L[j] == S.U[j]) continue;
      alpha[j] = S.col_dot(br, j);
      amax     = std::max(amax, std::fabs(alpha[j]));
    }
    const double ptol = std::max(1e-11, 1e-9 * amax);
    // Harris: pass 1 the largest step that keeps every reduced cost within tol_d of its sign, pass 2 the largest pivot under it
    double tmax = kInf;
    for (int j = 0; j < N; ++j) {
      const double a = sigma * alpha[j];
      if (S.pos[j] >= 0 || std::fabs(a) <= ptol) continue;
      const bool eligible = S.atU[j] ? a > 0.0 : a < 0.0;
      if (!eligible) continue;
      tmax = std::min(tmax, (std::fabs(S.d[j]) + tol_d) / std::fabs(a));
    }
    if (tmax == kInf && worst <= 1e-6 * (1.0 + std::fabs(to_low ? S.L[p] : S.U[p]))) {
      passed[r] = S.iterations + 1;
      continue;
    }
    if (tmax == kInf) {  // no entering variable: the row proves primal infeasibility
      if (std::getenv("CUOPT_AMD_SIMPLEX_DEBUG")) {
        std::fprintf(stderr, "[simplex] infeasible row %d var %d value %.12g bounds [%.6g, %.6g] amax %.3g ptol %.3g; nonbasic:\n", r, p, S.z[p], S.L[p], S.U[p], amax, ptol);
        for (int j = 0; j < N; ++j)
          if (S.pos[j] < 0 && alpha[j] != 0.0) std::fprintf(stderr, "   j %d alpha %.3g atU %d z %.6g [%.6g, %.6g] d %.3g\n", j, alpha[j], (int)S.atU[j], S.z[j], S.L[j], S.U[j], S.d[j]);
      }
      return 2;
    }
    int q        = -1;
    double apick = 0.0;
    for (int j = 0; j < N; ++j) {
      const double a = sigma * alpha[j];
      if (S.pos[j] >= 0 || std::fabs(a) <= ptol) continue;
      const bool eligible = S.atU[j] ? a > 0.0 : a < 0.0;
      if (!eligible) continue;
      if (std::fabs(S.d[j]) / std::fabs(a) <= tmax && std::fabs(a) > apick) apick = std::fabs(a), q = j;
    }
    if (q < 0) return 7;
    // entering column
    for (int i = 0; i < m; ++i) {
      const double* bi = &S.Binv[(size_t)i * m];
      double s         = 0.0;
      if (q >= n) s = -bi[q - n];
      else
        for (int k = S.cp[q]; k < S.cp[q + 1]; ++k) s += bi[S.ci[k]] * S.cv[k];
      w[i] = s;
    }
    if (std::fabs(w[r]) < 1e-11 || std::fabs(w[r] - alpha[q]) > 1e-6 * (1.0 + std::fabs(alpha[q]))) {
      // the inverse has drifted: rebuild it and look again
      if (since_refactor == 0 || !S.refactor()) return 7;
      S.recompute();
      since_refactor = 0;
      continue;
    }
    // duals: d_j -= theta alpha_rj, the entering variable's becomes 0, the leaving one's -theta
    const double theta = S.d[q] / alpha[q];
    for (int j = 0; j < N; ++j)
      if (S.pos[j] < 0 && alpha[j] != 0.0) S.d[j] -= theta * alpha[j];
    S.d[q] = 0.0;
    S.d[p] = -theta;
    // primal: the entering variable moves by tau, the basic ones by -w tau
    const double tau = -delta / w[r];
    for (int i = 0; i < m; ++i) S.z[S.basic[i]] -= w[i] * tau;
    S.z[q] += tau;
    S.z[p]   = to_low ? S.L[p] : S.U[p];
    S.atU[p] = !to_low;
    S.pos[p] = -1, S.pos[q] = r, S.basic[r] = q;
    // inverse: row r scaled, the others eliminated
    {
      double* rr       = &S.Binv[(size_t)r * m];
      const double inv = 1.0 / w[r];
      for (int k = 0; k < m; ++k) rr[k] *= inv;
      for (int i = 0; i < m; ++i) {
        if (i == r || w[i] == 0.0) continue;
        double* bi     = &S.Binv[(size_t)i * m];
        const double f = w[i];
        for (int k = 0; k < m; ++k) bi[k] -= f * rr[k];
      }
    }
    S.iterations += 1;
    if (++since_refactor >= 400) {  // (a refactorisation is O(m^3), a pivot O(m^2))
      if (!S.refactor()) return 7;
      S.recompute();
      since_refactor = 0;
      // a reduced cost that drifted to the wrong side of zero: put the variable on the bound that fits (boxed: always possible)
      for (int j = 0; j < N; ++j) {
        if (S.pos[j] >= 0 || S.L[j] == S.U[j]) continue;
        const bool want_upper = S.d[j] < -tol_d;
        const bool want_lower = S.d[j] > tol_d;
        if ((want_upper && !S.atU[j]) || (want_lower && S.atU[j])) {
          S.atU[j] = want_upper;
          S.z[j]   = S.atU[j] ? S.U[j] : S.L[j];
        }
      }
      S.recompute();
    }
  }
}

}  // namespace

extern "C" int cuoptamd_dual_simplex(const cuoptamd_lp* lp, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
                                     int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc)
{
  if (!lp || !status) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  const int m = lp->m, n = lp->n;
  const int64_t nnz = m > 0 ? lp->offsets[m] : 0;
  *status = 8;  // too large for a dense basis inverse (or empty): the caller keeps to PDLP
  if (iterations) *iterations = 0;
  if (m <= 0 || n <= 0 || m > 3000 || (int64_t)n + m > 60000 || nnz > 400000) return 0;
  Simplex S;
  S.m = m, S.n = n, S.N = n + m;
  // columns of A
  S.cp.assign(n + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) S.cp[lp->indices[k] + 1]++;
  for (int j = 0; j < n; ++j) S.cp[j + 1] += S.cp[j];
  S.ci.resize((size_t)nnz), S.cv.resize((size_t)nnz);
  {
    std::vector<int32_t> cur(S.cp.begin(), S.cp.end() - 1);
    for (int i = 0; i < m; ++i)
      for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k) {
        const int q = cur[lp->indices[k]]++;
        S.ci[q] = i, S.cv[q] = lp->values[k];
      }
  }
  const double sense = lp->maximize ? -1.0 : 1.0;
  double scale = 1.0;
  for (int j = 0; j < n; ++j) {
    if (std::isfinite(lp->lb[j])) scale = std::max(scale, std::fabs(lp->lb[j]));
    if (std::isfinite(lp->ub[j])) scale = std::max(scale, std::fabs(lp->ub[j]));
  }
  for (int i = 0; i < m; ++i) {
    if (std::isfinite(lp->lo[i])) scale = std::max(scale, std::fabs(lp->lo[i]));
    if (std::isfinite(lp->hi[i])) scale = std::max(scale, std::fabs(lp->hi[i]));
  }
  double prev_obj = 0.0;
  int code        = 7;
  std::vector<double> first_z, first_y, first_d;  // the vertex of the first (smaller) box, kept while the second one is tried
  std::vector<int> first_pos;
  if (time_limit <= 0.0 || !std::isfinite(time_limit)) time_limit = 1e30;
  if (iteration_limit <= 0) iteration_limit = std::numeric_limits<int32_t>::max();
  for (int attempt = 0; attempt < 2; ++attempt) {
    const double big = (attempt == 0 ? 1e5 : 1e8) * scale;
    S.g.assign(S.N, 0.0), S.L.assign(S.N, 0.0), S.U.assign(S.N, 0.0);
    S.boxedL.assign(S.N, 0), S.boxedU.assign(S.N, 0);
    for (int j = 0; j < S.N; ++j) {
      const double l = j < n ? lp->lb[j] : lp->lo[j - n], u = j < n ? lp->ub[j] : lp->hi[j - n];
      if (l > u) {
        *status = 2;  // contradictory bounds
        return 0;
      }
      S.g[j] = j < n ? sense * lp->c[j] : 0.0;
      S.L[j] = std::isfinite(l) ? l : -big, S.boxedL[j] = !std::isfinite(l);
      S.U[j] = std::isfinite(u) ? u : big, S.boxedU[j] = !std::isfinite(u);
    }
    S.iterations = 0;
    code         = run(S, iteration_limit, time_limit, t0, cancel);
    if (iterations) *iterations += S.iterations;
    if (code != 1) break;
    if (!S.refactor()) {  // the numbers that go out come from a fresh inverse
      code = 7;
      break;
    }
    S.recompute();
    // does the vertex lean on a box bound?
    bool leans = false;
    for (int j = 0; j < S.N && !leans; ++j) {
      const double tol = 1e-6 * big;
      leans = (S.boxedL[j] && S.z[j] <= S.L[j] + tol) || (S.boxedU[j] && S.z[j] >= S.U[j] - tol);
    }
    double obj = 0.0;
    for (int j = 0; j < n; ++j) obj += S.g[j] * S.z[j];
    if (std::getenv("CUOPT_AMD_SIMPLEX_DEBUG")) {
      std::fprintf(stderr, "[simplex] attempt %d box %.3g: objective %.17g, leans %d, iterations %d, x =", attempt, big, obj, (int)leans, S.iterations);
      for (int j = 0; j < std::min(n, 8); ++j) std::fprintf(stderr, " %.6g", S.z[j]);
      std::fprintf(stderr, "\n");
    }
    if (!leans) break;
    if (attempt == 1) {
      // still on the (1000 times wider) box: how fast did the objective follow it out, per unit of box and of cost?
      //   clearly (> 1e-4): unbounded;  not at all (< 1e-12): a ray of alternative optima, the first, smaller vertex is the answer;
      //   in between the ray's cost is inside any dual tolerance (datasets/mip/minrep_inf.mps: 2.5e-7 -- the reference's simplex
      //   calls that LP optimal, HiGHS unbounded): this engine abstains (numerical trouble) and the caller's PDLP answers
      double cmax = 0.0;
      for (int j = 0; j < n; ++j) cmax = std::max(cmax, std::fabs(S.g[j]));
      const double rate = (prev_obj - obj) / (big * std::max(cmax, 1e-300));
      if (rate > 1e-4) {
        code = 3;
      } else if (rate < 1e-12) {
        S.z = first_z, S.y = first_y, S.d = first_d, S.pos = first_pos;
      } else {
        code = 7;
      }
      break;
    }
    prev_obj = obj;
    first_z = S.z, first_y = S.y, first_d = S.d, first_pos = S.pos;
  }
  *status = code;
  if (code == 1) {
    double obj = 0.0;
    for (int j = 0; j < n; ++j) obj += lp->c[j] * S.z[j];
    if (objective) *objective = obj + lp->objective_offset;
    if (x) std::copy(S.z.begin(), S.z.begin() + n, x);
    // duals / reduced costs of the user's problem (a maximisation was solved as the minimisation of -c)
    if (y)
      for (int i = 0; i < m; ++i) y[i] = sense * S.y[i];
    if (rc)
      for (int j = 0; j < n; ++j) rc[j] = sense * (S.pos[j] >= 0 ? 0.0 : S.d[j]);
  }
  return 0;
}
