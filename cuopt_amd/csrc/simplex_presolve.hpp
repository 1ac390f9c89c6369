// Presolve in front of the simplex engine (dual_simplex.cpp) and the way back.  What the reference's simplex removes before it
// pivots (cpp/src/dual_simplex/presolve.cpp:26-212,585-662: empty columns, empty rows, fixed variables) plus singleton rows, which
// are bounds in disguise -- own code on this engine's form
//     min g.x   s.t.  lo <= A x <= hi,   lb <= x <= ub            (g = c, or -c for a maximisation)
// Reductions, repeated until none applies:
//   * a row without entries left: feasible if 0 is within its (shifted) bounds, else the LP is infeasible; its dual is 0;
//   * a row with ONE entry a x_j left: lo / a <= x_j <= hi / a (swapped for a < 0) tightens the column's bounds, the row goes;
//   * a column whose bounds have met: x_j = v, every row it is in is shifted by -a v;
//   * a column without entries left: it sits on the bound its cost points to (kept for the engine when that bound is infinite).
// The way back runs the removals in reverse.  A restored fixed / empty column gets z_j = g_j - sum_i a_ij y_i over the rows that
// have their dual by then; a restored singleton row takes over the part of its column's reduced cost that the column's bounds
// BEFORE the row tightened them cannot carry (x_j strictly inside them, or z_j of the wrong sign for the one x_j sits on):
// y_i = z_j / a, z_j = 0 -- the bound was the row's, so is the multiplier.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "cuopt_amd/pdlp_solver.h"

namespace cuopt_amd {

struct SimplexPresolve {
  enum Kind : uint8_t { kEmptyRow, kSingletonRow, kFixedColumn, kEmptyColumn };
  struct Step {
    Kind kind;
    int row, col;
    double a;         // the singleton row's entry
    double lbp, ubp;  // the column's bounds before a singleton row tightened them
    double value;     // a removed column's value
  };
  // the reduced LP (valid while this object lives)
  std::vector<int32_t> off, idx;
  std::vector<double> val, c, lo, hi, lb, ub;
  std::vector<int> rows, cols;  // reduced -> original numbers
  cuoptamd_lp reduced{};
  std::vector<Step> steps;
  bool infeasible = false;
  int removed_rows = 0, removed_cols = 0;

  static bool finite(double v) { return std::isfinite(v); }

  bool cancelled = false;
  // false: nothing to remove (use the LP as it is).  `cancel`: the flag of a Concurrent solve's other engine -- looked at while the
  // matrix is walked (run() returns true with `cancelled` set)
  bool run(const cuoptamd_lp* lp, const volatile int32_t* cancel = nullptr)
  {
    auto stop = [&] { return cancel && __atomic_load_n(const_cast<const int32_t*>(cancel), __ATOMIC_ACQUIRE) != 0 ? (cancelled = true) : false; };
    const int m = lp->m, n = lp->n;
    const double sense = lp->maximize ? -1.0 : 1.0;
    const int64_t nnz = lp->offsets[m];
    // columns of A
    std::vector<int32_t> cp(n + 1, 0), ci((size_t)nnz);
    std::vector<double> cv((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
      if ((k & 0xFFFF) == 0 && stop()) return true;
      cp[lp->indices[k] + 1]++;
    }
    for (int j = 0; j < n; ++j) cp[j + 1] += cp[j];
    {
      std::vector<int32_t> cur(cp.begin(), cp.end() - 1);
      for (int i = 0; i < m; ++i) {
        if ((i & 0xFFF) == 0 && stop()) return true;
        for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k) {
          const int q = cur[lp->indices[k]]++;
          ci[q] = i, cv[q] = lp->values[k];
        }
      }
    }
    std::vector<double> rlo(lp->lo, lp->lo + m), rhi(lp->hi, lp->hi + m), xl(lp->lb, lp->lb + n), xu(lp->ub, lp->ub + n);
    std::vector<int> rcount(m, 0), ccount(n, 0);
    std::vector<char> ralive(m, 1), calive(n, 1);
    for (int i = 0; i < m; ++i) {
      if ((i & 0xFFF) == 0 && stop()) return true;
      for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k)
        if (lp->values[k] != 0.0) rcount[i]++, ccount[lp->indices[k]]++;
    }
    std::vector<int> rq, cq;  // rows / columns to look at
    for (int i = 0; i < m; ++i)
      if (rcount[i] <= 1) rq.push_back(i);
    for (int j = 0; j < n; ++j)
      if (ccount[j] == 0 || xl[j] == xu[j]) cq.push_back(j);
    auto crossed = [](double l, double u) { return l > u + 1e-9 * (1.0 + std::fabs(l) + std::fabs(u)); };
    // a column (or row) whose OWN bounds cross is a verdict before any reduction: an empty column would otherwise be removed at
    // the bound its cost points to and the LP would come back Optimal with x outside its bounds
    for (int j = 0; j < n; ++j)
      if (crossed(xl[j], xu[j])) {
        infeasible = true;
        return true;
      }
    for (int i = 0; i < m; ++i)
      if (crossed(rlo[i], rhi[i])) {
        infeasible = true;
        return true;
      }
    while (!rq.empty() || !cq.empty()) {
      while (!rq.empty()) {
        const int i = rq.back();
        rq.pop_back();
        if (!ralive[i] || rcount[i] > 1) continue;
        if (rcount[i] == 0) {
          if (crossed(rlo[i], 0.0) || crossed(0.0, rhi[i])) {
            infeasible = true;
            return true;
          }
          ralive[i] = 0, ++removed_rows;
          steps.push_back({kEmptyRow, i, -1, 0.0, 0.0, 0.0, 0.0});
          continue;
        }
        int j    = -1;
        double a = 0.0;
        for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k)
          if (lp->values[k] != 0.0 && calive[lp->indices[k]]) j = lp->indices[k], a += lp->values[k];  // (+=: a file may list an entry twice)
        if (j < 0 || a == 0.0 || std::fabs(a) < 1e-9) continue;  // (left to the engine: a tiny entry makes a poor bound)
        double l = rlo[i] / a, u = rhi[i] / a;
        if (a < 0.0) std::swap(l, u);
        steps.push_back({kSingletonRow, i, j, a, xl[j], xu[j], 0.0});
        if (l > xl[j]) xl[j] = l;
        if (u < xu[j]) xu[j] = u;
        if (crossed(xl[j], xu[j])) {
          infeasible = true;
          return true;
        }
        if (xl[j] > xu[j]) xl[j] = xu[j] = 0.5 * (xl[j] + xu[j]);  // (crossed within the tolerance)
        ralive[i] = 0, ++removed_rows;
        if (--ccount[j] == 0 || xl[j] == xu[j]) cq.push_back(j);
      }
      while (!cq.empty()) {
        const int j = cq.back();
        cq.pop_back();
        if (!calive[j]) continue;
        if (crossed(xl[j], xu[j])) {  // (bounds tightened by singleton rows since the column was queued)
          infeasible = true;
          return true;
        }
        double v;
        Kind kind;
        if (xl[j] == xu[j]) {
          v = xl[j], kind = kFixedColumn;
          if (!finite(v)) continue;
        } else if (ccount[j] == 0) {
          const double g = sense * lp->c[j];
          v    = g > 0.0 ? xl[j] : g < 0.0 ? xu[j] : finite(xl[j]) ? xl[j] : finite(xu[j]) ? xu[j] : 0.0;
          kind = kEmptyColumn;
          if (!finite(v)) continue;  // the cost points to an infinite bound: the engine decides between unbounded and infeasible
        } else {
          continue;
        }
        calive[j] = 0, ++removed_cols;
        steps.push_back({kind, -1, j, 0.0, 0.0, 0.0, v});
        for (int e = cp[j]; e < cp[j + 1]; ++e) {
          const int i = ci[e];
          if (!ralive[i] || cv[e] == 0.0) continue;
          if (v != 0.0) rlo[i] -= cv[e] * v, rhi[i] -= cv[e] * v;
          if (--rcount[i] <= 1) rq.push_back(i);
        }
      }
    }
    if (removed_rows == 0 && removed_cols == 0) return false;
    // the reduced LP
    std::vector<int> cnew(n, -1);
    for (int j = 0; j < n; ++j)
      if (calive[j]) cnew[j] = (int)cols.size(), cols.push_back(j);
    off.assign(1, 0);
    for (int i = 0; i < m; ++i) {
      if ((i & 0xFFF) == 0 && stop()) return true;
      if (!ralive[i]) continue;
      rows.push_back(i);
      for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k)
        if (cnew[lp->indices[k]] >= 0) idx.push_back(cnew[lp->indices[k]]), val.push_back(lp->values[k]);
      off.push_back((int32_t)idx.size());
      lo.push_back(rlo[i]), hi.push_back(rhi[i]);
    }
    for (int j : cols) c.push_back(lp->c[j]), lb.push_back(xl[j]), ub.push_back(xu[j]);
    reduced.m = (int32_t)rows.size(), reduced.n = (int32_t)cols.size();
    reduced.offsets = off.data(), reduced.indices = idx.data(), reduced.values = val.data();
    reduced.c = c.data(), reduced.lo = lo.data(), reduced.hi = hi.data(), reduced.lb = lb.data(), reduced.ub = ub.data();
    reduced.maximize = lp->maximize, reduced.objective_offset = 0.0;
    return true;
  }

  // x, y, rc hold the reduced LP's solution in their first reduced.n / reduced.m entries (duals of the converted minimisation);
  // on return the original LP's (arrays of the original sizes).  Returns the objective c.x + offset.
  double undo(const cuoptamd_lp* lp, double* x, double* y, double* rc) const
  {
    const int m = lp->m, n = lp->n;
    const double sense = lp->maximize ? -1.0 : 1.0;
    std::vector<double> xr(x, x + reduced.n), yr(y, y + reduced.m), zr(rc, rc + reduced.n);
    std::fill(x, x + n, 0.0), std::fill(y, y + m, 0.0), std::fill(rc, rc + n, 0.0);
    for (size_t q = 0; q < cols.size(); ++q) x[cols[q]] = xr[q], rc[cols[q]] = zr[q];
    for (size_t q = 0; q < rows.size(); ++q) y[rows[q]] = yr[q];
    // (columns of A again: the reduced costs of the restored columns need them)
    const int64_t nnz = lp->offsets[m];
    std::vector<int32_t> cp(n + 1, 0), ci((size_t)nnz);
    std::vector<double> cv((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) cp[lp->indices[k] + 1]++;
    for (int j = 0; j < n; ++j) cp[j + 1] += cp[j];
    {
      std::vector<int32_t> cur(cp.begin(), cp.end() - 1);
      for (int i = 0; i < m; ++i)
        for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k) {
          const int q = cur[lp->indices[k]]++;
          ci[q] = i, cv[q] = lp->values[k];
        }
    }
    for (size_t s = steps.size(); s-- > 0;) {
      const Step& st = steps[s];
      switch (st.kind) {
        case kEmptyRow: y[st.row] = 0.0; break;
        case kFixedColumn:
        case kEmptyColumn: {
          x[st.col] = st.value;
          double z  = sense * lp->c[st.col];
          for (int e = cp[st.col]; e < cp[st.col + 1]; ++e) z -= cv[e] * y[ci[e]];  // (rows not restored yet still hold 0)
          rc[st.col] = z;
          break;
        }
        case kSingletonRow: {
          const int j    = st.col;
          const double z = rc[j], v = x[j], tol = 1e-9 * (1.0 + std::fabs(v));
          const bool at_low = finite(st.lbp) && v <= st.lbp + tol, at_up = finite(st.ubp) && v >= st.ubp - tol;
          const bool carried = z > 0.0 ? at_low : z < 0.0 ? at_up : true;  // can the column's earlier bounds hold this reduced cost?
          if (!carried) y[st.row] = z / st.a, rc[j] = 0.0;
          else y[st.row] = 0.0;
          break;
        }
      }
    }
    double obj = lp->objective_offset;
    for (int j = 0; j < n; ++j) obj += lp->c[j] * x[j];
    return obj;
  }
};

}  // namespace cuopt_amd
