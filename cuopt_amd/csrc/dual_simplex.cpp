// A bounded dual simplex: this library's second engine behind CUOPT_METHOD_DUAL_SIMPLEX, the Concurrent method and
// crossover requests (the reference runs its CPU dual simplex there: LP/solve.cu:295-347 run_dual_simplex, :383-443
// run_concurrent; cpp/src/dual_simplex/).  Own implementation, nothing of the reference's simplex is linked or restated:
// the textbook bounded dual simplex (dual steepest-edge pricing on the primal infeasibilities, Harris' two-pass dual ratio
// test as a long-step rule: groups of breakpoints the row's infeasibility outlasts are flipped to their other bounds) on
//     min c.x   s.t.  A x - s = 0,   lb <= x <= ub,   lo <= s <= hi
// The basis is kept as a SPARSE LU factorisation (singleton columns and rows peeled off, the nucleus right-looking with
// Markowitz' pivot choice under threshold partial pivoting; a dependent column is replaced by the slack of a row that found
// no pivot) with middle-product-form updates between L and U (spike and row of U^-1: see ftran), refactorised when the update file has cost as much as a factorisation, at
// the latest every 100 pivots (250 on large bases).  FTRAN / BTRAN start from the right-hand side's nonzeros (a depth-first
// reach over the factors, kept by columns and by rows) and fall back to dense loops when the reach is large; the pivot row is
// formed from the ROWS of A (only rows with a nonzero in row r of the inverse are visited, their nonbasic entries kept in front);
// two solves of a pivot whose results nothing needs before its end run on a helper thread on larger LPs (SolveHelper).  Infinite bounds are boxed (+-BIG) so that ANY basis is
// dual feasible once every nonbasic variable sits on the bound its reduced cost points to -- which is also what lets a
// solve start from a basis guessed from another engine's solution (cuoptamd_dual_simplex_from: the crossover of a PDLP
// solution -- a primal simplex, below, takes that basis to optimality first).  A solution that leans on a box bound is solved again with a 1000 times larger box, and if it still does,
// the LP is unbounded.  Host code: the reference's simplex is CPU code too; PDLP on the GPU stays the engine for large LPs.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <numeric>
#include <set>
#include <thread>
#include <vector>

#include "cuopt_amd/pdlp_solver.h"
#include "host_parallel.hpp"
#include "simplex_presolve.hpp"

namespace {

// the caller's cancel flag: written by another thread (a Concurrent solve), so every look at it is an atomic load
inline bool flag_set(const volatile int32_t* flag) { return flag && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0; }

constexpr double kInf = std::numeric_limits<double>::infinity();
// threshold partial pivoting in the nucleus: a pivot is at least this share of its column's largest entry.  0.1 -> 0.01 leaves Markowitz
// more choice: 15 % fewer entries in L + U of a random 4000 x 3000 LP's bases (11.6 -> 9.3 s), 7-25 % less time on staircase / power-law
// LPs; the drift check behind every entering column's solve is what watches the growth this allows
constexpr double kPivotThreshold = 0.01;
constexpr int kRefactorEvery = 100;  // pivots between factorisations at the latest in the primal simplex; the dual simplex: 500 (the entry counts decide before that)

struct Cancelled {};  // thrown out of a factorisation when the other engine of a Concurrent solve has finished

struct Simplex {
  int m = 0, n = 0, N = 0;
  const volatile int32_t* cancel = nullptr;  // the stop flag of the other engine: read through flag_set (an atomic load)
  const int32_t* rp = nullptr;  // rows of A: the caller's CSR
  const int32_t* rj = nullptr;
  const double* rv  = nullptr;
  // columns of A (CSC) -- column j of M = [A | -I] is A's column j for j < n, -e_(j-n) behind
  std::vector<int32_t> cp, ci;
  std::vector<double> cv;
  std::vector<double> g, L, U;      // cost, bounds of z = (x, s)
  std::vector<char> boxedL, boxedU; // the bound is an artificial box bound
  std::vector<int> basic, pos;      // basic[k] = variable of basis position k, pos[j] = position or -1
  std::vector<char> atU;            // nonbasic at its upper bound
  std::vector<double> z, d, y;      // primal values, reduced costs, duals
  std::vector<double> beta;         // dual steepest-edge weights, per VARIABLE (positions move at a refactorisation)
  int iterations = 0;
  bool steepest  = true;

  // ---- the factorisation: row prow[k] is the pivot of basis position k; L unit lower (by columns, original row numbers),
  //      U by columns (pivot numbers above the diagonal), Ud the diagonal
  std::vector<int> Lp, Li, Up, Ui, pinv, prow;
  std::vector<double> Lx, Ux, Ud;
  // the same factors by ROWS (pivot numbers), for the solves that start from a few nonzeros: LR[kk] = the entries L(prow[kk], k),
  // k < kk; UR[j] = the entries U(j, k), k > j
  std::vector<int> LRp, LRi, URp, URi;
  std::vector<double> LRx, URx;
  // Scratch of one solve.  W0 serves every solve of the thread that owns the Simplex (its fields are reachable under their plain
  // names below); a second thread solving against the same (unchanging) factors brings its own -- run()'s weight solve does.
  struct SolveWork {
    std::vector<int> vis, sstack, sptr, order, seed, lorder, plist, rmark;
    int vstamp = 0, rstamp = 0;
  };
  SolveWork W0;
  std::vector<int>&vis = W0.vis, &sstack = W0.sstack, &sptr = W0.sptr, &order = W0.order;  // depth-first searches of the sparse solves
  std::vector<int>&seed_buf = W0.seed, &lorder = W0.lorder, &plist = W0.plist, &rmark = W0.rmark;
  int &vstamp = W0.vstamp, &rstamp = W0.rstamp;
  int sparse_solve = 1;  // 0: always the dense loops (CUOPT_AMD_TUNE=simplex_solves=dense), 2: always the sparse ones, 1: by the size of the reach
  // work space
  std::vector<double> wx;
  std::vector<int> inpat, pattern, mark, topo, dstack, dptr, rowcnt;
  int stamp = 0, nucleus = 0;
  int64_t factor_ops = 0;  // work of the last factorisation (entries touched): what a refactorisation is weighed against
  int64_t factor_flops = 0;  // ... its share in the elimination's inner loops (multiply-adds on contiguous columns: cheap per entry), counted apart

  int col_count(int j) const { return j >= n ? 1 : cp[j + 1] - cp[j]; }
  double col_dot(const double* row, int j) const  // row . M_j
  {
    if (j >= n) return -row[j - n];
    double s = 0.0;
    for (int k = cp[j]; k < cp[j + 1]; ++k) s += row[ci[k]] * cv[k];
    return s;
  }

  // LU of the columns `cand` (variables, in priority order when there are more than fit); rewrites basic / pos.  Candidates
  // that are linearly dependent on the ones before them (or come after m pivots were found) are returned in `rejected`;
  // rows left without a pivot get their slack.
  void factor(const std::vector<int>& cand, std::vector<int>* rejected, int nprio = -1)
  {
    // ---- the order of the columns and, where the structure hands it out, the pivot row: bases of LPs are close to triangular.
    //  1. column singletons of the active part (a column with ONE entry in the rows that have no pivot yet: every slack to begin
    //     with): pivot there -- nothing below the pivot, no elimination at all;
    //  2. row singletons (a row that ONE remaining column reaches): pivot there if the entry is not small for its column -- the
    //     column's other entries become multipliers, but no other column has an entry in that row: no fill;
    //  3. what is left (the nucleus): right-looking with Markowitz' pivot choice, below.
    // (nprio >= 0: only the first nprio candidates are ordered this way; the others follow them, shortest first, so that of a
    // dependent set it is never one of the first nprio that is turned away because of one of the others)
    const auto tf0 = std::chrono::steady_clock::now();
    const int nc = nprio >= 0 ? std::min(nprio, (int)cand.size()) : (int)cand.size();
    std::vector<int> rstart(m + 1, 0), ccount(nc, 0), rcount(m, 0);
    auto each_entry = [&](int c, auto&& f) {
      const int j = cand[c];
      if (j >= n) f(j - n, -1.0);
      else
        for (int e = cp[j]; e < cp[j + 1]; ++e) f(ci[e], cv[e]);
    };
    for (int c = 0; c < nc; ++c) each_entry(c, [&](int i, double) { rstart[i + 1]++, ccount[c]++; });
    for (int i = 0; i < m; ++i) rcount[i] = rstart[i + 1], rstart[i + 1] += rstart[i];
    std::vector<int> rcols(rstart[m]), fill_at(rstart.begin(), rstart.end() - 1);
    for (int c = 0; c < nc; ++c) each_entry(c, [&](int i, double) { rcols[fill_at[i]++] = c; });
    std::vector<char> cdone(nc, 0), ractive(m, 1);
    std::vector<std::pair<int, int>> order;  // (candidate, forced pivot row or -1)
    order.reserve(cand.size());
    std::vector<int> queue;
    for (int c = 0; c < nc; ++c)
      if (ccount[c] == 1) queue.push_back(c);
    for (size_t h = 0; h < queue.size(); ++h) {
      const int c = queue[h];
      if (cdone[c] || ccount[c] != 1) continue;
      int row = -1;
      each_entry(c, [&](int i, double) { if (ractive[i]) row = i; });
      if (row < 0) continue;
      order.emplace_back(c, row), cdone[c] = 1, ractive[row] = 0;
      for (int e = rstart[row]; e < rstart[row + 1]; ++e) {
        const int c2 = rcols[e];
        if (!cdone[c2] && --ccount[c2] == 1) queue.push_back(c2);
      }
    }
    queue.clear();
    for (int i = 0; i < m; ++i) {
      if (!ractive[i]) continue;
      int live = 0;
      for (int e = rstart[i]; e < rstart[i + 1]; ++e) live += !cdone[rcols[e]];
      rcount[i] = live;
      if (live == 1) queue.push_back(i);
    }
    for (size_t h = 0; h < queue.size(); ++h) {
      const int i = queue[h];
      if (!ractive[i] || rcount[i] != 1) continue;
      int c = -1;
      for (int e = rstart[i]; e < rstart[i + 1]; ++e)
        if (!cdone[rcols[e]]) c = rcols[e];
      if (c < 0) continue;
      double here = 0.0, colmax = 0.0;
      each_entry(c, [&](int i2, double v) {
        if (!ractive[i2]) return;
        colmax = std::max(colmax, std::fabs(v));
        if (i2 == i) here += v;
      });
      if (std::fabs(here) < 0.01 * colmax || here == 0.0) continue;  // too small a pivot for its column: the nucleus decides
      order.emplace_back(c, i), cdone[c] = 1, ractive[i] = 0;
      each_entry(c, [&](int i2, double) {
        if (ractive[i2] && --rcount[i2] == 1) queue.push_back(i2);
      });
    }
    const auto tf1 = std::chrono::steady_clock::now();
    std::vector<int> nuc, tail;  // the nucleus (candidate numbers) and the candidates behind the first nprio
    for (int c = 0; c < nc; ++c)
      if (!cdone[c]) nuc.push_back(c);
    nucleus = (int)nuc.size();
    for (int c = nc; c < (int)cand.size(); ++c) tail.push_back(c);
    std::stable_sort(tail.begin(), tail.end(), [&](int x, int y) { return col_count(cand[x]) < col_count(cand[y]); });
    rowcnt.swap(rcount);
    factor_ops = rstart[m], factor_flops = 0;
    pinv.assign(m, -1), prow.assign(m, -1);
    Lp.assign(1, 0), Up.assign(1, 0);
    Li.clear(), Lx.clear(), Ui.clear(), Ux.clear(), Ud.clear();
    clear_updates();
    wx.assign(m, 0.0), inpat.assign(m, -1), mark.assign(m, -1);
    std::vector<int> nb;
    nb.reserve(m);
    int k = 0;
    stamp = 0;
    // One column against the pivots found so far (left-looking, Gilbert-Peierls): its entries in U are appended to Ui / Ux,
    // what is left of it in the rows without a pivot stays in wx over `pattern`.  Returns the largest entry of the column itself.
    auto eliminate = [&](int j) {
      if ((stamp & 255) == 0 && flag_set(cancel)) throw Cancelled{};  // (a large nucleus can take seconds: the caller is waiting)
      const int st = stamp++;
      pattern.clear();
      auto touch = [&](int i) {
        if (inpat[i] != st) inpat[i] = st, pattern.push_back(i);
      };
      double colmax = 1.0;
      if (j >= n) wx[j - n] = -1.0, touch(j - n);
      else {
        colmax = 0.0;
        for (int e = cp[j]; e < cp[j + 1]; ++e) wx[ci[e]] += cv[e], touch(ci[e]), colmax = std::max(colmax, std::fabs(cv[e]));  // (+=: a file may list an entry twice)
      }
      // the pivots this column reaches through L, in topological order (depth-first, iterative)
      topo.clear();
      const size_t n0 = pattern.size();
      for (size_t t = 0; t < n0; ++t) {
        const int start = pinv[pattern[t]];
        if (start < 0 || mark[start] == st) continue;
        dstack.assign(1, start), dptr.assign(1, Lp[start]);
        mark[start] = st;
        while (!dstack.empty()) {
          const int jj = dstack.back();
          int& e       = dptr.back();
          bool down    = false;
          while (e < Lp[jj + 1]) {
            const int nx = pinv[Li[e++]];
            if (nx >= 0 && mark[nx] != st) {
              mark[nx] = st;
              dstack.push_back(nx), dptr.push_back(Lp[nx]);
              down = true;
              break;
            }
          }
          if (!down) {
            topo.push_back(jj);
            dstack.pop_back(), dptr.pop_back();
          }
        }
      }
      for (size_t t = topo.size(); t-- > 0;) {
        const int jj    = topo[t];
        const double xj = wx[prow[jj]];
        if (xj == 0.0) continue;
        Ui.push_back(jj), Ux.push_back(xj);
        for (int e = Lp[jj]; e < Lp[jj + 1]; ++e) touch(Li[e]), wx[Li[e]] -= Lx[e] * xj;
        factor_ops += Lp[jj + 1] - Lp[jj] + 1;
      }
      factor_ops += (int64_t)pattern.size();
      return colmax;
    };
    // ... and its pivot: `forced` if that row is acceptable, else threshold partial pivoting with the sparser row preferred
    auto left_looking = [&](int j, int forced) {
      if (k == m) {
        if (rejected) rejected->push_back(j);
        return;
      }
      if (j >= n && pinv[j - n] < 0) {  // a slack whose row is free (most of a basis): -1 on the diagonal, nothing else
        Ud.push_back(-1.0), Up.push_back((int)Ui.size()), Lp.push_back((int)Li.size());
        pinv[j - n] = k, prow[k] = j - n, nb.push_back(j), ++k;
        ++stamp;
        return;
      }
      const double colmax = eliminate(j);
      double best = 0.0;
      for (int i : pattern)
        if (pinv[i] < 0) best = std::max(best, std::fabs(wx[i]));
      if (best <= std::max(1e-11, 1e-9 * colmax)) {  // dependent on the columns before it
        Ui.resize(Up.back()), Ux.resize(Up.back());
        for (int i : pattern) wx[i] = 0.0;
        if (rejected) rejected->push_back(j);
        return;
      }
      int piv = forced >= 0 && pinv[forced] < 0 && std::fabs(wx[forced]) >= 0.01 * best ? forced : -1;
      for (int i : pattern) {
        if (piv == forced && forced >= 0) break;
        if (pinv[i] >= 0 || std::fabs(wx[i]) < 0.1 * best) continue;
        if (piv < 0 || rowcnt[i] < rowcnt[piv] || (rowcnt[i] == rowcnt[piv] && std::fabs(wx[i]) > std::fabs(wx[piv]))) piv = i;
      }
      const double pv = wx[piv];
      Ud.push_back(pv), Up.push_back((int)Ui.size());
      for (int i : pattern)
        if (pinv[i] < 0 && i != piv && wx[i] != 0.0) Li.push_back(i), Lx.push_back(wx[i] / pv);
      Lp.push_back((int)Li.size());
      for (int i : pattern) wx[i] = 0.0;
      pinv[piv] = k, prow[k] = piv, nb.push_back(j), ++k;
    };
    const auto tf2 = std::chrono::steady_clock::now();
    for (const auto& oc : order) left_looking(cand[oc.first], oc.second);
    const auto tf3 = std::chrono::steady_clock::now();
    // ---- the nucleus: right-looking elimination with Markowitz' pivot choice.  Every nucleus column is first taken through the
    // triangular part (its U entries there), what is left of it lives in `cols` (local row numbers); a pivot is the entry with the
    // smallest (row count - 1)(column count - 1) among the entries within a factor 100 of their column's largest, looked for in
    // the four shortest columns; the pivot row goes to the U columns of the columns it touches, the multipliers to L.
    if (!nuc.empty() && k < m) {
      struct Entry {
        int r;
        double v;
      };
      const int q = (int)nuc.size();
      std::vector<int> rloc(m, -1), rglob;
      for (int i = 0; i < m; ++i)
        if (pinv[i] < 0) rloc[i] = (int)rglob.size(), rglob.push_back(i);
      const int nr = (int)rglob.size();
      std::vector<std::vector<Entry>> cols(q);
      std::vector<std::vector<int>> rows(nr);
      std::vector<std::vector<std::pair<int, double>>> ucol(q);
      std::vector<double> cmax0(q, 0.0);
      std::vector<int> rcnt(nr, 0), key(q, 0), where(nr, 0);
      for (int c = 0; c < q; ++c) {
        cmax0[c] = eliminate(cand[nuc[c]]);
        for (int e = Up.back(); e < (int)Ui.size(); ++e) ucol[c].emplace_back(Ui[e], Ux[e]);
        Ui.resize(Up.back()), Ux.resize(Up.back());
        for (int i : pattern) {
          if (pinv[i] < 0 && wx[i] != 0.0) cols[c].push_back({rloc[i], wx[i]}), rows[rloc[i]].push_back(c), rcnt[rloc[i]]++;
          wx[i] = 0.0;
        }
      }
      const auto tn1 = std::chrono::steady_clock::now();
      if (debug) nsec[0] += std::chrono::duration<double>(tn1 - tf3).count();
      // the columns by (count, index): a binary heap with lazy deletion -- an entry is current while its version is the column's
      // (same order as an ordered set gives, without its two tree walks and a node per changed count)
      struct ByCount {
        int cnt, c, ver;
        bool operator<(const ByCount& o) const { return cnt != o.cnt ? cnt > o.cnt : c > o.c; }  // (std::*_heap keep the LARGEST on top)
      };
      std::vector<ByCount> heap;
      heap.reserve(4 * (size_t)q);
      std::vector<int> ver(q, 0);
      auto heap_push = [&](int c) { heap.push_back({key[c], c, ver[c]}), std::push_heap(heap.begin(), heap.end()); };
      auto heap_pop_current = [&]() -> int {  // the shortest current column, taken out; -1 when there is none
        while (!heap.empty()) {
          std::pop_heap(heap.begin(), heap.end());
          const ByCount t = heap.back();
          heap.pop_back();
          if (t.ver == ver[t.c]) return t.c;
        }
        return -1;
      };
      for (int c = 0; c < q; ++c) key[c] = (int)cols[c].size(), heap_push(c);
      factor_ops += 40 * (int64_t)q + 2 * (int64_t)nr;
      std::vector<char> gone(q, 0);
      std::vector<Entry> lmul;
      int polls = 0;
      int shortest[4];
      while (k < m) {
        if ((++polls & 63) == 0 && flag_set(cancel)) throw Cancelled{};
        int pc = -1, pr = -1;
        double pval = 0.0;
        int64_t pcost = std::numeric_limits<int64_t>::max();
        int looked = 0;
        const auto ts0 = debug ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        for (; looked < 4; ++looked) {
          const int c = heap_pop_current();
          if (c < 0) break;
          shortest[looked] = c;
          double colmax = 0.0;
          for (const Entry& e : cols[c]) colmax = std::max(colmax, std::fabs(e.v));
          factor_ops += 2 * (int64_t)cols[c].size() + 8;
          if (colmax <= std::max(1e-11, 1e-9 * cmax0[c])) continue;  // (turned away below when it is the shortest one)
          for (const Entry& e : cols[c]) {
            if (std::fabs(e.v) < kPivotThreshold * colmax) continue;
            const int64_t cost = (int64_t)(rcnt[e.r] - 1) * (int64_t)(cols[c].size() - 1);
            if (cost < pcost || (cost == pcost && std::fabs(e.v) > std::fabs(pval))) pcost = cost, pc = c, pr = e.r, pval = e.v;
          }
          if (pcost == 0) {
            ++looked;
            break;
          }
        }
        if (looked == 0) break;  // no column left
        const auto ts1 = debug ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        if (debug) nsec[1] += std::chrono::duration<double>(ts1 - ts0).count();
        for (int t = 0; t < looked; ++t)  // back with the ones that were only looked at (the pivot column / the one turned away stays out)
          if (shortest[t] != (pc < 0 ? shortest[0] : pc)) heap_push(shortest[t]);
        if (pc < 0) {  // the shortest columns have nothing left: dependent on the pivots so far
          const int c = shortest[0];
          ++ver[c];
          gone[c] = 1;
          for (const Entry& e : cols[c]) rcnt[e.r]--;
          cols[c].clear();
          if (rejected) rejected->push_back(cand[nuc[c]]);
          continue;
        }
        ++ver[pc];
        gone[pc] = 1;
        Ud.push_back(pval);
        for (const auto& u : ucol[pc]) Ui.push_back(u.first), Ux.push_back(u.second);
        Up.push_back((int)Ui.size());
        lmul.clear();
        for (const Entry& e : cols[pc]) {
          rcnt[e.r]--;
          if (e.r != pr) lmul.push_back({e.r, e.v / pval}), Li.push_back(rglob[e.r]), Lx.push_back(e.v / pval);
        }
        Lp.push_back((int)Li.size());
        const int row = rglob[pr];
        pinv[row] = k, prow[k] = row, nb.push_back(cand[nuc[pc]]);
        for (int c : rows[pr]) {
          if (gone[c]) continue;
          std::vector<Entry>& col = cols[c];
          size_t at = 0;
          while (at < col.size() && col[at].r != pr) ++at;
          if (at == col.size()) continue;
          const double u = col[at].v;
          col[at]        = col.back(), col.pop_back();
          ucol[c].emplace_back(k, u);
          if (u != 0.0 && !lmul.empty()) {
            const size_t before = col.size();
            for (size_t t = 0; t < before; ++t) where[col[t].r] = (int)t + 1;
            for (const Entry& l : lmul) {
              if (where[l.r]) col[where[l.r] - 1].v -= l.v * u;
              else col.push_back({l.r, -l.v * u}), rows[l.r].push_back(c), rcnt[l.r]++;
            }
            for (size_t t = 0; t < before; ++t) where[col[t].r] = 0;
            factor_flops += (int64_t)before + (int64_t)lmul.size();
          }
          if ((int)col.size() != key[c]) key[c] = (int)col.size(), ++ver[c], heap_push(c);
          factor_ops += 40 + (int64_t)at;  // (the 40: the count order's upkeep, as the refactorisation rule was calibrated)
        }
        rows[pr].clear(), cols[pc].clear();
        ++k;
        if (debug) nsec[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count();
      }
      for (int c = heap_pop_current(); c >= 0; c = heap_pop_current())
        if (rejected) rejected->push_back(cand[nuc[c]]);
    } else if (!nuc.empty()) {
      for (int c : nuc)
        if (rejected) rejected->push_back(cand[c]);
    }
    for (int c : tail) left_looking(cand[c], -1);
    for (int i = 0; i < m; ++i)
      if (pinv[i] < 0) {
        Ud.push_back(-1.0), Up.push_back((int)Ui.size()), Lp.push_back((int)Li.size());
        pinv[i] = k, prow[k] = i, nb.push_back(n + i), ++k;
      }
    basic.swap(nb);
    std::fill(pos.begin(), pos.end(), -1);
    for (int q = 0; q < m; ++q) pos[basic[q]] = q;
    const auto tf4 = std::chrono::steady_clock::now();
    // row-wise copies of L and U
    LRp.assign(m + 1, 0), URp.assign(m + 1, 0);
    for (int kk = 0; kk < m; ++kk)
      for (int e = Lp[kk]; e < Lp[kk + 1]; ++e) LRp[pinv[Li[e]] + 1]++;
    for (int kk = 0; kk < m; ++kk)
      for (int e = Up[kk]; e < Up[kk + 1]; ++e) URp[Ui[e] + 1]++;
    for (int i = 0; i < m; ++i) LRp[i + 1] += LRp[i], URp[i + 1] += URp[i];
    LRi.resize(Li.size()), LRx.resize(Li.size()), URi.resize(Ui.size()), URx.resize(Ui.size());
    {
      std::vector<int> atl(LRp.begin(), LRp.end() - 1), atu(URp.begin(), URp.end() - 1);
      for (int kk = 0; kk < m; ++kk) {
        for (int e = Lp[kk]; e < Lp[kk + 1]; ++e) {
          const int q2 = atl[pinv[Li[e]]]++;
          LRi[q2] = kk, LRx[q2] = Lx[e];
        }
        for (int e = Up[kk]; e < Up[kk + 1]; ++e) {
          const int q2 = atu[Ui[e]]++;
          URi[q2] = kk, URx[q2] = Ux[e];
        }
      }
    }
    vis.assign(m, 0), vstamp = 0;
    if (debug) {
      auto ms = [](auto a, auto b) { return 1e3 * std::chrono::duration<double>(b - a).count(); };
      fsec[0] += ms(tf0, tf1), fsec[1] += ms(tf1, tf2), fsec[2] += ms(tf2, tf3), fsec[3] += ms(tf3, tf4), fsec[4] += ms(tf4, std::chrono::steady_clock::now());
    }
  }
  double nsec[3] = {0, 0, 0};  // debug, s: nucleus columns through the triangular part / pivot search / elimination
  double fsec[5] = {0, 0, 0, 0, 0};  // debug, ms: ordering / set-up / triangular part / nucleus + tail / row-wise copies
  // ---- solves that start from a few nonzeros (Gilbert-Peierls): a depth-first search over the factor's structure finds the pivots
  // the right-hand side reaches, in topological order; only those are visited.  `from` (pivot numbers) seeds the search, `next(k, f)`
  // calls f for every pivot k points to.  Gives up (returns false, nothing changed) once more than m / 6 pivots are reached:
  // the dense loops are faster then.
  template <class Next>
  bool reach(SolveWork& W, const std::vector<int>& from, Next&& next)
  {
    std::vector<int>&vis = W.vis, &sstack = W.sstack, &sptr = W.sptr, &order = W.order;
    int& vstamp = W.vstamp;
    order.clear();
    if (sparse_solve == 0) return false;
    const size_t limit = sparse_solve == 2 ? (size_t)m + 1 : (size_t)std::max(16, m / 6);
    if (from.size() > limit) return false;
    const int st = ++vstamp;
    for (int start : from) {
      if (vis[start] == st) continue;
      vis[start] = st;
      sstack.assign(1, start), sptr.assign(1, 0);
      while (!sstack.empty()) {
        const int k = sstack.back();
        int child   = -1;
        next(k, sptr.back(), child);  // advances the pointer, returns an unvisited neighbour in `child` or leaves it at -1
        if (child >= 0) {
          vis[child] = st;
          sstack.push_back(child), sptr.push_back(0);
        } else {
          order.push_back(k);
          sstack.pop_back(), sptr.pop_back();
          if (order.size() > limit) return false;
        }
      }
    }
    return true;  // `order` is a postorder: reversed, every pivot comes before the ones it points to
  }
  // ---- updates between two factorisations: the MIDDLE product form.  B = L M_1 ... M_K U with M_k = I + v_k e~_k^T,
  //      v_k = (L M_1 ... M_(k-1))^-1 a_q - U e_p  (the entering column after the L solve -- the "spike" -- minus the column of U it
  //      replaces) and e~_k = U^-T e_p (row p of U^-1): both fall out of the pivot's own solves half way (the spike before FTRAN's U
  //      stage, e~ behind BTRAN's U^T stage), and both are far sparser than the fully transformed column B^-1 a_q the plain product
  //      form stores (10 000-row block-angular LP: ~300 + ~1000 entries against ~8000 per pivot) -- every solve walks the whole update
  //      file, so that is what a pivot costs.  M_k^-1 x = x - v_k (e~_k . x) / w_p, w_p = e~_k . spike = the pivot element.  L and U never
  //      change.  Vectors are kept by ROW (position k <-> row prow[k]).
  std::vector<int> Mvp, Mvi, Mep, Mei;
  std::vector<double> Mvx, Mex, Mw;
  std::vector<int> spike_i, et_i;   // what the last FTRAN (keep = true) / BTRAN left half way
  std::vector<double> spike_x, et_x, scr;
  size_t update_entries() const { return Mvi.size() + Mei.size(); }
  void clear_updates() { Mvp.assign(1, 0), Mep.assign(1, 0), Mvi.clear(), Mvx.clear(), Mei.clear(), Mex.clear(), Mw.clear(); }
  // x (by ROW) <- M_K^-1 ... M_1^-1 x; with `track` the rows that become nonzero are appended to plist (marked with rstamp)
  void apply_updates_forward(SolveWork& W, std::vector<double>& x, bool track)
  {
    std::vector<int>&plist = W.plist, &rmark = W.rmark;
    const int rstamp = W.rstamp;
    const int nu = (int)Mw.size();
    for (int e = 0; e < nu; ++e) {
      double dot = 0.0;
      for (int q = Mep[e]; q < Mep[e + 1]; ++q) dot += Mex[q] * x[Mei[q]];
      if (dot == 0.0) continue;
      const double t = dot / Mw[e];
      for (int q = Mvp[e]; q < Mvp[e + 1]; ++q) {
        const int i = Mvi[q];
        if (track && rmark[i] != rstamp) rmark[i] = rstamp, plist.push_back(i);
        x[i] -= Mvx[q] * t;
      }
    }
  }
  // t (by POSITION) <- M_1^-T ... M_K^-T t; with `track` the positions that become nonzero are appended to plist
  void apply_updates_backward(std::vector<double>& t, bool track)
  {
    for (int e = (int)Mw.size() - 1; e >= 0; --e) {
      double dot = 0.0;
      for (int q = Mvp[e]; q < Mvp[e + 1]; ++q) dot += Mvx[q] * t[pinv[Mvi[q]]];
      if (dot == 0.0) continue;
      const double s = dot / Mw[e];
      for (int q = Mep[e]; q < Mep[e + 1]; ++q) {
        const int k = pinv[Mei[q]];
        if (track && rmark[k] != rstamp) rmark[k] = rstamp, plist.push_back(k);
        t[k] -= Mex[q] * s;
      }
    }
  }
  // w = B^-1 a.  x holds a by ROW with its nonzero rows in xrows; on return x is all zero, w (by POSITION, zero on entry) holds the
  // result with its nonzero positions in wlist.  keep: the vector between the update stage and the U stage (the spike of an entering
  // column) is copied to spike_i / spike_x.
  void ftran(std::vector<double>& x, const std::vector<int>& xrows, std::vector<double>& w, std::vector<int>& wlist, bool keep = false, SolveWork* work = nullptr)
  {
    SolveWork& W = work ? *work : W0;
    std::vector<int>&vis = W.vis, &order = W.order, &seed = W.seed, &lorder = W.lorder, &plist = W.plist, &rmark = W.rmark;
    int &vstamp = W.vstamp, &rstamp = W.rstamp;
    wlist.clear();
    if ((int)rmark.size() != m) rmark.assign(m, 0);
    if ((int)vis.size() != m) vis.assign(m, 0), vstamp = 0;
    if (!work && (int)scr.size() != m) scr.assign(m, 0.0);
    seed.clear();
    for (int r : xrows) seed.push_back(pinv[r]);
    bool sparse = reach(W, seed, [&](int k, int& ptr, int& child) {
      while (Lp[k] + ptr < Lp[k + 1]) {
        const int nx = pinv[Li[Lp[k] + ptr++]];
        if (vis[nx] != vstamp) { child = nx; return; }
      }
    });
    // ---- L
    if (sparse) {
      lorder.assign(order.rbegin(), order.rend());
      for (int k : lorder) {
        const double xj = x[prow[k]];
        if (xj == 0.0) continue;
        for (int e = Lp[k]; e < Lp[k + 1]; ++e) x[Li[e]] -= Lx[e] * xj;
      }
      ++rstamp;
      plist.clear();
      for (int k : lorder) rmark[prow[k]] = rstamp, plist.push_back(prow[k]);
    } else {
      for (int k = 0; k < m; ++k) {
        const double xj = x[prow[k]];
        if (xj == 0.0) continue;
        for (int e = Lp[k]; e < Lp[k + 1]; ++e) x[Li[e]] -= Lx[e] * xj;
      }
    }
    // ---- the updates
    if (!Mw.empty()) apply_updates_forward(W, x, sparse);
    if (keep) {
      spike_i.clear(), spike_x.clear();
      if (sparse) {
        for (int i : plist)
          if (x[i] != 0.0) spike_i.push_back(i), spike_x.push_back(x[i]);
      } else {
        for (int i = 0; i < m; ++i)
          if (x[i] != 0.0) spike_i.push_back(i), spike_x.push_back(x[i]);
      }
    }
    // ---- U
    if (sparse) {
      seed.clear();
      for (int i : plist)
        if (x[i] != 0.0) seed.push_back(pinv[i]);
      sparse = reach(W, seed, [&](int k, int& ptr, int& child) {
        while (Up[k] + ptr < Up[k + 1]) {
          const int nx = Ui[Up[k] + ptr++];
          if (vis[nx] != vstamp) { child = nx; return; }
        }
      });
      if (sparse) {
        for (size_t t = order.size(); t-- > 0;) {
          const int k = order[t];
          double v    = x[prow[k]];
          x[prow[k]]  = 0.0;
          if (v == 0.0) continue;
          v /= Ud[k];
          for (int e = Up[k]; e < Up[k + 1]; ++e) x[prow[Ui[e]]] -= Ux[e] * v;
          w[k] = v;
          wlist.push_back(k);
        }
        for (int i : plist) x[i] = 0.0;  // (rows the L stage / the updates touched that cancelled to a value the U stage never met)
        return;
      }
    }
    for (int k = m - 1; k >= 0; --k) {
      double v = x[prow[k]];
      if (v != 0.0) {
        v /= Ud[k];
        for (int e = Up[k]; e < Up[k + 1]; ++e) x[prow[Ui[e]]] -= Ux[e] * v;
        w[k] = v;
        wlist.push_back(k);
      }
    }
    std::fill(x.begin(), x.end(), 0.0);
  }
  // rho = B^-T t.  t by POSITION with its nonzero positions in tlist; on return t is all zero, rho (by ROW, zero on entry) holds the
  // result with its nonzero rows in rlist.  The vector behind the U^T stage (row p of U^-1 when t = e_p) is copied to et_i / et_x.
  void btran(std::vector<double>& t, std::vector<int>& tlist, std::vector<double>& rho, std::vector<int>& rlist)
  {
    rlist.clear();
    if ((int)rmark.size() != m) rmark.assign(m, 0), scr.assign(m, 0.0);
    bool sparse = reach(W0, tlist, [&](int k, int& ptr, int& child) {
      while (URp[k] + ptr < URp[k + 1]) {
        const int nx = URi[URp[k] + ptr++];
        if (vis[nx] != vstamp) { child = nx; return; }
      }
    });
    // ---- U^T
    et_i.clear(), et_x.clear();
    if (sparse) {
      lorder.assign(order.rbegin(), order.rend());
      for (int j : lorder) {  // by the rows of U
        double sv = t[j];
        if (sv == 0.0) continue;
        sv /= Ud[j];
        t[j] = sv;
        for (int e = URp[j]; e < URp[j + 1]; ++e) t[URi[e]] -= URx[e] * sv;
      }
      ++rstamp;
      plist.clear();
      for (int j : lorder) {
        rmark[j] = rstamp, plist.push_back(j);
        if (t[j] != 0.0) et_i.push_back(prow[j]), et_x.push_back(t[j]);
      }
    } else {
      // (too many pivots reached for the search to pay: all of them in their natural order, by the ROWS of U as above -- a position
      //  that is still zero costs a load, not its row; the dot-product form over U's columns touched every entry of U)
      for (int j = 0; j < m; ++j) {
        double sv = t[j];
        if (sv == 0.0) continue;
        sv /= Ud[j];
        t[j] = sv;
        for (int e = URp[j]; e < URp[j + 1]; ++e) t[URi[e]] -= URx[e] * sv;
        if (sv != 0.0) et_i.push_back(prow[j]), et_x.push_back(sv);
      }
    }
    // ---- the updates, last one first
    if (!Mw.empty()) apply_updates_backward(t, sparse);
    // ---- L^T
    if (sparse) {
      std::vector<int>& seed = seed_buf;
      seed.clear();
      for (int k : plist)
        if (t[k] != 0.0) seed.push_back(k);
      sparse = reach(W0, seed, [&](int k, int& ptr, int& child) {
        while (LRp[k] + ptr < LRp[k + 1]) {
          const int nx = LRi[LRp[k] + ptr++];
          if (vis[nx] != vstamp) { child = nx; return; }
        }
      });
      if (sparse) {
        for (size_t q = order.size(); q-- > 0;) {  // by the rows of L
          const int kk   = order[q];
          const double v = t[kk];
          t[kk]          = 0.0;
          if (v == 0.0) continue;
          rho[prow[kk]] = v;
          rlist.push_back(prow[kk]);
          for (int e = LRp[kk]; e < LRp[kk + 1]; ++e) t[LRi[e]] -= LRx[e] * v;
        }
        for (int k : plist) t[k] = 0.0;
        return;
      }
    }
    for (int kk = m - 1; kk >= 0; --kk) {  // by the rows of L, every position; zeros skipped
      const double v = t[kk];
      if (v == 0.0) continue;
      t[kk]         = 0.0;
      rho[prow[kk]] = v;
      rlist.push_back(prow[kk]);
      for (int e = LRp[kk]; e < LRp[kk + 1]; ++e) t[LRi[e]] -= LRx[e] * v;
    }
  }
  bool debug = false;
  // w = B^-1 a : `x` holds a by ROW and is destroyed, w comes back by POSITION
  void ftran_dense(std::vector<double>& x, std::vector<double>& w)
  {
    for (int k = 0; k < m; ++k) {
      const double xj = x[prow[k]];
      if (xj == 0.0) continue;
      for (int e = Lp[k]; e < Lp[k + 1]; ++e) x[Li[e]] -= Lx[e] * xj;
    }
    if (!Mw.empty()) apply_updates_forward(W0, x, false);
    for (int k = m - 1; k >= 0; --k) {
      double v = x[prow[k]];
      if (v != 0.0) {
        v /= Ud[k];
        for (int e = Up[k]; e < Up[k + 1]; ++e) x[prow[Ui[e]]] -= Ux[e] * v;
      }
      w[k] = v;
    }
  }
  // rho = B^-T t : `t` by POSITION (destroyed), rho by ROW
  void btran_dense(std::vector<double>& t, std::vector<double>& rho)
  {
    for (int k = 0; k < m; ++k) {
      double sv = t[k];
      for (int e = Up[k]; e < Up[k + 1]; ++e) sv -= Ux[e] * t[Ui[e]];
      t[k] = sv / Ud[k];
    }
    if (!Mw.empty()) apply_updates_backward(t, false);
    for (int k = m - 1; k >= 0; --k) {
      double sv = t[k];
      for (int e = Lp[k]; e < Lp[k + 1]; ++e) sv -= Lx[e] * rho[Li[e]];
      rho[prow[k]] = sv;
    }
  }
  // the pivot at position r: the last FTRAN with keep = true left the entering column's spike, the last BTRAN (of e_r) row r of U^-1
  void push_update(int r, double pivot)
  {
    // v = spike - U e_r
    for (size_t q = 0; q < spike_i.size(); ++q) scr[spike_i[q]] = spike_x[q];
    ++rstamp;
    plist.clear();
    for (int i : spike_i) rmark[i] = rstamp, plist.push_back(i);
    auto sub = [&](int row, double v) {
      if (rmark[row] != rstamp) rmark[row] = rstamp, plist.push_back(row);
      scr[row] -= v;
    };
    for (int e = Up[r]; e < Up[r + 1]; ++e) sub(prow[Ui[e]], Ux[e]);
    sub(prow[r], Ud[r]);
    for (int i : plist) {
      if (scr[i] != 0.0) Mvi.push_back(i), Mvx.push_back(scr[i]);
      scr[i] = 0.0;
    }
    Mvp.push_back((int)Mvi.size());
    Mei.insert(Mei.end(), et_i.begin(), et_i.end()), Mex.insert(Mex.end(), et_x.begin(), et_x.end());
    Mep.push_back((int)Mei.size());
    Mw.push_back(pivot);
  }
  // debug: || B w - a || and || B^T rho - t || for one right-hand side each
  void check_factor(const char* where)
  {
    std::vector<double> a(m), w(m), x(m), t(m), rho(m, 0.0);
    for (int i = 0; i < m; ++i) a[i] = std::sin(1.0 + i), t[i] = std::cos(2.0 + i);
    x = a;
    ftran_dense(x, w);
    std::vector<double> res(a);
    for (int k = 0; k < m; ++k) {
      const int j = basic[k];
      if (j >= n) res[j - n] += w[k];
      else
        for (int e = cp[j]; e < cp[j + 1]; ++e) res[ci[e]] -= cv[e] * w[k];
    }
    double e1 = 0.0, e2 = 0.0;
    for (int i = 0; i < m; ++i) e1 = std::max(e1, std::fabs(res[i]));
    x = t;
    btran_dense(x, rho);
    for (int k = 0; k < m; ++k) e2 = std::max(e2, std::fabs(col_dot(rho.data(), basic[k]) - t[k]));
    std::fprintf(stderr, "[simplex] %s: residual of B w = a %.3g, of B^T rho = t %.3g\n", where, e1, e2);
  }
  // z_B, y, d from the nonbasic values and the current factorisation
  void recompute()
  {
    std::vector<double> rhs(m, 0.0), w(m, 0.0);  // -N z_N
    for (int j = 0; j < N; ++j) {
      if (pos[j] >= 0) continue;
      const double v = z[j];
      if (v == 0.0) continue;
      if (j >= n) rhs[j - n] += v;
      else
        for (int k = cp[j]; k < cp[j + 1]; ++k) rhs[ci[k]] -= cv[k] * v;
    }
    ftran_dense(rhs, w);
    for (int k = 0; k < m; ++k) z[basic[k]] = w[k];
    for (int k = 0; k < m; ++k) w[k] = g[basic[k]];
    btran_dense(w, y);
    for (int j = 0; j < N; ++j) d[j] = pos[j] >= 0 ? 0.0 : g[j] - col_dot(y.data(), j);
  }
  // every nonbasic variable onto the bound its reduced cost points to (boxed: always possible); true when something moved
  bool make_dual_feasible(double tol_d)
  {
    bool moved = false;
    for (int j = 0; j < N; ++j) {
      if (pos[j] >= 0 || L[j] == U[j]) continue;
      const bool want_upper = d[j] < -tol_d, want_lower = d[j] > tol_d;
      if ((want_upper && !atU[j]) || (want_lower && atU[j])) {
        atU[j] = want_upper;
        z[j]   = atU[j] ? U[j] : L[j];
        moved  = true;
      }
    }
    return moved;
  }
  // a fresh factorisation of the current basis (repaired where it has become singular), then everything derived from it
  void rebuild(double tol_d)
  {
    const auto t_in = std::chrono::steady_clock::now();
    std::vector<int> rejected, cand(basic);
    const size_t eta_entries = update_entries();
    factor(cand, &rejected);
    if (debug) tsec[6] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count(), ops_factor += factor_ops + factor_flops;
    for (int j : rejected) {  // left the basis: onto the nearer bound
      atU[j] = std::fabs(U[j] - z[j]) < std::fabs(z[j] - L[j]);
      z[j]   = atU[j] ? U[j] : L[j];
    }
    for (int k = 0; k < m; ++k)
      if (beta[basic[k]] <= 0.0) beta[basic[k]] = 1.0;
    recompute();
    if (make_dual_feasible(tol_d)) recompute();
    if (debug) tsec[7] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
    if (debug && (++rebuilds % 10) == 0) {
      int structurals = 0;
      for (int k = 0; k < m; ++k) structurals += basic[k] < n;
      std::fprintf(stderr, "[simplex] rebuild %d: %d structural columns in the basis, nucleus %d, L %zu + U %zu off-diagonal entries (updates before: %zu), %zu repaired\n", rebuilds, structurals, nucleus, Li.size(), Ui.size(), eta_entries, rejected.size());
    }
  }
  // the same without the move to dual feasibility (the primal simplex keeps its nonbasic variables where they are)
  void rebuild_plain()
  {
    std::vector<int> rejected, cand(basic);
    factor(cand, &rejected);
    for (int j : rejected) {
      atU[j] = std::fabs(U[j] - z[j]) < std::fabs(z[j] - L[j]);
      z[j]   = atU[j] ? U[j] : L[j];
    }
    recompute();
  }
  int rebuilds = 0;
  int64_t ops_factor = 0, ops_solve = 0;
  double dens[6] = {0, 0, 0, 0, 0, 0};  // debug: sums over the pivots of |rho|, |w|, |tau|, |pivot row|, update entries, entries of L + U
  int64_t dens_pivots = 0, bf_pass = 0, bf_grp = 0, bf_rounds = 0;  // ... the pivots counted, those with bound flips, the flips
  double tsec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // debug: seconds in {pricing, btran, pivot row, ratio test, ftran, weights, factorisations, rebuild, duals + primal values, update file}
};
struct Lap {
  double* acc;
  std::chrono::steady_clock::time_point t0;
  Lap(Simplex& S, int slot) : acc(S.debug ? &S.tsec[slot] : nullptr) { if (acc) t0 = std::chrono::steady_clock::now(); }
  ~Lap() { if (acc) *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Two solves of a pivot have results nobody reads before its end: tau = B^-1 rho for the steepest-edge weights, and the change of
// the basic variables behind a set of bound flips.  On LPs where a solve is worth a hand-over they run on a helper thread -- own
// scratch, the same factors and update file, which nothing writes until the pivot is done -- next to the pivot row, the ratio test
// and the entering column's solve.  One job per slot at a time; the helper spins briefly, then yields, while it waits.
struct SolveHelper {
  struct Job {
    std::vector<double> rhs;  // by ROW (handed back zeroed by the solve)
    std::vector<int> rows;
    std::vector<double>* out = nullptr;
    std::vector<int>* outlist = nullptr;
    std::atomic<int> state{0};  // 0 idle, 1 posted, 2 done
  };
  Simplex& S;
  Job job[2];
  Simplex::SolveWork W;
  std::atomic<int> quit{0}, failed{0};
  std::thread th;
  explicit SolveHelper(Simplex& s) : S(s)
  {
    for (Job& j : job) j.rhs.assign(s.m, 0.0);
  }
  bool start()
  {
    try {
      th = std::thread([this] { loop(); });
    } catch (...) {
      return false;
    }
    return true;
  }
  static inline void cpu_relax()
  {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
  }
  static void relax(int& spins)  // a short spin (a pivot's jobs follow each other within microseconds), then yields, then naps:
  {                               // an idle helper next to a slow main loop must not keep a core busy for the length of the solve
    if (++spins < 20000) cpu_relax();  // (about a millisecond: the next pivot's job is usually here by then)
    else if (spins < 40000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  void loop()
  {
    int spins = 0;
    for (;;) {
      bool worked = false;
      for (Job& j : job)
        if (j.state.load(std::memory_order_acquire) == 1) {
          try {  // (nothing may leave a thread: a bad_alloc inside the solve's lists ends the ENGINE with status 7, not the process)
            S.ftran(j.rhs, j.rows, *j.out, *j.outlist, false, &W);
          } catch (...) {
            failed.store(1, std::memory_order_release);
          }
          j.state.store(2, std::memory_order_release);
          worked = true;
        }
      if (worked) {
        spins = 0;
        continue;
      }
      if (quit.load(std::memory_order_acquire)) return;
      relax(spins);
    }
  }
  void post(int slot) { job[slot].state.store(1, std::memory_order_release); }
  void wait(int slot)
  {
    int spins = 0;
    while (job[slot].state.load(std::memory_order_acquire) != 2) {  // (the helper is at work on it: no naps on this side)
      if (++spins < 4000) cpu_relax();
      else std::this_thread::yield();
    }
    job[slot].state.store(0, std::memory_order_relaxed);
  }
  ~SolveHelper()
  {
    if (!th.joinable()) return;
    for (int slot = 0; slot < 2; ++slot)
      if (job[slot].state.load(std::memory_order_acquire) == 1) wait(slot);
    quit.store(1, std::memory_order_release);
    th.join();
  }
};

// status: 1 optimal, 2 primal infeasible, 5 iteration limit, 6 time limit, 7 numerical trouble, 9 cancelled.  Continues from
// the basis and the nonbasic bounds S holds (factorised, recomputed, dual feasible).
int run(Simplex& S, int iteration_limit, double time_limit, const std::chrono::steady_clock::time_point& t0, const volatile int32_t* cancel)
{
  const int m = S.m, n = S.n, N = S.N;
  // (w, rho, tau, tvec, col: all zero between pivots -- the solves hand their right-hand sides back zeroed, the results are cleared
  //  through their index lists)
  std::vector<double> alpha(N, 0.0), w(m, 0.0), rho(m, 0.0), tvec(m, 0.0), tau(m, 0.0), col(m, 0.0);
  std::vector<int> touched, astamp(N, -1), wlist, rlist, taulist, tlist, clist;
  std::vector<int> passed(m, 0);  // iteration (+1) at which the position was passed over as "violated by rounding only"
  const double tol_d = 1e-9;
  int since_refactor = 0, sweep = 0;
  int64_t extra_ops = 0;
  // By POSITION, contiguous: the primal infeasibility of each basic variable (0 within the tolerance: 1e-7 relative to the bound;
  // the reference's simplex: 1e-6 absolute) and its steepest-edge weight -- the choice of the leaving position reads these two
  // arrays instead of chasing basic[] into four per-variable ones (0.86 ms -> 0.05 ms per pivot at 100 000 rows).  Kept up to date
  // where a pivot moves a value; rebuilt with the factorisation (positions move there; the weights live per variable in between).
  std::vector<double> pinf(m), bw(m);
  auto infeasibility = [&](int i) {
    const int b     = S.basic[i];
    const double v  = S.z[b];
    const double lo = S.L[b] - v, up = v - S.U[b];
    const double inf = std::max(lo, up);
    return inf > 1e-7 * (1.0 + std::fabs(lo > up ? S.L[b] : S.U[b])) ? inf : 0.0;
  };
  // ... and the positions that ARE infeasible, as a list with lazy removal: the choice walks the list, not all m positions
  std::vector<int> cand;
  std::vector<char> incand(m, 0);
  auto note = [&](int i) {
    if (pinf[i] != 0.0 && !incand[i]) incand[i] = 1, cand.push_back(i);
  };
  auto gather = [&] {
    cand.clear();
    for (int i = 0; i < m; ++i) {
      pinf[i] = infeasibility(i), bw[i] = S.beta[S.basic[i]], incand[i] = 0;
      note(i);
    }
  };
  // The rows of A with the NONBASIC columns first: a pivot row rho^T A_N is needed for nonbasic columns only, and in the later
  // part of a solve most structural columns a row meets are basic.  prj / prv: a copy of the caller's rows, row i's nonbasic entries
  // in [rp[i], pend[i]); where[e] = place of the column-wise entry e in that copy, cent[k] the way back -- a column that enters or
  // leaves the basis swaps its entries across the boundary of their rows (its length in work).  Laid out again from pos[] with
  // every factorisation (which may have replaced dependent columns).
  std::vector<int> prj(S.rj, S.rj + S.rp[m]), pend(m), where((size_t)S.rp[m]), cent((size_t)S.rp[m]);
  std::vector<double> prv(S.rv, S.rv + S.rp[m]);
  {
    std::vector<int> cur(S.cp.begin(), S.cp.end() - 1);  // (the columns were filled by a scan of the rows: same order)
    for (int i = 0; i < m; ++i)
      for (int k = S.rp[i]; k < S.rp[i + 1]; ++k) cent[k] = cur[S.rj[k]]++, where[cent[k]] = k;
  }
  auto swap_entries = [&](int a, int b) {
    if (a == b) return;
    std::swap(prj[a], prj[b]), std::swap(prv[a], prv[b]), std::swap(cent[a], cent[b]);
    where[cent[a]] = a, where[cent[b]] = b;
  };
  auto partition_rows = [&] {
    for (int i = 0; i < m; ++i) {
      int lo = S.rp[i], hi = S.rp[i + 1];
      while (lo < hi) {
        if (S.pos[prj[lo]] < 0) ++lo;
        else swap_entries(lo, --hi);
      }
      pend[i] = lo;
    }
  };
  auto column_to_basic = [&](int j) {  // (structural columns only; the slacks are not stored)
    for (int e = S.cp[j]; e < S.cp[j + 1]; ++e) swap_entries(where[e], --pend[S.ci[e]]);
  };
  auto column_to_nonbasic = [&](int j) {
    for (int e = S.cp[j]; e < S.cp[j + 1]; ++e) swap_entries(where[e], pend[S.ci[e]]++);
  };
  auto rebuild = [&] {
    for (int i = 0; i < m; ++i) S.beta[S.basic[i]] = bw[i];
    S.rebuild(tol_d);
    since_refactor = 0, extra_ops = 0;
    gather();
    partition_rows();
  };
  gather();
  partition_rows();
  const bool use_flips = cuopt_amd::tune_int("simplex_flips", 1) != 0;
  std::vector<int> flips, fwlist, fstamp(m, 0);
  std::vector<double> fw(m, 0.0);
  int fstamp_now = 0;
  std::unique_ptr<SolveHelper> helper;
  if (m >= cuopt_amd::tune_int("simplex_helper_rows", 2000) && cuopt_amd::host_threads() >= 2) {  // (cgroup / affinity aware: under a one-CPU quota the two would only take turns)
    helper.reset(new SolveHelper(S));
    helper->job[0].out = &tau, helper->job[0].outlist = &taulist;
    helper->job[1].out = &fw, helper->job[1].outlist = &fwlist;
    if (!helper->start()) helper.reset();
  }
  bool tau_posted = false, flips_posted = false;
  auto finish_tau = [&] {  // (every way out of a pivot passes here: nothing of the helper's is in flight when the factors change)
    if (tau_posted) helper->wait(0), tau_posted = false;
    if (flips_posted) helper->wait(1), flips_posted = false;
  };
  auto flip_rhs = [&](std::vector<double>& dst, std::vector<int>& rows) {  // sum_j N_j (other bound_j - bound_j) over the flips, by row
    rows.clear();
    ++fstamp_now;
    auto add = [&](int i, double v) {
      if (fstamp[i] != fstamp_now) fstamp[i] = fstamp_now, rows.push_back(i);
      dst[i] += v;
    };
    for (int j : flips) {
      const double dx = S.atU[j] ? S.L[j] - S.U[j] : S.U[j] - S.L[j];
      if (j >= n) add(j - n, -dx);
      else
        for (int k = S.cp[j]; k < S.cp[j + 1]; ++k) add(S.ci[k], S.cv[k] * dx);
    }
  };
  const int refactor_cap = (int)cuopt_amd::tune_int("simplex_refactor", 500);
  std::vector<char> fixedv(N);  // L == U, one byte per variable: the candidate pass of the ratio test reads this instead of two bounds
  for (int j = 0; j < N; ++j) fixedv[j] = S.L[j] == S.U[j];
  std::vector<int> cand_j;  // the ratio test's candidates (sign-eligible entries of the pivot row), compact: column, |alpha|, |d|
  std::vector<double> cand_a, cand_d;
  for (;;) {
    if (S.iterations >= iteration_limit) return 5;
    if (helper && helper->failed.load(std::memory_order_acquire)) return 7;  // the helper's solve threw (out of memory): the engine gives up
    if (flag_set(cancel)) return 9;  // the other engine of a Concurrent solve has finished
    if ((S.iterations & 15) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_limit) return 6;
    // leaving position: the largest primal infeasibility, squared over its steepest-edge weight.  A position whose violation is
    // within 1e-6 and that has no entering candidate is rounding, not a proof of infeasibility (the box bounds put values of 1e6
    // into the basis): it is passed over.
    int r = -1;
    double worst = 0.0, worst_inf = 0.0;
    { Lap lap(S, 0);
    const int skip = S.iterations + 1;
    for (size_t c = 0; c < cand.size();) {
      const int i      = cand[c];
      const double inf = pinf[i];
      if (inf == 0.0) {  // feasible by now: out of the list
        incand[i] = 0, cand[c] = cand.back(), cand.pop_back();
        continue;
      }
      ++c;
      if (passed[i] == skip) continue;
      const double score = S.steepest ? inf * inf / bw[i] : inf;
      if (score > worst || (score == worst && i < r)) worst = score, worst_inf = inf, r = i;  // (ties: the lowest position, whatever the list's order)
    }
    }
    if (r < 0) {
      for (int i = 0; i < m; ++i) S.beta[S.basic[i]] = bw[i];
      return 1;
    }
    const int p        = S.basic[r];
    const bool to_low  = S.z[p] < S.L[p];
    double delta = to_low ? S.L[p] - S.z[p] : S.U[p] - S.z[p];  // change the leaving variable needs
    const double sigma = to_low ? 1.0 : -1.0;
    // row r of the inverse, then of the tableau (from the rows of A that row touches)
    for (int i : rlist) rho[i] = 0.0;
    tvec[r] = 1.0;
    tlist.assign(1, r);
    { Lap lap(S, 1); S.btran(tvec, tlist, rho, rlist); }
    if (helper && S.steepest) {  // tau = B^-1 rho starts now (see SolveHelper); every way out of this pivot passes finish_tau()
      for (int i : taulist) tau[i] = 0.0;
      for (int i : rlist) helper->job[0].rhs[i] = rho[i];
      helper->job[0].rows = rlist;
      helper->post(0), tau_posted = true;
    }
    ++sweep;
    touched.clear();
    double amax = 0.0;
    { Lap lap(S, 2);
    for (int i : rlist) {
      const double ri = rho[i];
      if (std::fabs(ri) < 1e-14) continue;
      for (int k = S.rp[i], ke = pend[i]; k < ke; ++k) {
        const int j = prj[k];
        if (astamp[j] != sweep) astamp[j] = sweep, alpha[j] = 0.0, touched.push_back(j);
        alpha[j] += ri * prv[k];
      }
      if (S.pos[n + i] < 0) astamp[n + i] = sweep, alpha[n + i] = -ri, touched.push_back(n + i);
    }
    }
    cand_j.clear(), cand_a.clear(), cand_d.clear();
    double ptol = 0.0, tmax = kInf;
    { Lap lap(S, 3);
    for (int j : touched) {
      if (fixedv[j]) {  // a fixed variable never enters
        alpha[j] = 0.0;
        continue;
      }
      const double a = sigma * alpha[j];
      amax = std::max(amax, std::fabs(a));
      if (S.atU[j] ? a > 0.0 : a < 0.0) cand_j.push_back(j), cand_a.push_back(std::fabs(a)), cand_d.push_back(std::fabs(S.d[j]));
    }
    ptol = std::max(1e-11, 1e-9 * amax);
    // Harris: pass 1 the largest step that keeps every reduced cost within tol_d of its sign, pass 2 the largest pivot under it
    for (size_t c = 0; c < cand_j.size(); ++c)
      if (cand_a[c] > ptol) tmax = std::min(tmax, (cand_d[c] + tol_d) / cand_a[c]);
    }
    if (tmax == kInf && worst_inf <= 1e-6 * (1.0 + std::fabs(to_low ? S.L[p] : S.U[p]))) {
      passed[r] = S.iterations + 1;
      finish_tau();
      continue;
    }
    if (tmax == kInf) {  // no entering variable: the row proves primal infeasibility
      finish_tau();
      if (since_refactor != 0) {  // ... if a fresh factorisation says so too
        rebuild();
        continue;
      }
      // The row bounds z_p over the box: z_p + sum_j |alpha_j| * (room of j in its helping direction) stays short of the violated
      // bound.  That is a proof for the LP itself only if it does not rest on an ARTIFICIAL bound: the violated bound of p and the
      // bound every helping nonbasic variable is stopped by must be bounds of the LP (round-3 advisor: min x s.t. 1e-7 x >= 1 was
      // "infeasible" inside the 1e5 box).  Resting on the box -> 10: the caller widens the box or abstains.  Entries below the pivot
      // tolerance were left out of the ratio test; if what they could still add reaches the violation, the row proves nothing -> 7.
      bool on_box  = to_low ? S.boxedL[p] != 0 : S.boxedU[p] != 0;
      double reach = 0.0;
      for (int j : touched) {
        const double a = sigma * alpha[j];
        if (a == 0.0) continue;
        const bool up     = a < 0.0;  // the direction of j that moves z_p towards its bound
        const double room = up ? S.U[j] - S.z[j] : S.z[j] - S.L[j];
        const bool boxed  = up ? S.boxedU[j] != 0 : S.boxedL[j] != 0;
        // a helping direction that only the box ends (at the bound already, or -- an entry under the pivot tolerance -- on its way
        // there): in the LP itself that variable moves z_p as far as it likes.  (1e-12 amax: above the rounding of the row.)
        if (boxed && std::fabs(a) > 1e-12 * amax) on_box = true;
        if (room > 0.0) reach += std::fabs(a) * room;  // (only entries under the pivot tolerance still have room)
      }
      if (S.debug)
        std::fprintf(stderr, "[simplex] infeasible position %d var %d value %.12g bounds [%.6g, %.6g] amax %.3g ptol %.3g: rests on the box %d, reach of the small entries %.3g of %.3g\n", r, p, S.z[p], S.L[p], S.U[p], amax, ptol, (int)on_box, reach, worst_inf);
      if (on_box) return 10;
      if (reach >= 0.5 * worst_inf) return 7;
      return 2;
    }
    // Bound flipping (the long-step rule; the reference's simplex: phase2.cpp:348-470): the row's infeasibility |delta| is the slope of
    // the dual objective along this step; passing the breakpoint of candidate j costs |alpha_j| (U_j - L_j) of it.  While the slope
    // stays positive behind a whole Harris group, the group is FLIPPED to its other bounds instead of one of it entering, and the
    // test moves on to the next group: one pivot does the work of several.  Not across a group that holds a degenerate breakpoint
    // (|d_j| within the tolerance: passing those for free is what cycles on dual degenerate LPs), and never ONTO an artificial
    // bound (in the LP itself that range is infinite).
    flips.clear();
    if (use_flips) {
      Lap lap(S, 3);
      double slope = std::fabs(delta);
      for (int passes = 0; passes < 64; ++passes) {
        double sum = 0.0, tnext = kInf;
        bool stop = false;
        for (size_t c = 0; c < cand_j.size() && !stop; ++c) {
          if (cand_a[c] <= ptol) continue;
          if (cand_d[c] / cand_a[c] <= tmax) {
            const int j = cand_j[c];
            if (cand_d[c] <= tol_d || (S.atU[j] ? S.boxedL[j] != 0 : S.boxedU[j] != 0)) stop = true;
            sum += cand_a[c] * (S.U[j] - S.L[j]);
          } else {
            tnext = std::min(tnext, (cand_d[c] + tol_d) / cand_a[c]);
          }
        }
        if (stop || tnext == kInf || slope - sum < 0.0) break;  // the entering variable comes from this group
        for (size_t c = 0; c < cand_j.size(); ++c)
          if (cand_a[c] > ptol && cand_d[c] / cand_a[c] <= tmax) flips.push_back(cand_j[c]), cand_a[c] = 0.0;  // out of the test
        slope -= sum, tmax = tnext;
        if (S.debug) ++S.bf_rounds;
      }
    }
    int q        = -1;
    double apick = 0.0;
    { Lap lap(S, 3);
    for (size_t c = 0; c < cand_j.size(); ++c)
      if (cand_a[c] > ptol && cand_a[c] > apick && cand_d[c] / cand_a[c] <= tmax) apick = cand_a[c], q = cand_j[c];
    }
    if (q < 0) {
      finish_tau();
      return 7;
    }
    if (helper && !flips.empty()) {  // the basic variables' answer to the flips: the helper's second job (behind tau)
      for (int i : fwlist) fw[i] = 0.0;
      flip_rhs(helper->job[1].rhs, helper->job[1].rows);
      helper->post(1), flips_posted = true;
    }
    // entering column
    for (int i : wlist) w[i] = 0.0;
    clist.clear();
    if (q >= n) col[q - n] = -1.0, clist.push_back(q - n);
    else
      for (int k = S.cp[q]; k < S.cp[q + 1]; ++k) col[S.ci[k]] += S.cv[k], clist.push_back(S.ci[k]);
    { Lap lap(S, 4); S.ftran(col, clist, w, wlist, true); }
    if (std::fabs(w[r]) < 1e-11 || std::fabs(w[r] - alpha[q]) > 1e-6 * (1.0 + std::fabs(alpha[q]))) {
      // the factorisation has drifted: rebuild it and look again
      finish_tau();
      if (since_refactor == 0) return 7;
      rebuild();
      continue;
    }
    // steepest-edge weights: beta_i += kappa_i (kappa_i beta_r - 2 tau_i), kappa_i = w_i / w_r, tau = B^-1 rho
    if (S.steepest) {
      Lap lap(S, 5);
      double br = 0.0;
      for (int i : rlist) br += rho[i] * rho[i];
      S.beta[p] = br;  // exact, whatever the updates had made of it
      if (tau_posted) {
        helper->wait(0), tau_posted = false;
      } else {
        for (int i : taulist) tau[i] = 0.0;
        for (int i : rlist) col[i] = rho[i];
        S.ftran(col, rlist, tau, taulist);
      }
      const double wr = w[r];
      for (int i : wlist) {
        if (i == r || w[i] == 0.0) continue;
        const double kap = w[i] / wr;
        bw[i]            = std::max(bw[i] + kap * (kap * br - 2.0 * tau[i]), 1e-4);
      }
      bw[r] = std::max(br / (wr * wr), 1e-4);  // (the entering variable takes the position over below)
    } else {
      bw[r] = 1.0;
    }
    // duals: d_j -= theta alpha_rj, the entering variable's becomes 0, the leaving one's -theta
    Lap lap_values(S, 8);
    const double theta = S.d[q] / alpha[q];
    for (int j : touched)
      if (alpha[j] != 0.0) S.d[j] -= theta * alpha[j];
    S.d[q] = 0.0;
    S.d[p] = -theta;
    // the flipped variables jump to their other bounds (their reduced costs have just changed sign); the basic ones follow:
    // z_B -= B^-1 sum_j N_j (new_j - old_j), the leaving variable among them -- the slope left says it is still short of its bound
    if (!flips.empty()) {
      if (flips_posted) {
        helper->wait(1), flips_posted = false;
      } else {
        for (int i : fwlist) fw[i] = 0.0;
        flip_rhs(col, clist);
        Lap lap(S, 4);
        S.ftran(col, clist, fw, fwlist);
      }
      for (int j : flips) S.atU[j] = !S.atU[j], S.z[j] = S.atU[j] ? S.U[j] : S.L[j];
      for (int i : fwlist)
        if (fw[i] != 0.0) S.z[S.basic[i]] -= fw[i], pinf[i] = infeasibility(i), note(i);
      delta = to_low ? S.L[p] - S.z[p] : S.U[p] - S.z[p];
      if (S.debug) ++S.bf_pass, S.bf_grp += (int64_t)flips.size();
    }
    // primal: the entering variable moves by step, the basic ones by -w step
    const double step = -delta / w[r];
    S.z[q] += step;
    S.z[p]   = to_low ? S.L[p] : S.U[p];
    S.atU[p] = !to_low;
    { Lap lap(S, 9); S.push_update(r, w[r]); }
    S.pos[p] = -1, S.pos[q] = r, S.basic[r] = q;
    if (q < n) column_to_basic(q);
    if (p < n) column_to_nonbasic(p);
    for (int i : wlist)
      if (w[i] != 0.0 && i != r) S.z[S.basic[i]] -= w[i] * step, pinf[i] = infeasibility(i), note(i);
    pinf[r] = infeasibility(r), note(r);
    S.iterations += 1;
    if (S.debug) ++S.dens_pivots, S.dens[0] += rlist.size(), S.dens[1] += wlist.size(), S.dens[2] += taulist.size(), S.dens[3] += touched.size(), S.dens[4] += S.update_entries(), S.dens[5] += S.Li.size() + S.Ui.size();
    // a fresh factorisation when the update file has cost as much as one costs (every solve walks the whole file), at the latest
    // after kRefactorEvery pivots
    extra_ops += (S.steepest ? 3 : 2) * (int64_t)S.update_entries();
    if (S.debug) S.ops_solve += (S.steepest ? 3 : 2) * ((int64_t)S.update_entries() + (int64_t)S.Li.size() + (int64_t)S.Ui.size() + 2 * (int64_t)m);
    // (the factorisation's count is of entries touched; its depth-first searches and pivot choices make an entry cost ~8 times
    // what one costs in a solve: calibrated on a 10 000-row block-angular LP, 79 s -> 58 s; the multiply-adds of the nucleus'
    // elimination run over contiguous columns and cost about two solve entries each -- counted apart since a random 4000 x 3000
    // LP, whose nucleus fills in to 400 000 entries, spent 56 % of its time refactorising at a cap of 100 pivots: 15.5 -> 10.9 s)
    const int64_t rebuild_ops = 8 * S.factor_ops + 2 * S.factor_flops + 2 * (int64_t)S.cp[n] + 4 * ((int64_t)S.Li.size() + (int64_t)S.Ui.size()) + 8 * (int64_t)m;
    if (++since_refactor >= refactor_cap || extra_ops >= rebuild_ops) rebuild();
  }
}

// slack basis: B = -I, dual feasible by the choice of the nonbasic bounds (every bound is finite after boxing)
void start_from_slacks(Simplex& S)
{
  const int m = S.m, n = S.n, N = S.N;
  S.basic.resize(m), S.pos.assign(N, -1), S.atU.assign(N, 0);
  S.z.assign(N, 0.0), S.d.assign(N, 0.0), S.y.assign(m, 0.0), S.beta.assign(N, 1.0);
  for (int j = 0; j < n; ++j) {
    S.atU[j] = S.g[j] < 0.0;
    S.z[j]   = S.atU[j] ? S.U[j] : S.L[j];
  }
  std::vector<int> cand(m);
  for (int r = 0; r < m; ++r) cand[r] = n + r;
  S.factor(cand, nullptr);
  S.recompute();
}
// ---- primal simplex (bounded, phase 2): from a primal feasible basis to an optimal one.  Used behind a start from another
// engine's point, where the guessed basis is (nearly) primal feasible and dual infeasible -- the other way round from what the
// dual simplex above needs.  Pricing: the largest dual infeasibility squared over a static column weight; Harris' two-pass ratio
// test with bound flips; the reduced costs are updated from the pivot row like the dual simplex's.
// status: 1 optimal (for the bounds in force), 5 / 6 limits, 7 numerical trouble / stalled, 9 cancelled.
int run_primal(Simplex& S, int iteration_limit, double time_limit, const std::chrono::steady_clock::time_point& t0, const volatile int32_t* cancel)
{
  const int m = S.m, n = S.n, N = S.N;
  std::vector<double> alpha(N, 0.0), w(m, 0.0), rho(m, 0.0), tvec(m, 0.0), col(m, 0.0), gamma(N, 2.0);
  std::vector<int> touched, astamp(N, -1), wlist, rlist, tlist, clist;
  for (int j = 0; j < n; ++j) {
    double s = 1.0;
    for (int k = S.cp[j]; k < S.cp[j + 1]; ++k) s += S.cv[k] * S.cv[k];
    gamma[j] = s;
  }
  const double tol_d = 1e-9;
  int since_refactor = 0, sweep = 0, stalled = 0;
  int64_t extra_ops = 0;
  for (;;) {
    if (S.iterations >= iteration_limit) return 5;
    if (flag_set(cancel)) return 9;
    if ((S.iterations & 15) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_limit) return 6;
    // entering variable
    int q = -1;
    double best = 0.0;
    for (int j = 0; j < N; ++j) {
      if (S.pos[j] >= 0 || S.L[j] == S.U[j]) continue;
      const double inf = S.atU[j] ? S.d[j] : -S.d[j];
      if (inf <= tol_d * (1.0 + std::fabs(S.g[j]))) continue;
      const double score = inf * inf / gamma[j];
      if (score > best) best = score, q = j;
    }
    if (q < 0) return 1;
    const double dir = S.atU[q] ? -1.0 : 1.0;  // the entering variable moves up from its lower bound, down from its upper one
    for (int i : wlist) w[i] = 0.0;
    clist.clear();
    if (q >= n) col[q - n] = -1.0, clist.push_back(q - n);
    else
      for (int k = S.cp[q]; k < S.cp[q + 1]; ++k) col[S.ci[k]] += S.cv[k], clist.push_back(S.ci[k]);
    S.ftran(col, clist, w, wlist, true);
    // ratio test: basic i moves by -dir t w_i
    double wmax = 0.0;
    for (int i : wlist) wmax = std::max(wmax, std::fabs(w[i]));
    const double ptol = std::max(1e-11, 1e-9 * wmax);
    double tmax = S.U[q] - S.L[q];  // its own other bound: a flip
    for (int i : wlist) {
      const double a = dir * w[i];
      if (std::fabs(a) <= ptol) continue;
      const int b      = S.basic[i];
      const double gap = a > 0.0 ? S.z[b] - S.L[b] : S.U[b] - S.z[b];
      const double bnd = a > 0.0 ? S.L[b] : S.U[b];
      tmax = std::min(tmax, (std::max(gap, 0.0) + 1e-9 * (1.0 + std::fabs(bnd))) / std::fabs(a));
    }
    int r        = -1;
    double apick = 0.0, step = S.U[q] - S.L[q];
    for (int i : wlist) {
      const double a = dir * w[i];
      if (std::fabs(a) <= ptol) continue;
      const int b      = S.basic[i];
      const double gap = std::max(a > 0.0 ? S.z[b] - S.L[b] : S.U[b] - S.z[b], 0.0);
      if (gap / std::fabs(a) <= tmax && std::fabs(a) > apick) apick = std::fabs(a), r = i, step = gap / std::fabs(a);
    }
    if (r < 0 || step >= S.U[q] - S.L[q]) {  // the entering variable reaches its other bound first: no basis change
      step = S.U[q] - S.L[q];
      for (int i : wlist)
        if (w[i] != 0.0) S.z[S.basic[i]] -= dir * step * w[i];
      S.atU[q] = !S.atU[q];
      S.z[q]   = S.atU[q] ? S.U[q] : S.L[q];
      S.iterations += 1;
      continue;
    }
    stalled = step <= 0.0 ? stalled + 1 : 0;
    if (stalled > 50 * (m + 100)) return 7;  // (degenerate pivots only, for a long time)
    const int p = S.basic[r];
    // pivot row for the reduced costs
    for (int i : rlist) rho[i] = 0.0;
    tvec[r] = 1.0;
    tlist.assign(1, r);
    S.btran(tvec, tlist, rho, rlist);
    ++sweep;
    touched.clear();
    for (int i : rlist) {
      const double ri = rho[i];
      if (std::fabs(ri) < 1e-14) continue;
      for (int k = S.rp[i]; k < S.rp[i + 1]; ++k) {
        const int j = S.rj[k];
        if (S.pos[j] >= 0) continue;
        if (astamp[j] != sweep) astamp[j] = sweep, alpha[j] = 0.0, touched.push_back(j);
        alpha[j] += ri * S.rv[k];
      }
      if (S.pos[n + i] < 0) astamp[n + i] = sweep, alpha[n + i] = -ri, touched.push_back(n + i);
    }
    const double arq = astamp[q] == sweep ? alpha[q] : 0.0;
    if (std::fabs(w[r]) < 1e-11 || std::fabs(w[r] - arq) > 1e-6 * (1.0 + std::fabs(arq))) {
      if (since_refactor == 0) return 7;
      S.rebuild_plain();
      since_refactor = 0, extra_ops = 0;
      continue;
    }
    const double theta = S.d[q] / w[r];
    for (int j : touched)
      if (alpha[j] != 0.0) S.d[j] -= theta * alpha[j];
    S.d[q] = 0.0;
    S.d[p] = -theta;
    for (int i : wlist)
      if (w[i] != 0.0) S.z[S.basic[i]] -= dir * step * w[i];
    S.z[q] += dir * step;
    const bool to_low = dir * w[r] > 0.0;
    S.z[p]   = to_low ? S.L[p] : S.U[p];
    S.atU[p] = !to_low;
    { Lap lap(S, 9); S.push_update(r, w[r]); }
    S.pos[p] = -1, S.pos[q] = r, S.basic[r] = q;
    S.iterations += 1;
    extra_ops += 2 * (int64_t)S.update_entries();
    const int64_t rebuild_ops = 8 * S.factor_ops + 2 * S.factor_flops + 2 * (int64_t)S.cp[n] + 4 * ((int64_t)S.Li.size() + (int64_t)S.Ui.size()) + 8 * (int64_t)m;
    if (++since_refactor >= kRefactorEvery || extra_ops >= rebuild_ops) {
      S.rebuild_plain();
      since_refactor = 0, extra_ops = 0;
    }
  }
}

// A basis guessed from a point (x0, y0) -- the crossover of a first-order solution: the variables (structural and slack) that sit
// clearly between their bounds, the most interior first (score = relative room to the nearer bound - relative size of the reduced
// cost at y0: a variable with a visible reduced cost is nonbasic however much room the point's accuracy leaves it); the
// factorisation completes them with slacks and drops dependent ones.  The others go to their nearer bound.  Such a basis is close
// to primal feasible and NOT dual feasible (a first-order method's duals sit in the middle of the dual optimal face, not at a
// vertex): what is outside its bounds gets the bound moved to where it is, the PRIMAL simplex pivots to an optimal basis for
// those bounds, the bounds go back, and the caller's dual simplex removes what infeasibility that leaves.
// Returns the primal simplex's status (1: the basis is dual feasible for the true bounds).
int start_from_point(Simplex& S, const double* x0, const double* y0, double /*sense*/, int iteration_limit, double time_limit,
                     const std::chrono::steady_clock::time_point& t0, const volatile int32_t* cancel)
{
  const int m = S.m, n = S.n, N = S.N;
  S.pos.assign(N, -1), S.atU.assign(N, 0);
  S.z.assign(N, 0.0), S.d.assign(N, 0.0), S.y.assign(m, 0.0), S.beta.assign(N, 1.0);
  for (int j = 0; j < n; ++j) S.z[j] = x0[j];
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int k = S.rp[i]; k < S.rp[i + 1]; ++k) s += S.rv[k] * x0[S.rj[k]];
    S.z[n + i] = s;
  }
  std::vector<double> d0(N, 0.0);
  double cmax = 0.0;
  for (int j = 0; j < n; ++j) cmax = std::max(cmax, std::fabs(S.g[j]));
  if (y0) {
    std::vector<double> yi(m);
    for (int i = 0; i < m; ++i) yi[i] = y0[i], d0[n + i] = yi[i];  // (y0: duals of the converted minimisation, like every dual this library returns)
    for (int j = 0; j < n; ++j) d0[j] = S.g[j] - S.col_dot(yi.data(), j);
  }
  std::vector<std::pair<double, int>> inside;
  std::vector<int> onbound;
  for (int j = 0; j < N; ++j) {
    const double room  = std::max(0.0, std::min(S.z[j] - S.L[j], S.U[j] - S.z[j])) / (1.0 + std::fabs(S.z[j]));
    const double score = std::min(room, 1.0) - std::min(std::fabs(d0[j]) / (1.0 + cmax), 1.0);
    if (score > 1e-3) inside.emplace_back(-score, j);
    else if (y0 && std::fabs(d0[j]) <= 1e-7 * (1.0 + cmax)) onbound.push_back(j);  // no room, no reduced cost: degenerate basic at a vertex (y0 from a simplex)
  }
  std::sort(inside.begin(), inside.end());
  std::vector<int> cand;
  for (size_t t = 0; t < inside.size() && (int)cand.size() < m; ++t) cand.push_back(inside[t].second);
  const int nprio = (int)cand.size();
  for (size_t t = 0; t < onbound.size() && (int)cand.size() < m; ++t) cand.push_back(onbound[t]);
  for (int j = 0; j < N; ++j) {
    S.atU[j] = std::fabs(S.U[j] - S.z[j]) < std::fabs(S.z[j] - S.L[j]);
    S.z[j]   = S.atU[j] ? S.U[j] : S.L[j];
  }
  S.basic.assign(m, 0);
  std::vector<int> rejected;
  S.factor(cand, &rejected, nprio);  // (rejected candidates are nonbasic at their nearer bound already)
  S.recompute();
  // bounds out of the way of what is infeasible
  const std::vector<double> L_true(S.L), U_true(S.U);
  int moved = 0;
  for (int k = 0; k < m; ++k) {
    const int b = S.basic[k];
    if (S.z[b] < S.L[b]) S.L[b] = S.z[b], ++moved;
    if (S.z[b] > S.U[b]) S.U[b] = S.z[b], ++moved;
  }
  if (S.debug) {
    int wrong = 0;
    for (int j = 0; j < N; ++j) wrong += S.pos[j] < 0 && S.L[j] != S.U[j] && ((S.d[j] < -1e-9 && !S.atU[j]) || (S.d[j] > 1e-9 && S.atU[j]));
    std::fprintf(stderr, "[simplex] start from a point: %d variables clearly inside their bounds + %zu on a bound without a reduced cost, %zu turned away as dependent, %d basic values outside their bounds, %d reduced costs on the wrong side\n",
                 nprio, cand.size() - (size_t)nprio, rejected.size(), moved, wrong);
  }
  S.iterations   = 0;
  const int code = run_primal(S, iteration_limit, time_limit, t0, cancel);
  if (S.debug) std::fprintf(stderr, "[simplex] primal simplex from that basis: status %d after %d pivots / flips\n", code, S.iterations);
  S.L = L_true, S.U = U_true;
  for (int j = 0; j < N; ++j)
    if (S.pos[j] < 0) S.z[j] = S.atU[j] ? S.U[j] : S.L[j];
  S.rebuild(1e-9);
  return code;
}

int solve_core(const cuoptamd_lp* lp, const double* x0, const double* y0, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
               int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc);

// The engine behind a presolve (simplex_presolve.hpp: empty rows and columns, singleton rows, fixed columns) and the way back.
// CUOPT_AMD_TUNE="simplex_presolve=0" solves the LP as it comes.
int solve(const cuoptamd_lp* lp, const double* x0, const double* y0, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
          int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc)
{
  if (!lp || !status) return -1;
  const int m = lp->m, n = lp->n;
  // (an LP beyond the engine's limits is turned away by solve_core as it comes: no presolve is spent on it)
  int64_t max_rows = 200000, max_nnz = 4000000;
  if (const char* e = std::getenv("CUOPT_AMD_SIMPLEX_MAX_ROWS")) max_rows = std::atoll(e);
  if (const char* e = std::getenv("CUOPT_AMD_SIMPLEX_MAX_NNZ")) max_nnz = std::atoll(e);
  if (m <= 0 || n <= 0 || m > max_rows || (int64_t)n + m > 20 * max_rows || lp->offsets[m] > max_nnz || cuopt_amd::tune_int("simplex_presolve", 1) == 0)
    return solve_core(lp, x0, y0, time_limit, iteration_limit, cancel, status, iterations, objective, x, y, rc);
  cuopt_amd::SimplexPresolve P;
  const auto t_pre = std::chrono::steady_clock::now();
  const bool reduced = P.run(lp, cancel);
  // (the presolve's own time counts against the caller's limit: two column-wise copies of a 4e6-nonzero LP are not free)
  if (time_limit > 0.0) {
    const double spent = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pre).count();
    if (spent >= time_limit) {
      *status = 6;
      if (iterations) *iterations = 0;
      return 0;
    }
    time_limit -= spent;
  }
  if (!reduced) return solve_core(lp, x0, y0, time_limit, iteration_limit, cancel, status, iterations, objective, x, y, rc);
  if (P.cancelled) {
    *status = 9;
    if (iterations) *iterations = 0;
    return 0;
  }
  const bool debug = cuopt_amd::tune_int("simplex_debug", 0) != 0;
  if (debug)
    std::fprintf(stderr, "[simplex] presolve: %d of %d rows and %d of %d columns removed%s\n", P.removed_rows, m, P.removed_cols, n, P.infeasible ? ", infeasible" : "");
  if (iterations) *iterations = 0;
  if (P.infeasible) {
    *status = 2;
    return 0;
  }
  if (P.reduced.m == 0 && P.reduced.n > 0)  // columns whose cost points to an infinite bound are all that is left: the engine's call
    return solve_core(lp, x0, y0, time_limit, iteration_limit, cancel, status, iterations, objective, x, y, rc);
  std::vector<double> px((size_t)n, 0.0), py((size_t)m, 0.0), prc((size_t)n, 0.0);
  if (P.reduced.n == 0) {
    *status = 1;  // nothing left to decide
  } else {
    std::vector<double> cx0, cy0;
    if (x0) {
      for (int j : P.cols) cx0.push_back(x0[j]);
      if (y0)
        for (int i : P.rows) cy0.push_back(y0[i]);
    }
    const int ret = solve_core(&P.reduced, x0 ? cx0.data() : nullptr, x0 && y0 ? cy0.data() : nullptr, time_limit, iteration_limit, cancel, status,
                               iterations, nullptr, px.data(), py.data(), prc.data());
    if (ret != 0 || *status != 1) return ret;
  }
  const double obj = P.undo(lp, px.data(), py.data(), prc.data());
  if (objective) *objective = obj;
  if (x) std::copy(px.begin(), px.end(), x);
  if (y) std::copy(py.begin(), py.end(), y);
  if (rc) std::copy(prc.begin(), prc.end(), rc);
  return 0;
}

int solve_core(const cuoptamd_lp* lp, const double* x0, const double* y0, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
               int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc)
{
  if (!lp || !status) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  const int m = lp->m, n = lp->n;
  const int64_t nnz = m > 0 ? lp->offsets[m] : 0;
  *status = 8;  // too large (or empty): the caller keeps to PDLP
  if (iterations) *iterations = 0;
  int64_t max_rows = 200000, max_nnz = 4000000;
  if (const char* e = std::getenv("CUOPT_AMD_SIMPLEX_MAX_ROWS")) max_rows = std::atoll(e);
  if (const char* e = std::getenv("CUOPT_AMD_SIMPLEX_MAX_NNZ")) max_nnz = std::atoll(e);
  if (m <= 0 || n <= 0 || m > max_rows || (int64_t)n + m > 20 * max_rows || nnz > max_nnz) return 0;
  Simplex S;
  S.m = m, S.n = n, S.N = n + m;
  S.rp = lp->offsets, S.rj = lp->indices, S.rv = lp->values;
  S.cancel = cancel;
  {  // CUOPT_AMD_TUNE="simplex_pricing=dantzig|steepest,simplex_solves=dense|sparse|auto" (tests compare the variants' pivots)
    std::string e;
    if (cuopt_amd::tune_get("simplex_pricing", &e)) S.steepest = e != "dantzig";
    if (cuopt_amd::tune_get("simplex_solves", &e)) S.sparse_solve = e == "dense" ? 0 : e == "sparse" ? 2 : 1;
  }
  // columns of A
  // (the other engine of a Concurrent solve may be done before this one has even copied the matrix: the flag is looked at here too)
  auto cancelled = [&] { return flag_set(cancel); };
  S.cp.assign(n + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    if ((k & 0xFFFFF) == 0 && cancelled()) { *status = 9; return 0; }
    S.cp[lp->indices[k] + 1]++;
  }
  for (int j = 0; j < n; ++j) S.cp[j + 1] += S.cp[j];
  S.ci.resize((size_t)nnz), S.cv.resize((size_t)nnz);
  {
    std::vector<int32_t> cur(S.cp.begin(), S.cp.end() - 1);
    for (int i = 0; i < m; ++i) {
      if ((i & 0xFFFF) == 0 && cancelled()) { *status = 9; return 0; }
      for (int k = lp->offsets[i]; k < lp->offsets[i + 1]; ++k) {
        const int q = cur[lp->indices[k]]++;
        S.ci[q] = i, S.cv[q] = lp->values[k];
      }
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
  bool have_first = false;
  if (time_limit <= 0.0 || !std::isfinite(time_limit)) time_limit = 1e30;
  if (iteration_limit <= 0) iteration_limit = std::numeric_limits<int32_t>::max();
  const bool debug = cuopt_amd::tune_int("simplex_debug", 0) != 0;
  S.debug = debug;
  int total_iterations = 0;
  auto print_seconds = [&] {
    const double np = (double)std::max<int64_t>(1, S.dens_pivots);
    std::fprintf(stderr, "[simplex] per pivot: rho %.0f, w %.0f, tau %.0f, touched %.0f, update entries %.0f, L+U %.0f (m %d)\n", S.dens[0] / np, S.dens[1] / np, S.dens[2] / np, S.dens[3] / np, S.dens[4] / np, S.dens[5] / np, S.m);
    std::fprintf(stderr, "[simplex] pivots with bound flips: %lld (flips %lld, groups passed %lld)\n", (long long)S.bf_pass, (long long)S.bf_grp, (long long)S.bf_rounds);
    std::fprintf(stderr, "[simplex] nucleus s: columns through the triangular part %.2f, pivot search %.2f, elimination %.2f\n", S.nsec[0], S.nsec[1], S.nsec[2]);
    std::fprintf(stderr, "[simplex] factorisation ms: ordering %.0f, set-up %.0f, triangular part %.0f, nucleus %.0f, row-wise copies %.0f\n", S.fsec[0], S.fsec[1], S.fsec[2], S.fsec[3], S.fsec[4]);
    std::fprintf(stderr, "[simplex] seconds: pricing %.2f, btran %.2f, pivot row %.2f, ratio test %.2f, ftran %.2f, weights %.2f, values %.2f (update file %.2f), rebuilds %.2f (factorisations %.2f); ns per counted entry: factorisation %.2f, solves %.2f\n", S.tsec[0], S.tsec[1], S.tsec[2], S.tsec[3], S.tsec[4], S.tsec[5], S.tsec[8], S.tsec[9], S.tsec[7], S.tsec[6], 1e9 * S.tsec[6] / std::max<int64_t>(S.ops_factor, 1), 1e9 * (S.tsec[1] + S.tsec[4] + S.tsec[5]) / std::max<int64_t>(S.ops_solve, 1));
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    const double big = (attempt == 0 ? (double)cuopt_amd::tune_int("simplex_box", 100000) : 1e8) * scale;
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
    if (attempt == 0) {
      if (x0) {
        const int pc = start_from_point(S, x0, y0, sense, iteration_limit, time_limit, t0, cancel);
        total_iterations += S.iterations;
        if (pc == 6 || pc == 9 || pc == 5) {
          *status = pc;
          if (iterations) *iterations = total_iterations;
          return 0;
        }
      } else {
        start_from_slacks(S);
      }
    } else {
      // the wider box: same basis, the nonbasic variables follow their (box) bounds out
      for (int j = 0; j < S.N; ++j)
        if (S.pos[j] < 0) S.z[j] = S.atU[j] ? S.U[j] : S.L[j];
      S.rebuild(1e-9);
    }
    if (debug) {
      int structurals = 0;
      for (int k = 0; k < m; ++k) structurals += S.basic[k] < n;
      std::fprintf(stderr, "[simplex] attempt %d box %.3g: start basis with %d structural columns, L %zu + U %zu off-diagonal entries\n", attempt, big, structurals, S.Li.size(), S.Ui.size());
    }
    // (a final fresh factorisation may repair the basis or find rounding-size infeasibilities: then the loop goes on)
    for (int pass = 0; pass < 4; ++pass) {
      S.iterations = 0;
      code         = run(S, iteration_limit - total_iterations, time_limit, t0, cancel);
      total_iterations += S.iterations;
      if (code != 1) break;
      const std::vector<int> before(S.basic);
      S.rebuild(1e-9);
      std::vector<int> a(before), b(S.basic);
      std::sort(a.begin(), a.end()), std::sort(b.begin(), b.end());
      bool clean = a == b;
      for (int k = 0; k < m && clean; ++k) {
        const int v = S.basic[k];
        clean = S.z[v] >= S.L[v] - 1e-6 * (1.0 + std::fabs(S.L[v])) && S.z[v] <= S.U[v] + 1e-6 * (1.0 + std::fabs(S.U[v]));
      }
      if (clean) break;
      if (pass == 3) code = 7;
    }
    if (iterations) *iterations = total_iterations;
    if (code == 10) {  // "infeasible" only because of the artificial box: the wider one decides, from the same basis; after that: abstain
      if (attempt == 0) continue;
      code = 7;
    }
    if (code != 1) break;
    // does the vertex lean on a box bound?
    bool leans = false;
    for (int j = 0; j < S.N && !leans; ++j) {
      const double tol = 1e-6 * big;
      leans = (S.boxedL[j] && S.z[j] <= S.L[j] + tol) || (S.boxedU[j] && S.z[j] >= S.U[j] - tol);
    }
    double obj = 0.0;
    for (int j = 0; j < n; ++j) obj += S.g[j] * S.z[j];
    if (debug) {
      std::fprintf(stderr, "[simplex] attempt %d box %.3g: objective %.17g, leans %d, iterations %d, %.3f s, x =", attempt, big, obj, (int)leans, total_iterations,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      for (int j = 0; j < std::min(n, 8); ++j) std::fprintf(stderr, " %.6g", S.z[j]);
      std::fprintf(stderr, "\n");
      print_seconds();
    }
    if (!leans) break;
    if (attempt == 1) {
      // still on the (1000 times wider) box: how fast did the objective follow it out, per unit of box and of cost?
      //   clearly (> 1e-4): unbounded;  not at all (< 1e-12): a ray of alternative optima, the first, smaller vertex is the answer;
      //   in between the ray's cost is inside any dual tolerance (datasets/mip/minrep_inf.mps: 2.5e-7 -- the reference's simplex
      //   calls that LP optimal, HiGHS unbounded): this engine abstains (numerical trouble) and the caller's PDLP answers
      double cmax = 0.0;
      for (int j = 0; j < n; ++j) cmax = std::max(cmax, std::fabs(S.g[j]));
      if (!have_first) {  // the smaller box ended without a vertex: nothing to compare with
        code = 7;
        break;
      }
      const double rate = (prev_obj - obj) / (big * std::max(cmax, 1e-300));
      if (rate > 1e-4) {
        // Unbounded only with a certificate: the direction between the two vertices must be a ray of the LP itself -- no finite
        // bound of a variable or a row in its way, the objective falling along it.  Otherwise (an optimum beyond the wider box, for
        // instance) the engine abstains and PDLP answers.
        // The direction: how this basis' vertex moves when the box grows (the nonbasic variables on artificial bounds follow it, the
        // basic ones answer linearly) -- an exact derivative, so the test can be strict (1e-12): 1e-9 x <= 1 stops x at 1e9.
        const std::vector<double> z_at(S.z);
        for (int j = 0; j < S.N; ++j)
          if (S.pos[j] < 0 && (S.atU[j] ? S.boxedU[j] : S.boxedL[j])) S.z[j] = 2.0 * z_at[j];
        S.recompute();
        std::vector<double> dir(S.N);
        double dmax = 0.0, cd = 0.0;
        for (int j = 0; j < S.N; ++j) dir[j] = (S.z[j] - z_at[j]) / big, dmax = std::max(dmax, std::fabs(dir[j]));
        S.z = z_at;
        S.recompute();
        bool ray = dmax > 0.0;
        for (int j = 0; j < S.N && ray; ++j) {
          const double dj = dir[j] / dmax;
          if (!S.boxedL[j] && dj < -1e-12) ray = false;
          if (!S.boxedU[j] && dj > 1e-12) ray = false;
          if (j < n) cd += S.g[j] * dj;
        }
        code = ray && cd < -1e-9 * std::max(cmax, 1e-300) ? 3 : 7;
      } else if (rate < 1e-12) {
        S.z = first_z, S.y = first_y, S.d = first_d, S.pos = first_pos;
      } else {
        code = 7;
      }
      break;
    }
    prev_obj = obj, have_first = true;
    first_z = S.z, first_y = S.y, first_d = S.d, first_pos = S.pos;
  }
  *status = code;
  if (iterations) *iterations = total_iterations;
  if (debug && code != 1) print_seconds();
  if (code == 1) {
    double obj = 0.0;
    for (int j = 0; j < n; ++j) obj += lp->c[j] * S.z[j];
    if (objective) *objective = obj + lp->objective_offset;
    if (x) std::copy(S.z.begin(), S.z.begin() + n, x);
    // duals / reduced costs of the CONVERTED minimisation (a maximisation is solved as the minimisation of -c), unflipped: the
    // convention of the reference's two engines (dual_simplex/solve.cpp:256 hands lp_solution.y on as it is) and of the PDLP path
    // here (pdlp_solver.cpp negates c on the host and returns y, rc of that problem) -- one sign whichever engine answers
    if (y)
      for (int i = 0; i < m; ++i) y[i] = S.y[i];
    if (rc)
      for (int j = 0; j < n; ++j) rc[j] = S.pos[j] >= 0 ? 0.0 : S.d[j];
  }
  return 0;
}

}  // namespace

extern "C" int cuoptamd_dual_simplex(const cuoptamd_lp* lp, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
                                     int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc)
{
  try {
    return solve(lp, nullptr, nullptr, time_limit, iteration_limit, cancel, status, iterations, objective, x, y, rc);
  } catch (const Cancelled&) {
    *status = 9;
    return 0;
  } catch (const std::bad_alloc&) {
    *status = 8;  // no memory for the factorisation: too large for this engine
    return 0;
  }
}

extern "C" int cuoptamd_dual_simplex_from(const cuoptamd_lp* lp, const double* x0, const double* y0, double time_limit, int32_t iteration_limit,
                                          const volatile int32_t* cancel, int32_t* status, int32_t* iterations, double* objective,
                                          double* x, double* y, double* rc)
{
  if (!x0) return -1;
  try {
    return solve(lp, x0, y0, time_limit, iteration_limit, cancel, status, iterations, objective, x, y, rc);
  } catch (const Cancelled&) {
    *status = 9;
    return 0;
  } catch (const std::bad_alloc&) {
    *status = 8;
    return 0;
  }
}
