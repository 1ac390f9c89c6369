// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// C-ABI glue over the *reference's own* host-only sources, compiled in place from
// /root/reference (see oracle/Makefile target `_ref`): libmps_parser and the CPU dual simplex.
// It is used (a) to validate our own MPS parser against the reference parser, (b) as an
// objective-value oracle taken from the reference itself,
// and (c) to produce the small golden fixtures under tests/golden/ (scripts/make_golden.py).
//
// The conversion from (lo,hi) rows to the dual simplex's user_problem_t follows the recipe of
// cpp/src/linear_programming/translate.hpp:29-110 and, for maximisation, the negation done in
// cpp/src/mip/problem/problem_helpers.cuh:126-141.  This file contains no reference code.
#include <dual_simplex/solve.hpp>
#include <dual_simplex/sparse_matrix.hpp>
#include <dual_simplex/user_problem.hpp>
#include <mps_parser/parser.hpp>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace ds = cuopt::linear_programming::dual_simplex;
using model_t = cuopt::mps_parser::mps_data_model_t<int, double>;

struct ref_model {
  model_t m;
  std::string err;
  // constraint bounds, always materialised as (lo, hi)
  std::vector<double> lo, hi, lb, ub;
};

static void materialise_bounds(ref_model* r)
{
  const model_t& d = r->m;
  const int m      = d.get_n_constraints();
  const int n      = d.get_n_variables();
  const double inf = std::numeric_limits<double>::infinity();
  if (!d.get_constraint_lower_bounds().empty() || !d.get_constraint_upper_bounds().empty()) {
    r->lo = d.get_constraint_lower_bounds();
    r->hi = d.get_constraint_upper_bounds();
  } else {
    r->lo.assign(m, 0.0);
    r->hi.assign(m, 0.0);
    const auto& t = d.get_row_types();
    const auto& b = d.get_constraint_bounds();
    for (int i = 0; i < m; ++i) {
      const double bi = i < (int)b.size() ? b[i] : 0.0;
      const char ti   = i < (int)t.size() ? t[i] : 'E';
      if (ti == 'E') {
        r->lo[i] = bi;
        r->hi[i] = bi;
      } else if (ti == 'G') {
        r->lo[i] = bi;
        r->hi[i] = inf;
      } else {
        r->lo[i] = -inf;
        r->hi[i] = bi;
      }
    }
  }
  r->lb = d.get_variable_lower_bounds();
  r->ub = d.get_variable_upper_bounds();
  if (r->lb.empty()) r->lb.assign(n, 0.0);
  if (r->ub.empty()) r->ub.assign(n, inf);
}

extern "C" {

// returns 0 ok, 2 file error, 3 parse error (same split as cuopt_c.cpp:71-79)
int ref_mps_parse(const char* path, int fixed_format, void** out, char* errbuf, int errlen)
{
  *out = nullptr;
  try {
    auto* r = new ref_model{cuopt::mps_parser::parse_mps<int, double>(std::string(path),
                                                                       fixed_format != 0),
                            {}, {}, {}, {}, {}};
    materialise_bounds(r);
    *out = r;
    return 0;
  } catch (const std::exception& e) {
    if (errbuf && errlen > 0) std::snprintf(errbuf, errlen, "%s", e.what());
    return std::string(e.what()).find("Error opening MPS file") != std::string::npos ? 2 : 3;
  }
}

void ref_mps_free(void* h) { delete static_cast<ref_model*>(h); }

void ref_mps_dims(void* h, int* m, int* n, int* nnz, int* maximize, double* offset,
                  double* scaling)
{
  auto* r   = static_cast<ref_model*>(h);
  *m        = r->m.get_n_constraints();
  *n        = r->m.get_n_variables();
  *nnz      = r->m.get_nnz();
  *maximize = r->m.get_sense() ? 1 : 0;
  *offset   = r->m.get_objective_offset();
  *scaling  = r->m.get_objective_scaling_factor();
}

// flags: bit0 = model carried explicit (lo,hi) rows, bit1 = carried row types + rhs
int ref_mps_flags(void* h)
{
  auto* r = static_cast<ref_model*>(h);
  int f   = 0;
  if (!r->m.get_constraint_lower_bounds().empty()) f |= 1;
  if (!r->m.get_row_types().empty()) f |= 2;
  return f;
}

void ref_mps_arrays(void* h, int* offsets, int* indices, double* values, double* c, double* lo,
                    double* hi, double* lb, double* ub, char* var_types, char* row_types,
                    double* rhs)
{
  auto* r  = static_cast<ref_model*>(h);
  auto& d  = r->m;
  auto cpy = [](auto* dst, const auto& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cpy(offsets, d.get_constraint_matrix_offsets());
  cpy(indices, d.get_constraint_matrix_indices());
  cpy(values, d.get_constraint_matrix_values());
  cpy(c, d.get_objective_coefficients());
  cpy(lo, r->lo);
  cpy(hi, r->hi);
  cpy(lb, r->lb);
  cpy(ub, r->ub);
  if (var_types) {
    const auto& vt = d.get_variable_types();
    for (int j = 0; j < d.get_n_variables(); ++j) var_types[j] = j < (int)vt.size() ? vt[j] : 'C';
  }
  cpy(row_types, d.get_row_types());
  cpy(rhs, d.get_constraint_bounds());
}

int ref_mps_name(void* h, int kind, int idx, char* buf, int len)
{
  auto* r = static_cast<ref_model*>(h);
  std::string s;
  if (kind == 0) s = r->m.get_problem_name();
  if (kind == 1) s = r->m.get_objective_name();
  if (kind == 2 && idx < (int)r->m.get_variable_names().size()) s = r->m.get_variable_names()[idx];
  if (kind == 3 && idx < (int)r->m.get_row_names().size()) s = r->m.get_row_names()[idx];
  std::snprintf(buf, len, "%s", s.c_str());
  return (int)s.size();
}

// Reference CPU dual simplex on  min/max  s*(c.x)+off,  lo <= A x <= hi,  lb <= x <= ub.
// `c` is the user's objective; for maximize we negate it and pass obj_scale=-1 like the
// reference's problem_t does.  Returns the dual simplex lp_status_t as int.
int ref_dual_simplex(int m, int n, const int* offsets, const int* indices, const double* values,
                     const double* c, const double* lo, const double* hi, const double* lb,
                     const double* ub, int maximize, double obj_offset, double time_limit,
                     double* objective, double* x, double* y, double* z, int* iterations)
{
  const double inf = std::numeric_limits<double>::infinity();
  ds::user_problem_t<int, double> up;
  up.num_rows = m;
  up.num_cols = n;
  up.objective.assign(c, c + n);
  if (maximize)
    for (auto& v : up.objective) v = -v;
  ds::csr_matrix_t<int, double> csr;
  csr.m      = m;
  csr.n      = n;
  csr.nz_max = offsets[m];
  csr.x.assign(values, values + offsets[m]);
  csr.j.assign(indices, indices + offsets[m]);
  csr.row_start.assign(offsets, offsets + m + 1);
  csr.to_compressed_col(up.A);
  up.rhs.resize(m);
  up.row_sense.resize(m);
  for (int i = 0; i < m; ++i) {
    if (lo[i] == hi[i]) {
      up.row_sense[i] = 'E';
      up.rhs[i]       = lo[i];
    } else if (hi[i] == inf) {
      up.row_sense[i] = 'G';
      up.rhs[i]       = lo[i];
    } else if (lo[i] == -inf) {
      up.row_sense[i] = 'L';
      up.rhs[i]       = hi[i];
    } else {
      up.row_sense[i] = 'E';
      up.rhs[i]       = lo[i];
      up.range_rows.push_back(i);
      up.range_value.push_back(hi[i] - lo[i]);
    }
  }
  up.num_range_rows = (int)up.range_rows.size();
  up.lower.assign(lb, lb + n);
  up.upper.assign(ub, ub + n);
  up.obj_constant = obj_offset;
  up.obj_scale    = maximize ? -1.0 : 1.0;
  up.var_types.assign(n, ds::variable_type_t::CONTINUOUS);

  ds::simplex_solver_settings_t<int, double> settings;
  settings.set_log(false);
  settings.time_limit = time_limit > 0 ? time_limit : inf;
  ds::lp_solution_t<int, double> sol(m, n);
  ds::lp_status_t st = ds::solve_linear_program<int, double>(up, settings, sol);
  *objective         = sol.user_objective;
  *iterations        = sol.iterations;
  if (x) std::memcpy(x, sol.x.data(), sizeof(double) * std::min<size_t>(n, sol.x.size()));
  if (y) std::memcpy(y, sol.y.data(), sizeof(double) * std::min<size_t>(m, sol.y.size()));
  if (z) std::memcpy(z, sol.z.data(), sizeof(double) * std::min<size_t>(n, sol.z.size()));
  return (int)st;
}

}  // extern "C"
