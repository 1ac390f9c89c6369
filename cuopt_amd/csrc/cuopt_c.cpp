// libcuopt C API (include/cuopt/linear_programming/cuopt_c.h) on top of the MI355X-native PDLP
// host driver.  Behaviour mirrors cuOpt 25.08 cpp/src/linear_programming/cuopt_c.cpp (handles,
// null checks, return codes) and cpp/src/math_optimization/solver_settings.cu:66-330 (parameter
// registry: names, types, ranges, defaults, string conversions).  No HIP header is included here.
#include <cuopt/linear_programming/cuopt_c.h>

#include <algorithm>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <atomic>
#include <chrono>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cuopt_amd/pdlp_solver.h"
#include "host_parallel.hpp"
#include "mps_reader.hpp"

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

// ---- problem ------------------------------------------------------------------------------------
// optimization_problem_t (host-resident here; the reference keeps device copies and synchronises
// in every getter, cuopt_c.cpp:208-438 -- the getters below have the same observable behaviour)
struct Problem {
  int32_t m = 0, n = 0;
  bool maximize = false;
  double objective_offset = 0.0;
  std::vector<int32_t> offsets, indices;
  std::vector<double> values, c, lb, ub;
  std::vector<char> row_types;   // empty when built from ranged bounds
  std::vector<double> rhs;       // idem
  std::vector<double> lo, hi;    // always materialised (problem_helpers.cuh:33-58)
  std::vector<char> var_types;
  std::vector<std::string> var_names;  // from the MPS file (empty for array-built problems)
  std::vector<std::string> row_names;  // idem
  std::string problem_name, objective_name;
  bool has_integers() const
  {
    for (char t : var_types)
      if (t == CUOPT_INTEGER) return true;
    return false;
  }
};

// ---- settings -----------------------------------------------------------------------------------
struct Settings {
  // pdlp_solver_settings_t
  double tol[6] = {1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4};  // abs/rel dual, abs/rel primal, abs/rel gap
  double primal_infeasible_tolerance = 1e-8, dual_infeasible_tolerance = 1e-8;
  double time_limit = kInf;
  int32_t iteration_limit = INT_MAX, pdlp_solver_mode = CUOPT_PDLP_SOLVER_MODE_STABLE2,
          method = CUOPT_METHOD_CONCURRENT, num_cpu_threads = -1;
  // extensions of this library (not in the reference's registry): row blocks / GPUs of one solve (0 = the
  // CUOPT_AMD_NUM_GPUS environment variable, default 1) and the simplex-grade emulation switch (-1 = the key simplex_grade of
  // the CUOPT_AMD_TUNE environment string, default on)
  int32_t num_gpus = 0, simplex_grade = -1;
  int32_t dual_simplex = -1;  // the dual simplex engine: -1 = CUOPT_AMD_DUAL_SIMPLEX (default on), 0 off, 1 on
  bool infeasibility_detection = false, strict_infeasibility = false, per_constraint_residual = false,
       save_best_primal_so_far = false, first_primal_feasible = false, log_to_console = true,
       crossover = false, mip_scaling = true, mip_heuristics_only = false;
  // MIP tolerances are registered so that clients that set them keep working
  double mip_abs_tol = 1e-4, mip_rel_tol = 1e-4, mip_int_tol = 1e-5, mip_abs_gap = 1e-10, mip_rel_gap = 1e-4;
  std::string log_file, solution_file, user_problem_file;

  struct FloatParam { const char* name; double* value; double lo, hi; };
  struct IntParam { const char* name; int32_t* value; int32_t lo, hi; };
  struct BoolParam { const char* name; bool* value; };
  struct StringParam { const char* name; std::string* value; };
  std::vector<FloatParam> floats;
  std::vector<IntParam> ints;
  std::vector<BoolParam> bools;
  std::vector<StringParam> strings;

  Settings()
  {
    floats = {{CUOPT_TIME_LIMIT, &time_limit, 0.0, kInf},
              {CUOPT_ABSOLUTE_DUAL_TOLERANCE, &tol[0], 0.0, 1e-1},
              {CUOPT_RELATIVE_DUAL_TOLERANCE, &tol[1], 0.0, 1e-1},
              {CUOPT_ABSOLUTE_PRIMAL_TOLERANCE, &tol[2], 0.0, 1e-1},
              {CUOPT_RELATIVE_PRIMAL_TOLERANCE, &tol[3], 0.0, 1e-1},
              {CUOPT_ABSOLUTE_GAP_TOLERANCE, &tol[4], 0.0, 1e-1},
              {CUOPT_RELATIVE_GAP_TOLERANCE, &tol[5], 0.0, 1e-1},
              {CUOPT_MIP_ABSOLUTE_TOLERANCE, &mip_abs_tol, 0.0, 1e-1},
              {CUOPT_MIP_RELATIVE_TOLERANCE, &mip_rel_tol, 0.0, 1e-1},
              {CUOPT_MIP_INTEGRALITY_TOLERANCE, &mip_int_tol, 0.0, 1e-1},
              {CUOPT_MIP_ABSOLUTE_GAP, &mip_abs_gap, 0.0, 1e-1},
              {CUOPT_MIP_RELATIVE_GAP, &mip_rel_gap, 0.0, 1e-1},
              {CUOPT_PRIMAL_INFEASIBLE_TOLERANCE, &primal_infeasible_tolerance, 0.0, 1e-1},
              {CUOPT_DUAL_INFEASIBLE_TOLERANCE, &dual_infeasible_tolerance, 0.0, 1e-1}};
    ints = {{CUOPT_ITERATION_LIMIT, &iteration_limit, 0, INT_MAX},
            {CUOPT_PDLP_SOLVER_MODE, &pdlp_solver_mode, CUOPT_PDLP_SOLVER_MODE_STABLE1, CUOPT_PDLP_SOLVER_MODE_FAST1},
            {CUOPT_METHOD, &method, CUOPT_METHOD_CONCURRENT, CUOPT_METHOD_DUAL_SIMPLEX},
            {CUOPT_NUM_CPU_THREADS, &num_cpu_threads, -1, INT_MAX},
            {"amd_num_gpus", &num_gpus, 0, 16},
            {"amd_simplex_grade", &simplex_grade, -1, 1},
            {"amd_dual_simplex", &dual_simplex, -1, 1}};
    bools = {{CUOPT_INFEASIBILITY_DETECTION, &infeasibility_detection},
             {CUOPT_STRICT_INFEASIBILITY, &strict_infeasibility},
             {CUOPT_PER_CONSTRAINT_RESIDUAL, &per_constraint_residual},
             {CUOPT_SAVE_BEST_PRIMAL_SO_FAR, &save_best_primal_so_far},
             {CUOPT_FIRST_PRIMAL_FEASIBLE, &first_primal_feasible},
             {CUOPT_MIP_SCALING, &mip_scaling},
             {CUOPT_MIP_HEURISTICS_ONLY, &mip_heuristics_only},
             {CUOPT_LOG_TO_CONSOLE, &log_to_console},
             {CUOPT_CROSSOVER, &crossover}};
    strings = {{CUOPT_LOG_FILE, &log_file}, {CUOPT_SOLUTION_FILE, &solution_file},
               {CUOPT_USER_PROBLEM_FILE, &user_problem_file}};
  }
  Settings(const Settings&)            = delete;
  Settings& operator=(const Settings&) = delete;

  // each setter returns false for "no such parameter of this type"; throws on a bad value
  bool set_int(const std::string& name, int32_t v)
  {
    for (auto& p : ints)
      if (name == p.name) {
        if (v < p.lo || v > p.hi) throw std::out_of_range(name);
        *p.value = v;
        return true;
      }
    return false;
  }
  bool set_float(const std::string& name, double v)
  {
    for (auto& p : floats)
      if (name == p.name) {
        if (!(v >= p.lo && v <= p.hi)) throw std::out_of_range(name);
        *p.value = v;
        return true;
      }
    return false;
  }
  bool set_bool(const std::string& name, bool v)
  {
    for (auto& p : bools)
      if (name == p.name) {
        *p.value = v;
        return true;
      }
    return false;
  }
  bool set_string(const std::string& name, const std::string& v)
  {
    for (auto& p : strings)
      if (name == p.name) {
        *p.value = v;
        return true;
      }
    return false;
  }
  // set_parameter_from_string, solver_settings.cu:121-189
  void set_from_string(const std::string& name, const std::string& value)
  {
    bool found = false;
    for (auto& p : ints)
      if (name == p.name) {
        size_t used = 0;
        long long v = std::stoll(value, &used);  // throws invalid_argument like the reference
        if (v < p.lo || v > p.hi) throw std::out_of_range(name);
        *p.value = (int32_t)v;
        found    = true;
      }
    for (auto& p : floats)
      if (name == p.name) {
        double v = std::stod(value);
        if (!(v >= p.lo && v <= p.hi)) throw std::out_of_range(name);
        *p.value = v;
        found    = true;
      }
    for (auto& p : bools)
      if (name == p.name) {
        // string_to_bool, solver_settings.cu:47-60
        if (value == "true" || value == "True" || value == "TRUE" || value == "1" || value == "t" || value == "T")
          *p.value = true;
        else if (value == "false" || value == "False" || value == "FALSE" || value == "0" || value == "f" || value == "F")
          *p.value = false;
        else
          throw std::invalid_argument(name);
        found = true;
      }
    for (auto& p : strings)
      if (name == p.name) {
        *p.value = value;
        found    = true;
      }
    if (!found) throw std::invalid_argument("Parameter " + name + " not found");
  }
  // get_parameter_as_string, solver_settings.cu:276-292 (std::to_string formats)
  std::string get_as_string(const std::string& name) const
  {
    for (auto& p : ints)
      if (name == p.name) return std::to_string(*p.value);
    for (auto& p : floats)
      if (name == p.name) return std::to_string(*p.value);
    for (auto& p : bools)
      if (name == p.name) return *p.value ? "true" : "false";
    for (auto& p : strings)
      if (name == p.name) return *p.value;
    throw std::invalid_argument("Parameter " + name + " not found");
  }
};

// ---- solution -----------------------------------------------------------------------------------
struct Solution {
  bool is_mip = false;
  int32_t termination_status = CUOPT_TERIMINATION_STATUS_NO_TERMINATION;
  int32_t error_status       = CUOPT_SUCCESS;
  std::string error_message;
  std::vector<double> x, y, rc;
  double objective = 0.0, solve_time = 0.0;
  cuoptamd_result stats{};
  std::string solve_info;  // cuOptAmdGetSolveInfo: which engine / attempt answered
};

// cuopt::logic_error message format, cpp/include/cuopt/error.hpp:110-125
std::string error_json(const char* type, const std::string& msg)
{
  return std::string("{\"CUOPT_ERROR_TYPE\": \"") + type + "\", \"msg\": \"" + msg + "\"}";
}

// problem_checking_t::check_problem_representation (LP/utilities/problem_checking.cu:100-250)
// CUOPT_USER_PROBLEM_FILE (solve.cu:586-589 -> problem_t::write_as_mps, mip/problem/write_mps.cu): the problem as the
// solver sees it, free-format MPS, values with 17 significant digits.  Own writer; two deliberate differences from
// the reference's: a two-sided row is written as a 'G' row with RHS = lower bound and RANGES = upper - lower (the
// reference writes such rows as 'L' with the lower bound as RHS, which MPS readers -- its own included -- read back as
// [lower - range, lower]), and the objective offset is kept (negated RHS entry of the objective row).
bool write_problem_as_mps(const Problem& p, const std::string& path)
{
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  auto row = [&](int32_t i) { return (size_t)i < p.row_names.size() ? p.row_names[i] : "R" + std::to_string(i); };
  auto col = [&](int32_t j) { return (size_t)j < p.var_names.size() ? p.var_names[j] : "C" + std::to_string(j); };
  const std::string obj = p.objective_name.empty() ? "OBJ" : p.objective_name;
  std::fprintf(f, "NAME          %s\n", p.problem_name.c_str());
  if (p.maximize) std::fprintf(f, "OBJSENSE\n    MAX\n");
  std::fprintf(f, "ROWS\n N  %s\n", obj.c_str());
  // a two-sided row is a 'G' row [rhs, rhs + range] or an 'L' row [rhs - range, rhs]: whichever reproduces the other
  // bound exactly in floating point (lo + (hi - lo) may be one ulp off hi)
  auto ranged_as_L = [&](int32_t i) {
    const double lo = p.lo[i], hi = p.hi[i], r = hi - lo;
    return std::isfinite(lo) && std::isfinite(hi) && lo != hi && lo + r != hi && hi - r == lo;
  };
  for (int32_t i = 0; i < p.m; ++i) {
    const double lo = p.lo[i], hi = p.hi[i];
    // (a free row, -inf..+inf, has no MPS spelling other than 'N'; every reader -- this library's and the reference's --
    // drops N rows after the first, so such rows disappear on a round trip: the LP is unchanged, its row count is not.
    // The reference's writer turns them into 'G' rows with right-hand side 0, which changes the LP.)
    char type = lo == hi ? 'E' : (std::isinf(lo) && lo < 0 ? (std::isinf(hi) ? 'N' : 'L') : 'G');
    if (ranged_as_L(i)) type = 'L';
    std::fprintf(f, " %c  %s\n", type, row(i).c_str());
  }
  // COLUMNS wants the matrix by column: counting sort of the CSR entries
  std::vector<int64_t> start((size_t)p.n + 1, 0);
  for (int32_t j : p.indices) start[(size_t)j + 1] += 1;
  for (int32_t j = 0; j < p.n; ++j) start[(size_t)j + 1] += start[j];
  std::vector<int32_t> rows_of(p.indices.size());
  std::vector<double> vals_of(p.indices.size());
  {
    std::vector<int64_t> cursor(start.begin(), start.end() - 1);
    for (int32_t i = 0; i < p.m; ++i)
      for (int32_t k = p.offsets[i]; k < p.offsets[i + 1]; ++k) {
        const int64_t q = cursor[p.indices[k]]++;
        rows_of[q] = i, vals_of[q] = p.values[k];
      }
  }
  std::fprintf(f, "COLUMNS\n");
  bool in_int = false;
  for (int32_t j = 0; j < p.n; ++j) {
    const bool integer = p.var_types[j] == CUOPT_INTEGER;
    if (integer && !in_int) std::fprintf(f, "    MARK0001  'MARKER'                 'INTORG'\n"), in_int = true;
    if (!integer && in_int) std::fprintf(f, "    MARK0001  'MARKER'                 'INTEND'\n"), in_int = false;
    bool any = false;
    if (p.c[j] != 0.0) std::fprintf(f, "    %s %s %.17g\n", col(j).c_str(), obj.c_str(), p.c[j]), any = true;
    for (int64_t q = start[j]; q < start[(size_t)j + 1]; ++q)
      std::fprintf(f, "    %s %s %.17g\n", col(j).c_str(), row(rows_of[q]).c_str(), vals_of[q]), any = true;
    if (!any) std::fprintf(f, "    %s %s 0\n", col(j).c_str(), obj.c_str());  // a column must appear to exist
  }
  if (in_int) std::fprintf(f, "    MARK0001  'MARKER'                 'INTEND'\n");
  std::fprintf(f, "RHS\n");
  if (p.objective_offset != 0.0) std::fprintf(f, "    RHS1      %s %.17g\n", obj.c_str(), -p.objective_offset);
  for (int32_t i = 0; i < p.m; ++i) {
    const double rhs = (std::isinf(p.lo[i]) || ranged_as_L(i)) ? p.hi[i] : p.lo[i];
    if (std::isfinite(rhs) && rhs != 0.0) std::fprintf(f, "    RHS1      %s %.17g\n", row(i).c_str(), rhs);
  }
  bool ranges = false;
  for (int32_t i = 0; i < p.m; ++i)
    if (std::isfinite(p.lo[i]) && std::isfinite(p.hi[i]) && p.lo[i] != p.hi[i]) {
      if (!ranges) std::fprintf(f, "RANGES\n"), ranges = true;
      std::fprintf(f, "    RNG1      %s %.17g\n", row(i).c_str(), p.hi[i] - p.lo[i]);
    }
  std::fprintf(f, "BOUNDS\n");
  for (int32_t j = 0; j < p.n; ++j) {
    const double lb = p.lb[j], ub = p.ub[j];
    const std::string name = col(j);
    if (std::isinf(lb) && lb < 0 && std::isinf(ub) && ub > 0) {
      std::fprintf(f, " FR BOUND1    %s\n", name.c_str());
    } else if (lb == ub) {
      std::fprintf(f, " FX BOUND1    %s %.17g\n", name.c_str(), lb);
    } else {
      if (std::isinf(lb) && lb < 0) std::fprintf(f, " MI BOUND1    %s\n", name.c_str());
      else if (lb != 0.0) std::fprintf(f, " LO BOUND1    %s %.17g\n", name.c_str(), lb);
      if (std::isfinite(ub)) std::fprintf(f, " UP BOUND1    %s %.17g\n", name.c_str(), ub);
      else if (p.var_types[j] == CUOPT_INTEGER && lb == 0.0) std::fprintf(f, " PL BOUND1    %s\n", name.c_str());
    }
  }
  std::fprintf(f, "ENDATA\n");
  return std::fclose(f) == 0;
}

std::string validate(const Problem& p)
{
  if (p.offsets.empty()) return "A_offsets must be set before calling the solver.";
  if (p.offsets[0] != 0) return "A_offsets first value should be 0.";
  for (size_t i = 1; i < p.offsets.size(); ++i)
    if (p.offsets[i] < p.offsets[i - 1]) return "A_offsets values must in an increasing order.";
  if (p.indices.size() != p.values.size()) return "A_index and A_values must have same sizes.";
  for (int32_t j : p.indices)
    if (j < 0 || j >= p.n) return "A_indices values must positive lower than the number of variables (c size).";
  for (char t : p.row_types)
    if (t != 'E' && t != 'G' && t != 'L') return "row_types values must equal to 'E', 'G' or 'L'.";
  for (int32_t j = 0; j < p.n; ++j)
    if (std::isnan(p.lb[j]) || std::isnan(p.ub[j])) return "Variable bounds must not be NaN.";
  return "";
}

// Inputs may be host or device arrays (reference: cuopt_c.cpp:119, raft::copy at :261-403); the problem object is
// host-resident here, so device arrays are copied down once (pdlpdev_copy_in asks the HIP runtime what the pointer is).
template <class T>
void fetch(std::vector<T>& dst, const T* src, size_t count)
{
  dst.resize(count);
  if (count && pdlpdev_copy_in(dst.data(), src, count * sizeof(T)) != 0) throw std::runtime_error(pdlpdev_last_error());
}

int create_common(Problem* p, cuopt_int_t m, cuopt_int_t n, cuopt_int_t sense, double offset,
                  const double* c, const cuopt_int_t* off, const cuopt_int_t* idx, const double* val,
                  const double* lb, const double* ub, const char* types)
{
  if (m < 0 || n < 0) return CUOPT_INVALID_ARGUMENT;
  p->m = m, p->n = n;
  p->maximize         = sense == CUOPT_MAXIMIZE;
  p->objective_offset = offset;
  fetch(p->c, c, (size_t)n);
  fetch(p->offsets, off, (size_t)m + 1);
  const int64_t nnz = p->offsets[m];
  if (nnz < 0) return CUOPT_INVALID_ARGUMENT;
  fetch(p->indices, idx, (size_t)nnz);
  fetch(p->values, val, (size_t)nnz);
  fetch(p->lb, lb, (size_t)n);
  fetch(p->ub, ub, (size_t)n);
  std::vector<char> t;
  fetch(t, types, (size_t)n);
  p->var_types.resize(n);
  for (int32_t j = 0; j < n; ++j) p->var_types[j] = t[j] == CUOPT_CONTINUOUS ? CUOPT_CONTINUOUS : CUOPT_INTEGER;
  return CUOPT_SUCCESS;
}

}  // namespace

extern "C" {

int8_t cuOptGetFloatSize() { return (int8_t)sizeof(cuopt_float_t); }
int8_t cuOptGetIntSize() { return (int8_t)sizeof(cuopt_int_t); }

cuopt_int_t cuOptReadProblem(const char* filename, cuOptOptimizationProblem* problem_ptr)
{
  if (filename == nullptr || problem_ptr == nullptr) return CUOPT_INVALID_ARGUMENT;
  *problem_ptr = nullptr;
  try {
    cuopt_amd::MpsModel mdl = cuopt_amd::read_mps_file(filename);
    auto p                  = std::make_unique<Problem>();
    p->m = (int32_t)mdl.row_names.size(), p->n = (int32_t)mdl.var_names.size();
    p->maximize         = mdl.maximize;
    p->objective_offset = mdl.objective_offset;
    p->offsets.assign(mdl.offsets.begin(), mdl.offsets.end());
    p->indices.assign(mdl.indices.begin(), mdl.indices.end());
    p->values = std::move(mdl.values), p->c = std::move(mdl.c);
    p->lb = std::move(mdl.lb), p->ub = std::move(mdl.ub);
    p->row_types = std::move(mdl.row_types), p->rhs = std::move(mdl.rhs);
    p->lo = std::move(mdl.lo), p->hi = std::move(mdl.hi);
    p->var_types = std::move(mdl.var_types);
    p->var_names = std::move(mdl.var_names);
    p->row_names = std::move(mdl.row_names);
    p->problem_name = mdl.problem_name, p->objective_name = mdl.objective_name;
    for (char& t : p->var_types) t = t == 'I' ? CUOPT_INTEGER : CUOPT_CONTINUOUS;
    *problem_ptr = p.release();
  } catch (const cuopt_amd::MpsError& e) {
    return e.cannot_open ? CUOPT_MPS_FILE_ERROR : CUOPT_MPS_PARSE_ERROR;  // cuopt_c.cpp:71-79
  } catch (const std::exception&) {
    return CUOPT_MPS_PARSE_ERROR;
  }
  return CUOPT_SUCCESS;
}

cuopt_int_t cuOptCreateProblem(cuopt_int_t num_constraints, cuopt_int_t num_variables,
                               cuopt_int_t objective_sense, cuopt_float_t objective_offset,
                               const cuopt_float_t* objective_coefficients,
                               const cuopt_int_t* constraint_matrix_row_offsets,
                               const cuopt_int_t* constraint_matrix_column_indices,
                               const cuopt_float_t* constraint_matrix_coefficent_values,
                               const char* constraint_sense, const cuopt_float_t* rhs,
                               const cuopt_float_t* lower_bounds, const cuopt_float_t* upper_bounds,
                               const char* variable_types, cuOptOptimizationProblem* problem_ptr)
{
  if (problem_ptr == nullptr || objective_coefficients == nullptr ||
      constraint_matrix_row_offsets == nullptr || constraint_matrix_column_indices == nullptr ||
      constraint_matrix_coefficent_values == nullptr || constraint_sense == nullptr || rhs == nullptr ||
      lower_bounds == nullptr || upper_bounds == nullptr || variable_types == nullptr)
    return CUOPT_INVALID_ARGUMENT;
  try {
    auto p = std::make_unique<Problem>();
    int rc = create_common(p.get(), num_constraints, num_variables, objective_sense, objective_offset,
                           objective_coefficients, constraint_matrix_row_offsets,
                           constraint_matrix_column_indices, constraint_matrix_coefficent_values,
                           lower_bounds, upper_bounds, variable_types);
    if (rc != CUOPT_SUCCESS) return rc;
    fetch(p->row_types, constraint_sense, (size_t)num_constraints);
    fetch(p->rhs, rhs, (size_t)num_constraints);
    p->lo.resize(num_constraints), p->hi.resize(num_constraints);
    for (int32_t i = 0; i < num_constraints; ++i) {  // set_constraint_bounds_if_not_set, problem_helpers.cuh:33-58
      const char t = p->row_types[i];
      p->lo[i]     = (t == 'E' || t == 'G') ? p->rhs[i] : -kInf;
      p->hi[i]     = (t == 'E' || t == 'L') ? p->rhs[i] : kInf;
    }
    *problem_ptr = p.release();
  } catch (const std::exception&) {
    return CUOPT_INVALID_ARGUMENT;
  }
  return CUOPT_SUCCESS;
}

cuopt_int_t cuOptCreateRangedProblem(cuopt_int_t num_constraints, cuopt_int_t num_variables,
                                     cuopt_int_t objective_sense, cuopt_float_t objective_offset,
                                     const cuopt_float_t* objective_coefficients,
                                     const cuopt_int_t* constraint_matrix_row_offsets,
                                     const cuopt_int_t* constraint_matrix_column_indices,
                                     const cuopt_float_t* constraint_matrix_coefficients,
                                     const cuopt_float_t* constraint_lower_bounds,
                                     const cuopt_float_t* constraint_upper_bounds,
                                     const cuopt_float_t* variable_lower_bounds,
                                     const cuopt_float_t* variable_upper_bounds,
                                     const char* variable_types, cuOptOptimizationProblem* problem_ptr)
{
  if (problem_ptr == nullptr || objective_coefficients == nullptr ||
      constraint_matrix_row_offsets == nullptr || constraint_matrix_column_indices == nullptr ||
      constraint_matrix_coefficients == nullptr || constraint_lower_bounds == nullptr ||
      constraint_upper_bounds == nullptr || variable_lower_bounds == nullptr ||
      variable_upper_bounds == nullptr || variable_types == nullptr)
    return CUOPT_INVALID_ARGUMENT;
  try {
    auto p = std::make_unique<Problem>();
    int rc = create_common(p.get(), num_constraints, num_variables, objective_sense, objective_offset,
                           objective_coefficients, constraint_matrix_row_offsets,
                           constraint_matrix_column_indices, constraint_matrix_coefficients,
                           variable_lower_bounds, variable_upper_bounds, variable_types);
    if (rc != CUOPT_SUCCESS) return rc;
    fetch(p->lo, constraint_lower_bounds, (size_t)num_constraints);
    fetch(p->hi, constraint_upper_bounds, (size_t)num_constraints);
    *problem_ptr = p.release();
  } catch (const std::exception&) {
    return CUOPT_INVALID_ARGUMENT;
  }
  return CUOPT_SUCCESS;
}

void cuOptDestroyProblem(cuOptOptimizationProblem* problem_ptr)
{
  if (problem_ptr == nullptr || *problem_ptr == nullptr) return;
  delete static_cast<Problem*>(*problem_ptr);
  *problem_ptr = nullptr;
}

#define PROBLEM_GETTER_PROLOGUE(out)                                         \
  if (problem == nullptr || (out) == nullptr) return CUOPT_INVALID_ARGUMENT; \
  const Problem* p = static_cast<const Problem*>(problem)

cuopt_int_t cuOptGetNumConstraints(cuOptOptimizationProblem problem, cuopt_int_t* num_constraints_ptr)
{
  PROBLEM_GETTER_PROLOGUE(num_constraints_ptr);
  *num_constraints_ptr = p->m;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetNumVariables(cuOptOptimizationProblem problem, cuopt_int_t* num_variables_ptr)
{
  PROBLEM_GETTER_PROLOGUE(num_variables_ptr);
  *num_variables_ptr = p->n;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetObjectiveSense(cuOptOptimizationProblem problem, cuopt_int_t* objective_sense_ptr)
{
  PROBLEM_GETTER_PROLOGUE(objective_sense_ptr);
  *objective_sense_ptr = p->maximize ? CUOPT_MAXIMIZE : CUOPT_MINIMIZE;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetObjectiveOffset(cuOptOptimizationProblem problem, cuopt_float_t* objective_offset_ptr)
{
  PROBLEM_GETTER_PROLOGUE(objective_offset_ptr);
  *objective_offset_ptr = p->objective_offset;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetObjectiveCoefficients(cuOptOptimizationProblem problem, cuopt_float_t* objective_coefficients_ptr)
{
  PROBLEM_GETTER_PROLOGUE(objective_coefficients_ptr);
  std::copy(p->c.begin(), p->c.end(), objective_coefficients_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetNumNonZeros(cuOptOptimizationProblem problem, cuopt_int_t* num_non_zeros_ptr)
{
  PROBLEM_GETTER_PROLOGUE(num_non_zeros_ptr);
  *num_non_zeros_ptr = (cuopt_int_t)p->values.size();
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetConstraintMatrix(cuOptOptimizationProblem problem,
                                     cuopt_int_t* constraint_matrix_row_offsets_ptr,
                                     cuopt_int_t* constraint_matrix_column_indices_ptr,
                                     cuopt_float_t* constraint_matrix_coefficients_ptr)
{
  if (problem == nullptr || constraint_matrix_row_offsets_ptr == nullptr ||
      constraint_matrix_column_indices_ptr == nullptr || constraint_matrix_coefficients_ptr == nullptr)
    return CUOPT_INVALID_ARGUMENT;
  const Problem* p = static_cast<const Problem*>(problem);
  std::copy(p->offsets.begin(), p->offsets.end(), constraint_matrix_row_offsets_ptr);
  std::copy(p->indices.begin(), p->indices.end(), constraint_matrix_column_indices_ptr);
  std::copy(p->values.begin(), p->values.end(), constraint_matrix_coefficients_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetConstraintSense(cuOptOptimizationProblem problem, char* constraint_sense_ptr)
{
  PROBLEM_GETTER_PROLOGUE(constraint_sense_ptr);
  std::copy(p->row_types.begin(), p->row_types.end(), constraint_sense_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetConstraintRightHandSide(cuOptOptimizationProblem problem, cuopt_float_t* rhs_ptr)
{
  PROBLEM_GETTER_PROLOGUE(rhs_ptr);
  std::copy(p->rhs.begin(), p->rhs.end(), rhs_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetConstraintLowerBounds(cuOptOptimizationProblem problem, cuopt_float_t* lower_bounds_ptr)
{
  PROBLEM_GETTER_PROLOGUE(lower_bounds_ptr);
  std::copy(p->lo.begin(), p->lo.end(), lower_bounds_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetConstraintUpperBounds(cuOptOptimizationProblem problem, cuopt_float_t* upper_bounds_ptr)
{
  PROBLEM_GETTER_PROLOGUE(upper_bounds_ptr);
  std::copy(p->hi.begin(), p->hi.end(), upper_bounds_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetVariableLowerBounds(cuOptOptimizationProblem problem, cuopt_float_t* lower_bounds_ptr)
{
  PROBLEM_GETTER_PROLOGUE(lower_bounds_ptr);
  std::copy(p->lb.begin(), p->lb.end(), lower_bounds_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetVariableUpperBounds(cuOptOptimizationProblem problem, cuopt_float_t* upper_bounds_ptr)
{
  PROBLEM_GETTER_PROLOGUE(upper_bounds_ptr);
  std::copy(p->ub.begin(), p->ub.end(), upper_bounds_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetVariableTypes(cuOptOptimizationProblem problem, char* variable_types_ptr)
{
  PROBLEM_GETTER_PROLOGUE(variable_types_ptr);
  std::copy(p->var_types.begin(), p->var_types.end(), variable_types_ptr);
  return CUOPT_SUCCESS;
}

// ---- settings -------------------------------------------------------------------------------------
cuopt_int_t cuOptCreateSolverSettings(cuOptSolverSettings* settings_ptr)
{
  if (settings_ptr == nullptr) return CUOPT_INVALID_ARGUMENT;
  *settings_ptr = new (std::nothrow) Settings();
  return *settings_ptr ? CUOPT_SUCCESS : CUOPT_OUT_OF_MEMORY;
}
void cuOptDestroySolverSettings(cuOptSolverSettings* settings_ptr)
{
  if (settings_ptr == nullptr) return;
  delete static_cast<Settings*>(*settings_ptr);
  *settings_ptr = nullptr;
}
cuopt_int_t cuOptSetParameter(cuOptSolverSettings settings, const char* parameter_name, const char* parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr || parameter_value == nullptr) return CUOPT_INVALID_ARGUMENT;
  try {
    static_cast<Settings*>(settings)->set_from_string(parameter_name, parameter_value);
  } catch (const std::exception&) {
    return CUOPT_INVALID_ARGUMENT;
  }
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetParameter(cuOptSolverSettings settings, const char* parameter_name,
                              cuopt_int_t parameter_value_size, char* parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr || parameter_value == nullptr || parameter_value_size <= 0)
    return CUOPT_INVALID_ARGUMENT;
  try {
    std::string v = static_cast<Settings*>(settings)->get_as_string(parameter_name);
    std::snprintf(parameter_value, (size_t)parameter_value_size, "%s", v.c_str());
  } catch (const std::exception&) {
    return CUOPT_INVALID_ARGUMENT;
  }
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptSetIntegerParameter(cuOptSolverSettings settings, const char* parameter_name, cuopt_int_t parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr) return CUOPT_INVALID_ARGUMENT;
  Settings* s = static_cast<Settings*>(settings);
  try {
    if (s->set_int(parameter_name, parameter_value)) return CUOPT_SUCCESS;
    // maybe a boolean parameter (cuopt_c.cpp:493-505)
    if (s->set_bool(parameter_name, parameter_value != 0)) return CUOPT_SUCCESS;
  } catch (const std::exception&) {
  }
  return CUOPT_INVALID_ARGUMENT;
}
cuopt_int_t cuOptGetIntegerParameter(cuOptSolverSettings settings, const char* parameter_name, cuopt_int_t* parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr || parameter_value == nullptr) return CUOPT_INVALID_ARGUMENT;
  const Settings* s = static_cast<const Settings*>(settings);
  for (auto& p : s->ints)
    if (std::strcmp(p.name, parameter_name) == 0) {
      *parameter_value = *p.value;
      return CUOPT_SUCCESS;
    }
  for (auto& p : s->bools)
    if (std::strcmp(p.name, parameter_name) == 0) {
      *parameter_value = *p.value ? 1 : 0;
      return CUOPT_SUCCESS;
    }
  return CUOPT_INVALID_ARGUMENT;
}
cuopt_int_t cuOptSetFloatParameter(cuOptSolverSettings settings, const char* parameter_name, cuopt_float_t parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr) return CUOPT_INVALID_ARGUMENT;
  try {
    if (static_cast<Settings*>(settings)->set_float(parameter_name, parameter_value)) return CUOPT_SUCCESS;
  } catch (const std::exception&) {
  }
  return CUOPT_INVALID_ARGUMENT;
}
cuopt_int_t cuOptGetFloatParameter(cuOptSolverSettings settings, const char* parameter_name, cuopt_float_t* parameter_value)
{
  if (settings == nullptr || parameter_name == nullptr || parameter_value == nullptr) return CUOPT_INVALID_ARGUMENT;
  const Settings* s = static_cast<const Settings*>(settings);
  for (auto& p : s->floats)
    if (std::strcmp(p.name, parameter_name) == 0) {
      *parameter_value = *p.value;
      return CUOPT_SUCCESS;
    }
  return CUOPT_INVALID_ARGUMENT;
}

// ---- solve ----------------------------------------------------------------------------------------
cuopt_int_t cuOptIsMIP(cuOptOptimizationProblem problem, cuopt_int_t* is_mip_ptr)
{
  PROBLEM_GETTER_PROLOGUE(is_mip_ptr);
  *is_mip_ptr = p->has_integers() ? 1 : 0;
  return CUOPT_SUCCESS;
}

cuopt_int_t cuOptSolve(cuOptOptimizationProblem problem, cuOptSolverSettings settings, cuOptSolution* solution_ptr)
{
  if (problem == nullptr || settings == nullptr || solution_ptr == nullptr) return CUOPT_INVALID_ARGUMENT;
  const Problem* p  = static_cast<const Problem*>(problem);
  const Settings* s = static_cast<const Settings*>(settings);
  Solution* sol     = new (std::nothrow) Solution();
  if (!sol) return CUOPT_OUT_OF_MEMORY;
  *solution_ptr = sol;
  auto error    = [&](int code, const char* type, const std::string& msg) {
    sol->error_status  = code;
    sol->error_message = error_json(type, msg);
    return (cuopt_int_t)code;
  };
  try {
    if (p->has_integers()) {
      sol->is_mip = true;
      return error(CUOPT_VALIDATION_ERROR, "ValidationError",
                   "MILP is outside the scope of the MI355X-native PDLP library: only continuous LPs can be solved");
    }
    std::string bad = validate(*p);  // before anything walks the arrays (the MPS writer below trusts the indices)
    if (!bad.empty()) return error(CUOPT_VALIDATION_ERROR, "ValidationError", bad);
    if (!s->user_problem_file.empty() && !write_problem_as_mps(*p, s->user_problem_file))
      return error(CUOPT_RUNTIME_ERROR, "RuntimeError", "could not write the user problem file " + s->user_problem_file);

    cuoptamd_hyper hyper;
    cuoptamd_hyper_preset(s->pdlp_solver_mode, &hyper);
    cuoptamd_settings st;
    cuoptamd_default_settings(&st);
    st.absolute_dual_tolerance = s->tol[0], st.relative_dual_tolerance = s->tol[1];
    st.absolute_primal_tolerance = s->tol[2], st.relative_primal_tolerance = s->tol[3];
    st.absolute_gap_tolerance = s->tol[4], st.relative_gap_tolerance = s->tol[5];
    auto env_int = [](const char* name, int fallback) {
      const char* v = std::getenv(name);
      return v && *v ? std::atoi(v) : fallback;
    };
    const int gpus = std::max(1, s->num_gpus > 0 ? s->num_gpus : env_int("CUOPT_AMD_NUM_GPUS", 1));
    // CUOPT_METHOD: PDLP is this library's engine; small LPs have a second one, the dual simplex of dual_simplex.cpp (below).
    // Where that one is off, abstains or finds the LP too large, Concurrent (the default) and DualSimplex requests are
    // served by PDLP and say so in the log and in cuOptAmdGetSolveInfo; crossover = true is ignored (and said).  The reference's
    // Concurrent / DualSimplex return the simplex VERTEX on small LPs (the CPU simplex wins the race there:
    // c_api_test.c:761-873 expects 32.0 +- 1e-3 at default settings), so for such requests on small LPs (<= 1e5
    // nonzeros, microseconds per iteration) PDLP aims at simplex-grade tolerances (1e-8) -- the "simplex-grade
    // emulation", switched off by the integer parameter "amd_simplex_grade" = 0.
    // The caller's own tolerances stay in force as the ACCEPTANCE set (cuoptamd_settings::accept_tolerance): the
    // first iterate that meets them is kept and returned as Optimal if the tight solve runs out of the caller's
    // iteration / time limit or of the emulation's budget, so limits behave as in the reference's Concurrent mode.
    const bool other_method  = s->method != CUOPT_METHOD_PDLP;
    const bool grade_on      = (s->simplex_grade >= 0 ? s->simplex_grade : (int)cuopt_amd::tune_int("simplex_grade", 1)) != 0;
    cuoptamd_lp lp{p->m, p->n, p->offsets.data(), p->indices.data(), p->values.data(), p->c.data(),
                   p->lo.data(), p->hi.data(), p->lb.data(), p->ub.data(), p->maximize ? 1 : 0,
                   p->objective_offset};
    // Round 3: a SECOND engine, an own bounded dual simplex on the host (dual_simplex.cpp: sparse LU; the reference's second engine is its
    // CPU dual simplex, LP/solve.cu:295-347).  CUOPT_METHOD_DUAL_SIMPLEX: it answers, if it can (<= 200 000 rows and 4e6 nonzeros by default; it abstains on
    // numerical trouble) -- otherwise PDLP serves the request as before.  CUOPT_METHOD_CONCURRENT: it races PDLP (a host thread
    // against the GPU; whoever finishes first with a verdict answers, the other one is cancelled: LP/solve.cu:383-443).
    // CUOPT_AMD_DUAL_SIMPLEX=0 / "amd_dual_simplex" = 0 switch it off (then: the simplex-grade emulation below).  No GPU -> the
    // engine is not consulted either: this library has no CPU-only mode.
    const bool engine_on = other_method && gpus == 1 && pdlpdev_device_count() >= 1 &&
                           (s->dual_simplex >= 0 ? s->dual_simplex : env_int("CUOPT_AMD_DUAL_SIMPLEX", 1)) != 0;
    struct SimplexRun {
      int32_t status = 8, iterations = 0;
      double objective = 0.0, seconds = 0.0;
      std::vector<double> x, y, rc;
      int32_t cancel = 0;  // written / read through the atomic builtins only (the engine's C interface takes a plain int32_t*)
      void set_cancel(int32_t v) { __atomic_store_n(&cancel, v, __ATOMIC_RELEASE); }
      bool cancelled() const { return __atomic_load_n(&cancel, __ATOMIC_ACQUIRE) != 0; }
      std::atomic<int> done{0};
      bool conclusive() const { return status == 1 || status == 2 || status == 3; }
    } sx;
    if (engine_on || s->crossover) sx.x.assign(p->n, 0.0), sx.y.assign(p->m, 0.0), sx.rc.assign(p->n, 0.0);  // (before any thread exists)
    auto run_simplex = [&](double tlim, int32_t itlim, const double* from_x = nullptr, const double* from_y = nullptr) {
      const auto t0 = std::chrono::steady_clock::now();
      if (tlim <= 0.0 || itlim <= 0) {  // no budget at all: the limit is the verdict (0 means "none" to the engine's own interface)
        sx.status = tlim <= 0.0 ? 6 : 5;
        sx.done.store(1);
        return;
      }
      if (from_x)
        (void)cuoptamd_dual_simplex_from(&lp, from_x, from_y, std::isfinite(tlim) ? tlim : 0.0, itlim == INT_MAX ? 0 : itlim, &sx.cancel, &sx.status,
                                         &sx.iterations, &sx.objective, sx.x.data(), sx.y.data(), sx.rc.data());
      else
        (void)cuoptamd_dual_simplex(&lp, std::isfinite(tlim) ? tlim : 0.0, itlim == INT_MAX ? 0 : itlim, &sx.cancel, &sx.status, &sx.iterations,
                                    &sx.objective, sx.x.data(), sx.y.data(), sx.rc.data());
      sx.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      sx.done.store(1);
    };
    bool engine_ran = false, engine_answered = false;
    if (engine_on && s->method == CUOPT_METHOD_DUAL_SIMPLEX) {
      run_simplex(s->time_limit, s->iteration_limit);
      engine_ran      = sx.status != 8;
      engine_answered = sx.conclusive() || sx.status == 5 || sx.status == 6;  // its own limits count as its answer
    }
    const bool racing        = engine_on && s->method == CUOPT_METHOD_CONCURRENT;
    // (also while racing: a simplex that abstains on a small LP leaves the request to PDLP at simplex-grade tolerances, as before round 3)
    const bool simplex_grade = other_method && grade_on && gpus == 1 && p->values.size() <= 100000 && !engine_answered;
    st.iteration_limit         = s->iteration_limit;
    // a simplex that used part of the caller's time and then abstained: PDLP gets what is left of it
    st.time_limit              = engine_ran && std::isfinite(s->time_limit) ? std::max(1e-3, s->time_limit - sx.seconds) : s->time_limit;
    st.per_constraint_residual = s->per_constraint_residual;
    st.first_primal_feasible   = s->first_primal_feasible;
    // a simplex would prove infeasibility / unboundedness: the emulation runs PDLP WITH its infeasibility detection
    // (small LPs only -- large ones keep the reference PDLP's default, detection off unless asked for)
    st.detect_infeasibility        = s->infeasibility_detection || simplex_grade;
    st.strict_infeasibility        = s->strict_infeasibility;
    st.unbounded_from_feasible_iterates = simplex_grade;  // a simplex would say UNBOUNDED
    st.primal_infeasible_tolerance = s->primal_infeasible_tolerance;
    st.dual_infeasible_tolerance   = s->dual_infeasible_tolerance;
    st.save_best_primal_so_far     = s->save_best_primal_so_far;
    st.log_to_console              = s->log_to_console;
    st.log_file                    = s->log_file.empty() ? nullptr : s->log_file.c_str();
    auto say = [&](const std::string& line) {
      if (s->log_to_console) std::fputs(line.c_str(), stdout), std::fflush(stdout);
      if (!s->log_file.empty())
        if (FILE* f = std::fopen(s->log_file.c_str(), "a")) std::fputs(line.c_str(), f), std::fclose(f);
    };
    const char* method_name = s->method == CUOPT_METHOD_CONCURRENT ? "Concurrent" : s->method == CUOPT_METHOD_DUAL_SIMPLEX ? "DualSimplex" : "PDLP";
    if (other_method || s->crossover)
      say(std::string("cuopt_amd: method ") + method_name + (s->crossover ? " + crossover (the dual simplex from the basis PDLP's point suggests; not done beyond its size limits)" : "") + " requested: " +
          (engine_answered ? "answered by the dual simplex\n"
           : racing        ? "the dual simplex (host thread) races PDLP (GPU)\n"
                           : std::string("served by PDLP") + (engine_ran ? " (the dual simplex abstained)" : "") +
                                 (simplex_grade ? ", simplex-grade tolerances 1e-8 with the requested ones as acceptance set\n" : "\n")));
    const cuoptamd_settings st_user = st;
    const int32_t kSimplexGradeBudget = (int32_t)std::max<long long>(1, cuopt_amd::tune_int("simplex_grade_budget", 50000));  // (tests shrink it)
    bool tightened = false;
    if (simplex_grade) {
      double* tols[6]   = {&st.absolute_gap_tolerance,    &st.relative_gap_tolerance,  &st.absolute_primal_tolerance,
                           &st.relative_primal_tolerance, &st.absolute_dual_tolerance, &st.relative_dual_tolerance};
      for (int i = 0; i < 6; ++i) {
        st.accept_tolerance[i] = *tols[i];
        tightened              = tightened || *tols[i] > 1e-8;
        *tols[i]               = std::min(*tols[i], 1e-8);
      }
      if (tightened) {
        st.accept_enabled  = 1;
        st.iteration_limit = std::min(st.iteration_limit, kSimplexGradeBudget);
      }
    }
    cuoptamd_result res{};
    double first_attempt_seconds = 0.0;
    int32_t first_attempt_steps = 0, first_attempt_attempts = 0;  // work of a simplex-grade attempt that a second solve followed
    bool second_leg = false;  // ... inside a Concurrent race (the simplex kept running)
    std::string answered = simplex_grade && tightened ? "simplex_grade_1e-8" : "requested_tolerances";
    sol->x.assign(p->n, 0.0), sol->y.assign(p->m, 0.0), sol->rc.assign(p->n, 0.0);
    auto take_simplex = [&]() {  // the dual simplex's verdict as the solve's result
      static const int map[10] = {0, CUOPT_TERIMINATION_STATUS_OPTIMAL, CUOPT_TERIMINATION_STATUS_INFEASIBLE, CUOPT_TERIMINATION_STATUS_UNBOUNDED, 0,
                                  CUOPT_TERIMINATION_STATUS_ITERATION_LIMIT, CUOPT_TERIMINATION_STATUS_TIME_LIMIT, CUOPT_TERIMINATION_STATUS_NUMERICAL_ERROR,
                                  CUOPT_TERIMINATION_STATUS_NUMERICAL_ERROR, CUOPT_TERIMINATION_STATUS_NUMERICAL_ERROR};
      res                  = cuoptamd_result{};
      res.status           = map[std::max(0, std::min(9, (int)sx.status))];
      res.steps_taken      = res.attempted_steps = sx.iterations;
      res.primal_objective = res.dual_objective = sx.status == 1 ? sx.objective : 0.0;
      res.loop_seconds     = sx.seconds;
      res.gpus             = 1;
      if (sx.status == 1) sol->x = sx.x, sol->y = sx.y, sol->rc = sx.rc;
      answered = "dual_simplex";
    };
    if (engine_answered) {
      take_simplex();
    } else if (gpus > 1) {
      const int soft = (int)cuopt_amd::tune_int("soft_communicator", 0)  /* tests: ranks = contexts on one device */;
      const int rc   = cuoptamd_solve_sharded(&lp, &hyper, &st, gpus, soft, &res, sol->x.data(), sol->y.data(), sol->rc.data());
      if (rc != 0) {
        const std::string msg = cuoptamd_last_error();
        if (rc == -7) return error(CUOPT_VALIDATION_ERROR, "ValidationError", msg);
        return error(CUOPT_RUNTIME_ERROR, "RuntimeError", msg);
      }
    } else {
      cuoptamd_solver* solver = nullptr;
      int rc = cuoptamd_solver_create(&solver, &lp, &hyper, &st, nullptr, nullptr, 0, 0, 1, nullptr);
      if (rc != 0) {
        std::string msg = cuoptamd_last_error();
        cuoptamd_solver_destroy(solver);
        if (rc == -7) return error(CUOPT_VALIDATION_ERROR, "ValidationError", msg);
        return error(CUOPT_RUNTIME_ERROR, "RuntimeError", msg);
      }
      if (racing) {
        // Concurrent: the simplex on a host thread, PDLP here in batches of a few major iterations; the first verdict wins
        std::thread worker([&] {
          try {
            run_simplex(s->time_limit, s->iteration_limit);
          } catch (...) {  // (the engine's entry points catch their own; nothing may leave a thread)
            sx.status = 7;
            sx.done.store(1);
          }
        });
        for (;;) {
          for (;;) {
            if (sx.done.load() && sx.conclusive()) break;
            rc = cuoptamd_solver_advance(solver, 400, &res);
            if (rc != 0 || res.status != CUOPT_TERIMINATION_STATUS_NO_TERMINATION) break;
          }
          // The simplex-grade attempt ran out of ITS OWN budget (not a limit of the caller's) with nothing accepted and the simplex is
          // still at work: the race goes on -- PDLP at the requested tolerances with what is left of the caller's limits against the
          // same simplex run (round-4 advisor: the internal budget used to cancel the simplex and a second, unraced solve followed)
          const bool own_budget = rc == 0 && tightened && !second_leg && res.status == CUOPT_TERIMINATION_STATUS_ITERATION_LIMIT &&
                                  !res.accepted_at_looser_tolerances && st_user.iteration_limit > kSimplexGradeBudget &&
                                  !(sx.done.load() && sx.conclusive());
          if (!own_budget) break;
          first_attempt_seconds = res.setup_seconds + res.loop_seconds;
          first_attempt_steps = res.steps_taken, first_attempt_attempts = res.attempted_steps;
          cuoptamd_settings st_rest = st_user;
          if (st_user.iteration_limit != INT_MAX) st_rest.iteration_limit = std::max(0, st_user.iteration_limit - res.steps_taken);
          if (std::isfinite(st_rest.time_limit)) st_rest.time_limit = std::max(0.0, st_rest.time_limit - first_attempt_seconds);
          if (!(st_rest.iteration_limit > 0 && st_rest.time_limit > 0.0)) break;
          second_leg = true;
          answered   = "requested_tolerances_after_simplex_grade_budget";
          rc         = cuoptamd_solver_reset(solver, nullptr, nullptr, nullptr, nullptr, &st_rest, nullptr, nullptr);
          if (rc != 0) break;
        }
        const bool pdlp_done = rc == 0 && res.status != CUOPT_TERIMINATION_STATUS_NO_TERMINATION;
        // whenever the loop ends without a finished, conclusive simplex -- PDLP has its verdict, hit a limit, or FAILED -- the simplex
        // is told to stop (round-3 advisor: after a failing advance the join used to wait for the simplex to finish on its own)
        if (!(sx.done.load() && sx.conclusive())) sx.set_cancel(1);
        worker.join();
        engine_ran = sx.status != 8;
        // PDLP's Optimal / infeasibility verdicts stand; a limit or an error of PDLP's is overruled by a verdict of the simplex
        const bool pdlp_verdict = pdlp_done && (res.status == CUOPT_TERIMINATION_STATUS_OPTIMAL || res.status == CUOPT_TERIMINATION_STATUS_INFEASIBLE ||
                                                res.status == CUOPT_TERIMINATION_STATUS_UNBOUNDED);
        if (rc == 0 && sx.conclusive() && !(pdlp_verdict && sx.cancelled())) {
          engine_answered = true;
          cuoptamd_solver_destroy(solver);
          solver = nullptr;
          take_simplex();
        }
      } else {
        rc = cuoptamd_solver_advance(solver, INT_MAX, &res);
      }
      if (!engine_answered) {
      const bool on_limit = res.status == CUOPT_TERIMINATION_STATUS_ITERATION_LIMIT || res.status == CUOPT_TERIMINATION_STATUS_TIME_LIMIT;
      if (rc == 0 && res.accepted_at_looser_tolerances) answered = "requested_tolerances_kept_during_simplex_grade_attempt";
      if (rc == 0 && tightened && on_limit && !res.accepted_at_looser_tolerances && !second_leg) {
        // the tight attempt ended on a limit and no iterate met even the requested tolerances: whatever is left of the
        // caller's limits goes to a plain solve at the requested tolerances (same solver object: matrices and scaling kept)
        first_attempt_seconds = res.setup_seconds + res.loop_seconds;
        first_attempt_steps = res.steps_taken, first_attempt_attempts = res.attempted_steps;
        cuoptamd_settings st_rest = st_user;
        if (st_user.iteration_limit != INT_MAX) st_rest.iteration_limit = std::max(0, st_user.iteration_limit - res.steps_taken);
        if (std::isfinite(st_rest.time_limit)) st_rest.time_limit = std::max(0.0, st_rest.time_limit - first_attempt_seconds);
        if (st_rest.iteration_limit > 0 && st_rest.time_limit > 0.0) {
          answered = "requested_tolerances_after_simplex_grade_budget";
          rc = cuoptamd_solver_reset(solver, nullptr, nullptr, nullptr, nullptr, &st_rest, nullptr, nullptr);
          if (rc == 0) rc = cuoptamd_solver_advance(solver, INT_MAX, &res);
        } else {
          answered = "limit_reached_during_simplex_grade_attempt";
        }
      }
      if (rc == 0) rc = cuoptamd_solver_get_solution(solver, sol->x.data(), sol->y.data(), sol->rc.data());
      if (rc != 0) {
        std::string msg = cuoptamd_last_error();
        cuoptamd_solver_destroy(solver);
        return error(CUOPT_RUNTIME_ERROR, "RuntimeError", msg);
      }
      cuoptamd_solver_destroy(solver);
      }
    }
    // crossover (LP/solve.cu:467-547: from PDLP's point to a basic solution): the dual simplex started from the basis PDLP's point
    // suggests (cuoptamd_dual_simplex_from: the variables and rows strictly inside their bounds, then the ones with the smallest
    // reduced costs; repaired with slacks) -- a few pivots when the point is close to a vertex.  Its vertex replaces PDLP's point
    // when it confirms the objective; LPs beyond the engine's size limits keep PDLP's point, and the solve info says which it was.
    const char* crossover_by = "none";
    if (s->crossover && !engine_answered && gpus == 1 && res.status == CUOPT_TERIMINATION_STATUS_OPTIMAL && pdlpdev_device_count() >= 1 &&
        (s->dual_simplex >= 0 ? s->dual_simplex : env_int("CUOPT_AMD_DUAL_SIMPLEX", 1)) != 0) {
      sx.set_cancel(0);
      const std::vector<double> px(sol->x), py(sol->y);
      run_simplex(std::isfinite(s->time_limit) ? std::max(1e-3, s->time_limit - (res.setup_seconds + res.loop_seconds)) : s->time_limit, INT_MAX, px.data(), py.data());
      engine_ran = engine_ran || sx.status != 8;
      if (sx.status == 1 && std::fabs(sx.objective - res.primal_objective) <= 1e-2 * (1.0 + std::fabs(res.primal_objective))) {
        sol->x = sx.x, sol->y = sx.y, sol->rc = sx.rc;
        res.primal_objective = res.dual_objective = sx.objective;
        res.gap = res.relative_gap = 0.0;
        res.loop_seconds += sx.seconds;
        crossover_by = "dual_simplex_from_the_pdlp_point";
      } else {
        crossover_by = sx.status == 8 ? "not_done_lp_too_large_for_the_dual_simplex" : "not_done_dual_simplex_disagreed_or_abstained";
      }
    } else if (s->crossover) {
      crossover_by = engine_answered ? "not_needed_vertex_from_the_dual_simplex" : "not_done";
    }
    {
      char info[1024];
      std::snprintf(info, sizeof info,
                    "{\"engine\": \"%s\", \"requested_method\": \"%s\", \"crossover_requested\": %s, \"simplex_grade_emulation\": %s, "
                    "\"dual_simplex_consulted\": %s, \"dual_simplex_status\": %d, \"crossover\": \"%s\", "
                    "\"answered_by\": \"%s\", \"gpus\": %d, \"iterations\": %d, \"simplex_grade_attempt_iterations\": %d}",
                    engine_answered ? "dual_simplex" : "pdlp", method_name, s->crossover ? "true" : "false", simplex_grade ? "true" : "false",
                    engine_ran ? "true" : "false", (int)sx.status, crossover_by, answered.c_str(), gpus,
                    res.steps_taken + (answered == "requested_tolerances_after_simplex_grade_budget" ? first_attempt_steps : 0),
                    answered == "requested_tolerances_after_simplex_grade_budget" ? first_attempt_steps : 0);
      sol->solve_info = info;
      if (other_method || s->crossover) say("cuopt_amd: " + sol->solve_info + "\n");
    }
    // a second solve after the simplex-grade budget starts over (its trajectory at the looser tolerances is a different one):
    // the iterations of BOTH are what the call cost, and that is what the statistics and the solve info report
    if (answered != "requested_tolerances_after_simplex_grade_budget") first_attempt_steps = first_attempt_attempts = 0;
    res.steps_taken += first_attempt_steps, res.attempted_steps += first_attempt_attempts;
    res.setup_seconds += first_attempt_seconds;
    sol->stats              = res;
    sol->termination_status = res.status;
    sol->objective          = res.primal_objective;  // solver_solution.cu:307-310
    sol->solve_time         = res.setup_seconds + res.loop_seconds;
    if (!s->solution_file.empty()) {
      // write_to_sol_file (solver_solution.cu:370-387, math_optimization/solution_writer.cu)
      if (FILE* f = std::fopen(s->solution_file.c_str(), "w")) {
        const bool ok = res.status == CUOPT_TERIMINATION_STATUS_OPTIMAL || res.status == CUOPT_TERIMINATION_STATUS_PRIMAL_FEASIBLE;
        std::fprintf(f, "# Status: %s\n", !ok ? "Infeasible" : (res.status == CUOPT_TERIMINATION_STATUS_OPTIMAL ? "Optimal" : "PrimalFeasible"));
        if (ok) {
          std::fprintf(f, "# Objective value: %.18g\n", sol->objective);
          for (size_t j = 0; j < p->var_names.size() && j < sol->x.size(); ++j)
            std::fprintf(f, "%s %.18g\n", p->var_names[j].c_str(), sol->x[j]);
        }
        std::fclose(f);
      }
    }
  } catch (const std::bad_alloc&) {
    return error(CUOPT_OUT_OF_MEMORY, "OutOfMemoryError", "out of host memory");
  } catch (const std::exception& e) {
    return error(CUOPT_RUNTIME_ERROR, "RuntimeError", e.what());
  }
  return CUOPT_SUCCESS;
}

// Extension (not part of the reference's 41 functions): a JSON line saying which engine answered the request --
// {"engine", "requested_method", "crossover_requested", "simplex_grade_emulation", "answered_by", "gpus", "iterations"}.
cuopt_int_t cuOptAmdGetSolveInfo(cuOptSolution solution, char* buffer, cuopt_int_t buffer_size)
{
  if (solution == nullptr || buffer == nullptr || buffer_size <= 0) return CUOPT_INVALID_ARGUMENT;
  const Solution* sol = static_cast<const Solution*>(solution);
  std::snprintf(buffer, (size_t)buffer_size, "%s", sol->solve_info.c_str());
  return CUOPT_SUCCESS;
}

void cuOptDestroySolution(cuOptSolution* solution_ptr)
{
  if (solution_ptr == nullptr || *solution_ptr == nullptr) return;
  delete static_cast<Solution*>(*solution_ptr);
  *solution_ptr = nullptr;
}

#define SOLUTION_GETTER_PROLOGUE(out)                                         \
  if (solution == nullptr || (out) == nullptr) return CUOPT_INVALID_ARGUMENT; \
  const Solution* sol = static_cast<const Solution*>(solution)

cuopt_int_t cuOptGetTerminationStatus(cuOptSolution solution, cuopt_int_t* termination_status_ptr)
{
  SOLUTION_GETTER_PROLOGUE(termination_status_ptr);
  *termination_status_ptr = sol->termination_status;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetErrorStatus(cuOptSolution solution, cuopt_int_t* error_status_ptr)
{
  SOLUTION_GETTER_PROLOGUE(error_status_ptr);
  *error_status_ptr = sol->error_status;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetErrorString(cuOptSolution solution, char* error_string_ptr, cuopt_int_t error_string_size)
{
  SOLUTION_GETTER_PROLOGUE(error_string_ptr);
  if (error_string_size <= 0) return CUOPT_INVALID_ARGUMENT;
  std::snprintf(error_string_ptr, (size_t)error_string_size, "%s", sol->error_message.c_str());
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetPrimalSolution(cuOptSolution solution, cuopt_float_t* solution_values)
{
  SOLUTION_GETTER_PROLOGUE(solution_values);
  std::copy(sol->x.begin(), sol->x.end(), solution_values);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetObjectiveValue(cuOptSolution solution, cuopt_float_t* objective_value_ptr)
{
  SOLUTION_GETTER_PROLOGUE(objective_value_ptr);
  *objective_value_ptr = sol->objective;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetSolveTime(cuOptSolution solution, cuopt_float_t* solve_time_ptr)
{
  SOLUTION_GETTER_PROLOGUE(solve_time_ptr);
  *solve_time_ptr = sol->solve_time;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetMIPGap(cuOptSolution solution, cuopt_float_t* mip_gap_ptr)
{
  SOLUTION_GETTER_PROLOGUE(mip_gap_ptr);
  if (!sol->is_mip) return CUOPT_INVALID_ARGUMENT;
  *mip_gap_ptr = kInf;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetSolutionBound(cuOptSolution solution, cuopt_float_t* solution_bound_ptr)
{
  SOLUTION_GETTER_PROLOGUE(solution_bound_ptr);
  if (!sol->is_mip) return CUOPT_INVALID_ARGUMENT;
  *solution_bound_ptr = -kInf;
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetDualSolution(cuOptSolution solution, cuopt_float_t* dual_solution_ptr)
{
  SOLUTION_GETTER_PROLOGUE(dual_solution_ptr);
  if (sol->is_mip) return CUOPT_INVALID_ARGUMENT;
  std::copy(sol->y.begin(), sol->y.end(), dual_solution_ptr);
  return CUOPT_SUCCESS;
}
cuopt_int_t cuOptGetReducedCosts(cuOptSolution solution, cuopt_float_t* reduced_cost_ptr)
{
  SOLUTION_GETTER_PROLOGUE(reduced_cost_ptr);
  if (sol->is_mip) return CUOPT_INVALID_ARGUMENT;
  std::copy(sol->rc.begin(), sol->rc.end(), reduced_cost_ptr);
  return CUOPT_SUCCESS;
}

// Extension: the counterpart of CUOPT_SOLUTION_FILE -- reads a .sol file (what write_to_sol_file produces, or the MIPLIB
// flavour) back into the variable order of `problem`.  Behaviour of the reference's reader
// (cpp/src/math_optimization/solution_reader.cu:57-145): '#' / '=' lines may carry "objective value" / "obj" and "status:",
// every other non-empty line is "name value"; a variable of the problem that the file does not mention is an error.
cuopt_int_t cuOptAmdReadSolutionFile(cuOptOptimizationProblem problem, const char* filename, cuopt_float_t* values,
                                     cuopt_float_t* objective_value, char* status, cuopt_int_t status_size)
{
  if (problem == nullptr || filename == nullptr || values == nullptr) return CUOPT_INVALID_ARGUMENT;
  const Problem* p = static_cast<const Problem*>(problem);
  if (p->var_names.size() != (size_t)p->n) return CUOPT_VALIDATION_ERROR;  // names come from an MPS file
  FILE* f = std::fopen(filename, "r");
  if (!f) return CUOPT_MPS_FILE_ERROR;
  std::vector<std::pair<std::string, double>> found;
  double obj = std::numeric_limits<double>::quiet_NaN();
  std::string stat;
  char line[4096];
  auto lower = [](std::string t) {
    for (char& c : t) c = (char)std::tolower((unsigned char)c);
    return t;
  };
  while (std::fgets(line, sizeof line, f)) {
    std::string t(line);
    while (!t.empty() && (t.back() == '\n' || t.back() == '\r')) t.pop_back();
    if (t.empty()) continue;
    if (t[0] == '#' || t[0] == '=') {
      const std::string lo = lower(t);
      size_t pos = lo.find("objective value");
      if (pos == std::string::npos) pos = lo.find("obj");
      if (pos != std::string::npos) {
        const size_t at = t.find_first_of("-+.0123456789", pos);
        if (at != std::string::npos) obj = std::strtod(t.c_str() + at, nullptr);
        continue;
      }
      pos = lo.find("status:");
      if (pos != std::string::npos) {
        const size_t a = t.find_first_not_of(" \t:", pos + 7);
        if (a != std::string::npos) stat = t.substr(a, t.find_first_of(" \t", a) - a);
      }
      continue;
    }
    char name[2048];
    double v;
    if (std::sscanf(t.c_str(), "%2047s %lf", name, &v) == 2) found.emplace_back(name, v);
  }
  std::fclose(f);
  std::sort(found.begin(), found.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  for (int32_t j = 0; j < p->n; ++j) {
    auto it = std::lower_bound(found.begin(), found.end(), p->var_names[j],
                               [](const auto& a, const std::string& key) { return a.first < key; });
    if (it == found.end() || it->first != p->var_names[j]) return CUOPT_VALIDATION_ERROR;  // "Variable not found in solution"
    // (a name listed twice: the last occurrence wins, like the reference's map assignment)
    auto last = it;
    while (last + 1 != found.end() && (last + 1)->first == it->first) ++last;
    values[j] = last->second;
  }
  if (objective_value) *objective_value = obj;
  if (status && status_size > 0) std::snprintf(status, (size_t)status_size, "%s", stat.c_str());
  return CUOPT_SUCCESS;
}

// Extension: variable / row names of a problem read from an MPS file (the reference exposes them through its Python data
// model, data_model.py:592-600; its C API has no getter).  kind 0 = variable, 1 = constraint row.
cuopt_int_t cuOptAmdGetName(cuOptOptimizationProblem problem, cuopt_int_t kind, cuopt_int_t index, char* buffer,
                            cuopt_int_t buffer_size)
{
  if (problem == nullptr || buffer == nullptr || buffer_size <= 0 || (kind != 0 && kind != 1)) return CUOPT_INVALID_ARGUMENT;
  const Problem* p = static_cast<const Problem*>(problem);
  const std::vector<std::string>& names = kind == 0 ? p->var_names : p->row_names;
  if (index < 0 || (size_t)index >= names.size()) return CUOPT_INVALID_ARGUMENT;
  std::snprintf(buffer, (size_t)buffer_size, "%s", names[(size_t)index].c_str());
  return CUOPT_SUCCESS;
}

// Not part of the reference ABI: full PDLP statistics of a solution (additional_termination_information_t
// is only reachable through the C++/Python API in the reference).  Used by tests and benches.
cuopt_int_t cuOptAmdGetPdlpStats(cuOptSolution solution, cuoptamd_result* stats)
{
  SOLUTION_GETTER_PROLOGUE(stats);
  *stats = sol->stats;
  return CUOPT_SUCCESS;
}

}  // extern "C"
