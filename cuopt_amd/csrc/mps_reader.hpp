// MPS (free format) reader feeding cuOptReadProblem.
// Behavioural reference: cuOpt 25.08 cpp/libmps_parser/src/mps_parser.cpp (parse_string :304-512,
// fill_problem :96-267) as used by cuOptReadProblem with input_mps_strict=false
// (cpp/src/linear_programming/cuopt_c.cpp:62-88).  Own implementation (token based), checked
// against the reference parser on every file of datasets/linear_programming (tests/test_mps_reader.py).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

namespace cuopt_amd {

struct MpsError : std::runtime_error {
  bool cannot_open;  // true -> CUOPT_MPS_FILE_ERROR, false -> CUOPT_MPS_PARSE_ERROR
  MpsError(bool open_failure, const std::string& msg) : std::runtime_error(msg), cannot_open(open_failure) {}
};

struct MpsModel {
  std::string problem_name, objective_name;
  bool maximize           = false;
  double objective_offset = 0.0;
  // CSR by constraint row (the objective row is not a constraint), entries in file order
  std::vector<int> offsets{0}, indices;
  std::vector<double> values;
  std::vector<double> c, lb, ub;       // per variable
  std::vector<double> rhs, lo, hi;     // per constraint
  std::vector<char> row_types;         // 'E','L','G'
  std::vector<char> var_types;         // 'C','I'
  std::vector<std::string> row_names, var_names;
};

MpsModel read_mps_file(const std::string& path);

}  // namespace cuopt_amd
