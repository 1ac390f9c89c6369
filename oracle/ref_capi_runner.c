/*
 * TEST INFRASTRUCTURE ONLY.  Drives the REFERENCE's own pure-C API test translation unit
 * (cpp/tests/linear_programming/c_api_tests/c_api_test.c, compiled IN PLACE from /root/reference by
 * oracle/Makefile -- nothing is copied) against OUR libcuopt.so, the way the reference's gtest wrapper
 * c_api_tests.cpp:31-97 drives it.  Prints one "name expected got" line per check and exits non-zero if
 * any check that is in scope fails.  usage: ref_capi_runner <afiro.mps>
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "c_api_tests.h"

static int failures = 0;
static void check(const char* name, long expected, long got, int in_scope)
{
  const int ok = expected == got;
  printf("%-28s expected %ld got %ld %s\n", name, expected, got, ok ? "OK" : (in_scope ? "FAIL" : "DIFFERS(out of scope)"));
  if (!ok && in_scope) ++failures;
}

int main(int argc, char** argv)
{
  if (argc < 2) {
    printf("usage: %s afiro.mps\n", argv[0]);
    return 2;
  }
  cuopt_int_t status = -1;
  cuopt_float_t objective = 0.0, solve_time = 0.0;
  check("int_size", 4, test_int_size(), 1);                                         /* c_api_tests.cpp:27 */
  check("float_size", 8, test_float_size(), 1);                                     /* :29 */
  check("afiro_rc", CUOPT_SUCCESS, solve_mps_file(argv[1], 60, CUOPT_INFINITY, &status, &solve_time, CUOPT_METHOD_DUAL_SIMPLEX), 1); /* :31-39 */
  check("afiro_status", CUOPT_TERIMINATION_STATUS_OPTIMAL, status, 1);
  check("afiro_pdlp_rc", CUOPT_SUCCESS, solve_mps_file(argv[1], 60, CUOPT_INFINITY, &status, &solve_time, CUOPT_METHOD_PDLP), 1);
  check("afiro_pdlp_status", CUOPT_TERIMINATION_STATUS_OPTIMAL, status, 1);
  check("iteration_limit_rc", CUOPT_SUCCESS, solve_mps_file(argv[1], 60, 1, &status, &solve_time, CUOPT_METHOD_DUAL_SIMPLEX), 1); /* :73-80 */
  check("iteration_limit_status", CUOPT_TERIMINATION_STATUS_ITERATION_LIMIT, status, 1);
  check("bad_parameter_name", CUOPT_INVALID_ARGUMENT, test_bad_parameter_name(), 1); /* :82 */
  check("missing_file", CUOPT_MPS_FILE_ERROR, test_missing_file(), 1);               /* :86 */
  check("infeasible_problem", CUOPT_SUCCESS, test_infeasible_problem(), 1);          /* :88 */
  check("ranged_rc", CUOPT_SUCCESS, test_ranged_problem(&status, &objective), 1);    /* :90-97 */
  check("ranged_status", CUOPT_TERIMINATION_STATUS_OPTIMAL, status, 1);
  check("ranged_objective_32", 1, fabs(objective - 32.0) <= 1e-3, 1);
  /* MILP is outside this library's scope: the knapsack MIP must be rejected, not mis-solved */
  check("burglar_mip", CUOPT_SUCCESS, burglar_problem(), 0);                         /* :84 */
  printf("failures %d\n", failures);
  return failures ? 1 : 0;
}
