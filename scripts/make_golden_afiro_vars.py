#!/usr/bin/env python
"""Extracts the known-answer vector of the reference's test_parse_var_names (afiro: variable names + the primal solution
returned by cuOpt's PDLP at default settings) into tests/golden/afiro_pdlp_vars.json.  Needs /root/reference."""
import json
import re

src = open("/root/reference/python/cuopt/cuopt/tests/linear_programming/test_lp_solver.py").read()
blk = src[src.index("def test_parse_var_names"):src.index("def test_parser_and_batch_solver")]
names = re.findall(r'^\s+"(X\d+)",$', blk[blk.index("expected_names"):blk.index("for i, name")], flags=re.M)
vals = dict((k, float(v)) for k, v in re.findall(r'"(X\d+)":\s*([-0-9.e]+)', blk[blk.index("expected_dict"):]))
json.dump(dict(source="python/cuopt/cuopt/tests/linear_programming/test_lp_solver.py:386-475 (test_parse_var_names): variable "
               "names of afiro_original.mps and the primal solution the reference's PDLP returns at default settings "
               "(method = PDLP, tolerances 1e-4), compared there with rel = 1e-4",
               generated_by="scripts/make_golden_afiro_vars.py", expected_names=names, expected_values=vals),
          open("tests/golden/afiro_pdlp_vars.json", "w"), indent=1)
