// Free-format MPS reader (see mps_reader.hpp for the behavioural reference).
#include "mps_reader.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <set>
#include <string_view>
#include <unordered_map>

namespace cuopt_amd {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

[[noreturn]] void parse_error(const std::string& msg) { throw MpsError(false, msg); }

// whitespace-separated fields of one data line; a field starting with '$' comments out the rest
std::vector<std::string_view> fields_of(std::string_view line)
{
  std::vector<std::string_view> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) ++i;
    if (i >= line.size()) break;
    size_t j = i;
    while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') ++j;
    out.push_back(line.substr(i, j - i));
    i = j;
  }
  return out;
}

double number(std::string_view tok, const char* where, std::string_view line)
{
  std::string s(tok);
  char* end = nullptr;
  double v  = std::strtod(s.c_str(), &end);
  if (end == s.c_str()) parse_error(std::string("Bad value found in ") + where + "! line=" + std::string(line));
  return v;
}

enum class Section { None, Rows, Columns, Rhs, Bounds, Ranges, ObjSense, ObjName };

bool starts_with(std::string_view s, const char* prefix)
{
  return s.rfind(prefix, 0) == 0;
}

struct Builder {
  MpsModel model;
  std::unordered_map<std::string, int> row_id, var_id;
  std::set<std::string> ignored_objectives, sections_seen;
  std::set<int> bounded_vars;
  std::vector<std::vector<int>> row_cols;
  std::vector<std::vector<double>> row_vals;
  std::vector<double> range;  // NaN = not given
  std::vector<bool> range_given;
  bool in_integer_block = false, bounds_initialised = false;

  void objsense(std::string_view word, std::string_view line)
  {
    if (word == "MIN" || word == "MINIMIZE")
      model.maximize = false;
    else if (word == "MAX" || word == "MAXIMIZE")
      model.maximize = true;
    else
      parse_error("Invalid objective sense in OBJSENSE section! line=" + std::string(line));
  }
  void objname(std::string_view word)
  {
    if (!model.objective_name.empty()) parse_error("OBJNAME section should appear before ROWS section");
    model.objective_name = std::string(word);
  }

  void row_line(std::string_view line)
  {
    auto f = fields_of(line);
    if (f.size() < 2) parse_error("ROWS entries need a type and a name! line=" + std::string(line));
    const char type = f[0][0];
    std::string name(f[1]);
    if (type == 'N') {  // first free row is the objective, further ones are dropped
      if (model.objective_name.empty())
        model.objective_name = name;
      else
        ignored_objectives.insert(name);
      return;
    }
    if (type != 'E' && type != 'L' && type != 'G') parse_error("Unsupported row type! line=" + std::string(line));
    if (row_id.count(name)) parse_error("Duplicate row named '" + name + "' found! line=" + std::string(line));
    row_id.emplace(name, (int)model.row_names.size());
    model.row_names.push_back(name);
    model.row_types.push_back(type);
  }

  void entry(int var, std::string_view row, std::string_view num, std::string_view line)
  {
    std::string r(row);
    if (ignored_objectives.count(r)) return;
    const double v = number(num, "COLUMNS", line);
    if (r == model.objective_name) {
      model.c[var] = v;
      return;
    }
    auto it = row_id.find(r);
    if (it == row_id.end()) parse_error("Bad row name found '" + r + "' in COLUMNS! line=" + std::string(line));
    row_cols[it->second].push_back(var);
    row_vals[it->second].push_back(v);
  }

  void column_line(std::string_view line)
  {
    auto f = fields_of(line);
    if (f.empty()) return;
    if (line.find("'MARKER'") != std::string_view::npos) {
      if (line.find("INTORG") != std::string_view::npos) {
        if (in_integer_block) parse_error("Cannot capture an int section while already capturing an int section");
        in_integer_block = true;
      }
      if (line.find("INTEND") != std::string_view::npos) {
        if (!in_integer_block) parse_error("Cannot stop int capture when a previous capture is not started");
        in_integer_block = false;
      }
      return;
    }
    std::string name(f[0]);
    if (model.var_names.empty() || model.var_names.back() != name) {
      if (var_id.count(name))
        parse_error("All rows for the column (" + name + ") should occur contiguously! line=" + std::string(line));
      var_id.emplace(name, (int)model.var_names.size());
      model.var_names.push_back(name);
      model.var_types.push_back(in_integer_block ? 'I' : 'C');
      model.c.push_back(0.0);
    }
    const int var = (int)model.var_names.size() - 1;
    for (size_t k = 1; k < f.size() && k <= 3; k += 2) {
      if (f[k][0] == '$') break;
      if (k + 1 >= f.size()) parse_error("Bad value found for row=" + std::string(f[k]) + " in COLUMNS! line=" + std::string(line));
      entry(var, f[k], f[k + 1], line);
    }
    if (f.size() < 3 && !(f.size() >= 2 && f[1][0] == '$'))
      parse_error("COLUMNS should have at least 3 entities! line=" + std::string(line));
  }

  void rhs_line(std::string_view line)
  {
    auto f = fields_of(line);
    if (f.empty()) return;
    size_t k = 1;  // skip the RHS set name unless the first field already is a row name
    if (std::string(f[0]) == model.objective_name || row_id.count(std::string(f[0]))) k = 0;
    for (int pair = 0; pair < 2 && k < f.size(); ++pair, k += 2) {
      if (f[k][0] == '$') break;
      std::string r(f[k]);
      if (k + 1 >= f.size()) parse_error("Bad value found for row=" + r + " in RHS! line=" + std::string(line));
      const double v = number(f[k + 1], "RHS", line);
      if (r == model.objective_name) {
        model.objective_offset = -v;  // RHS on the objective row is minus the constant term
      } else {
        auto it = row_id.find(r);
        if (it == row_id.end()) parse_error("Bad row name found '" + r + "' in RHS! line=" + std::string(line));
        model.rhs[it->second] = v;
      }
    }
  }

  void range_line(std::string_view line)
  {
    auto f = fields_of(line);
    size_t k = 1;  // the RANGES set name is mandatory in free format
    for (int pair = 0; pair < 2 && k < f.size(); ++pair, k += 2) {
      if (f[k][0] == '$') break;
      std::string r(f[k]);
      if (k + 1 >= f.size()) parse_error("Bad value found in RANGES! line=" + std::string(line));
      auto it = row_id.find(r);
      if (it == row_id.end()) parse_error("Bad row name found '" + r + "' in RANGES! line=" + std::string(line));
      range[it->second]       = number(f[k + 1], "RANGES", line);
      range_given[it->second] = true;
    }
  }

  void init_bounds()
  {
    if (bounds_initialised) return;
    bounds_initialised = true;
    model.lb.assign(model.var_names.size(), 0.0);
    model.ub.assign(model.var_names.size(), kInf);
  }

  void bound_line(std::string_view line)
  {
    auto f = fields_of(line);
    if (f.size() < 2) parse_error("BOUNDS should have at least 2 entities! line=" + std::string(line));
    std::string type(f[0]);
    // "TYPE set var [value]" or, when the set name is omitted, "TYPE var [value]"
    size_t var_field = 2;
    if (var_id.count(std::string(f[1])) || f.size() == 2) var_field = 1;
    if (var_field >= f.size()) parse_error("BOUNDS entry without a variable! line=" + std::string(line));
    if (f[var_field][0] == '$') return;
    std::string name(f[var_field]);
    auto it = var_id.find(name);
    if (it == var_id.end()) {  // a variable that never appeared in COLUMNS
      it = var_id.emplace(name, (int)model.var_names.size()).first;
      model.var_names.push_back(name);
      model.var_types.push_back('C');
      model.c.push_back(0.0);
      model.lb.push_back(0.0);
      model.ub.push_back(kInf);
    }
    const int j = it->second;
    auto value  = [&]() -> double {
      if (var_field + 1 >= f.size()) parse_error("Bad value found in BOUNDS! line=" + std::string(line));
      return number(f[var_field + 1], "BOUNDS", line);
    };
    const bool first = !bounded_vars.count(j);
    if (type == "LO") {
      model.lb[j] = value();
    } else if (type == "UP") {
      model.ub[j] = value();
      if (first && model.ub[j] < 0.0) model.lb[j] = -kInf;  // CPLEX convention
    } else if (type == "FX") {
      model.lb[j] = model.ub[j] = value();
    } else if (type == "FR") {
      model.lb[j] = -kInf, model.ub[j] = kInf;
    } else if (type == "MI") {
      model.lb[j] = -kInf;
    } else if (type == "PL") {
      model.ub[j] = kInf;
    } else if (type == "BV") {
      model.lb[j] = 0.0, model.ub[j] = 1.0, model.var_types[j] = 'I';
    } else if (type == "LI") {
      if (first) model.ub[j] = kInf;
      model.lb[j] = value(), model.var_types[j] = 'I';
    } else if (type == "UI") {
      model.ub[j] = value();
      if (first && model.ub[j] < 0.0) model.lb[j] = -kInf;
      model.var_types[j] = 'I';
    } else if (type == "LC" || type == "SC") {
      parse_error("Unsupported semi continuous bound type found! Line=" + std::string(line));
    } else {
      parse_error("Invalid variable bound type found in BOUNDS section! Bound type=" + type);
    }
    bounded_vars.insert(j);
  }

  void finish()
  {
    if (model.objective_name.empty()) parse_error("No objective found!");
    if (!sections_seen.count("ROWS")) parse_error("ROWS section is missing");
    if (!sections_seen.count("COLUMNS")) parse_error("COLUMNS section is missing");
    if (!sections_seen.count("RHS")) parse_error("RHS section is missing");
    init_bounds();
    const size_t n = model.var_names.size(), m = model.row_names.size();
    for (size_t j = 0; j < n; ++j) {
      if (!bounded_vars.count((int)j) && model.var_types[j] == 'I') model.lb[j] = 0.0, model.ub[j] = 1.0;
      if (!(model.lb[j] <= model.ub[j])) parse_error("Variable " + model.var_names[j] + " has lower bound > upper bound");
    }
    row_cols.resize(m), row_vals.resize(m), model.rhs.resize(m, 0.0);
    range.resize(m, 0.0), range_given.resize(m, false);
    model.offsets.assign(1, 0);
    for (size_t i = 0; i < m; ++i) {
      model.indices.insert(model.indices.end(), row_cols[i].begin(), row_cols[i].end());
      model.values.insert(model.values.end(), row_vals[i].begin(), row_vals[i].end());
      model.offsets.push_back((int)model.indices.size());
      // row type + RHS (+ RANGES) -> [lo, hi]
      double lo, hi;
      const double b = model.rhs[i], r = range[i];
      switch (model.row_types[i]) {
        case 'E':
          lo = hi = b;
          if (range_given[i]) (r < 0.0 ? lo : hi) += r;
          break;
        case 'G':
          lo = b, hi = range_given[i] ? b + std::fabs(r) : kInf;
          break;
        default:
          hi = b, lo = range_given[i] ? b - std::fabs(r) : -kInf;
          break;
      }
      if (std::isnan(lo) || std::isnan(hi)) parse_error("Constraint bound cannot be nan");
      model.lo.push_back(lo), model.hi.push_back(hi);
    }
  }
};

}  // namespace

MpsModel read_mps_file(const std::string& path)
{
  FILE* fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw MpsError(true, "Error opening MPS file! Given path: " + path);
  std::string text;
  char chunk[1 << 16];
  size_t got;
  while ((got = std::fread(chunk, 1, sizeof(chunk), fp)) > 0) text.append(chunk, got);
  std::fclose(fp);

  Builder b;
  Section section = Section::None;
  size_t pos = 0;
  bool any_line = false;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    std::string_view line(text.data() + pos, eol - pos);
    pos = eol + 1;
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.remove_suffix(1);
    if (line.empty() || line[0] == '*' || line[0] == '$') continue;
    any_line = true;
    if (line[0] != ' ' && line[0] != '\t') {  // section header
      auto f = fields_of(line);
      if (starts_with(line, "NAME")) {
        b.sections_seen.insert("NAME");
        if (f.size() > 1) b.model.problem_name = std::string(f[1]);
        section = Section::None;
      } else if (starts_with(line, "ROWS") || starts_with(line, "LAZYCONS")) {
        b.sections_seen.insert("ROWS");
        section = Section::Rows;
      } else if (starts_with(line, "COLUMNS")) {
        b.sections_seen.insert("COLUMNS");
        section = Section::Columns;
        b.row_cols.resize(b.model.row_names.size());
        b.row_vals.resize(b.model.row_names.size());
        b.model.rhs.assign(b.model.row_names.size(), 0.0);
      } else if (starts_with(line, "RHS")) {
        b.sections_seen.insert("RHS");
        section = Section::Rhs;
        b.model.rhs.resize(b.model.row_names.size(), 0.0);
      } else if (starts_with(line, "BOUNDS")) {
        b.sections_seen.insert("BOUNDS");
        section = Section::Bounds;
        b.init_bounds();
      } else if (starts_with(line, "RANGES")) {
        b.sections_seen.insert("RANGES");
        section = Section::Ranges;
        b.range.assign(b.model.row_names.size(), 0.0);
        b.range_given.assign(b.model.row_names.size(), false);
      } else if (starts_with(line, "OBJSENSE")) {
        if (f.size() > 1)
          b.objsense(f[1], line);
        else
          section = Section::ObjSense;
      } else if (starts_with(line, "OBJNAME")) {
        if (f.size() > 1)
          b.objname(f[1]);
        else
          section = Section::ObjName;
      } else if (starts_with(line, "ENDATA")) {
        b.sections_seen.insert("ENDATA");
        break;
      } else {
        parse_error("Invalid named block found! Line=" + std::string(line));
      }
      continue;
    }
    switch (section) {
      case Section::Rows: b.row_line(line); break;
      case Section::Columns: b.column_line(line); break;
      case Section::Rhs: b.rhs_line(line); break;
      case Section::Bounds: b.bound_line(line); break;
      case Section::Ranges: b.range_line(line); break;
      case Section::ObjSense: {
        auto f = fields_of(line);
        if (!f.empty()) b.objsense(f[0], line);
        break;
      }
      case Section::ObjName: {
        auto f = fields_of(line);
        if (!f.empty()) b.objname(f[0]);
        break;
      }
      default: parse_error("Ended up at a bad parser state! Line=" + std::string(line));
    }
  }
  if (!any_line) parse_error("Error parsing MPS file! The file is empty");
  b.finish();
  return std::move(b.model);
}

}  // namespace cuopt_amd
