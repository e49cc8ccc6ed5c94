// input.cpp — the script layer: the subset of the reference's input-script grammar that drives the
// hot path.  Command names, argument order, defaults and error strings follow
// src/input.cpp (file :181, one :327-355, execute_command :689-), src/read_data.cpp, src/atom.cpp
// (data_atoms / data_bonds :1235-1282), src/special.cpp, src/force.cpp (special_bonds :748-).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "engine.h"

#include <array>

namespace lmp_le {

double numeric(const std::string &s) {
  if (s.empty()) throw LammpsError("Expected floating point parameter instead of NULL or empty string");
  char *end = nullptr;
  double v = strtod(s.c_str(), &end);
  if (*end != '\0') throw LammpsError("Expected floating point parameter instead of '" + s + "' in input script or data file");
  return v;
}
int inumeric(const std::string &s) {
  if (s.empty()) throw LammpsError("Expected integer parameter instead of NULL or empty string");
  char *end = nullptr;
  long v = strtol(s.c_str(), &end, 10);
  if (*end != '\0') throw LammpsError("Expected integer parameter instead of '" + s + "' in input script or data file");
  return (int)v;
}
std::vector<std::string> split_words(const std::string &line) {
  std::vector<std::string> w;
  size_t i = 0, n = line.size();
  while (i < n) {
    while (i < n && isspace((unsigned char)line[i])) i++;
    if (i >= n) break;
    if (line[i] == '"' || line[i] == '\'') {
      char q = line[i++];
      size_t j = line.find(q, i);
      if (j == std::string::npos) throw LammpsError("Unbalanced quotes in input line");
      w.push_back(line.substr(i, j - i));
      i = j + 1;
    } else {
      size_t j = i;
      while (j < n && !isspace((unsigned char)line[j])) j++;
      w.push_back(line.substr(i, j - i));
      i = j;
    }
  }
  return w;
}

// ---------------------------------------------------------------------------------------------
// formulas (src/variable.cpp Variable::evaluate, the subset a bead-spring script uses): numbers, + - * / % ^, unary -,
// comparisons and && || !, parentheses, sqrt exp ln log abs sin cos tan floor ceil round min max, thermo keywords
// (step dt atoms bonds vol temp press pe ke etotal epair emol), v_name, f_ID[k]
// ---------------------------------------------------------------------------------------------
namespace {
struct Parser {
  Engine *e; const std::string &s; size_t i = 0;
  Parser(Engine *eng, const std::string &str) : e(eng), s(str) {}
  void ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
  bool eat(const char *tok) { ws(); size_t n = strlen(tok); if (s.compare(i, n, tok) == 0) { i += n; return true; } return false; }
  [[noreturn]] void bad() { throw LammpsError("Invalid syntax in variable formula: " + s); }
  double expr() { return lor(); }
  double lor() { double v = land(); while (eat("||")) { double r = land(); v = (v != 0.0 || r != 0.0) ? 1.0 : 0.0; } return v; }
  double land() { double v = cmp(); while (eat("&&")) { double r = cmp(); v = (v != 0.0 && r != 0.0) ? 1.0 : 0.0; } return v; }
  double cmp() {
    double v = add();
    for (;;) {
      if (eat("==")) v = (v == add()) ? 1.0 : 0.0;
      else if (eat("!=")) v = (v != add()) ? 1.0 : 0.0;
      else if (eat("<=")) v = (v <= add()) ? 1.0 : 0.0;
      else if (eat(">=")) v = (v >= add()) ? 1.0 : 0.0;
      else if (eat("<")) v = (v < add()) ? 1.0 : 0.0;
      else if (eat(">")) v = (v > add()) ? 1.0 : 0.0;
      else return v;
    }
  }
  double add() { double v = mul(); for (;;) { if (eat("+")) v += mul(); else if (eat("-")) v -= mul(); else return v; } }
  double mul() {
    double v = unary();
    for (;;) {
      if (eat("*")) v *= unary();
      else if (eat("/")) { double r = unary(); if (r == 0.0) throw LammpsError("Divide by 0 in variable formula"); v /= r; }
      else if (eat("%")) { double r = unary(); if (r == 0.0) throw LammpsError("Modulo 0 in variable formula"); v = fmod(v, r); }
      else return v;
    }
  }
  double unary() { if (eat("-")) return -unary(); if (eat("!")) return unary() == 0.0 ? 1.0 : 0.0; return power(); }
  double power() { double v = atom(); if (eat("^")) { double r = unary(); v = pow(v, r); } return v; }
  double atom() {
    ws();
    if (i >= s.size()) bad();
    if (s[i] == '(') { i++; double v = expr(); if (!eat(")")) bad(); return v; }
    if (isdigit((unsigned char)s[i]) || s[i] == '.') {
      char *end = nullptr;
      double v = strtod(s.c_str() + i, &end);
      if (end == s.c_str() + i) bad();
      i = end - s.c_str();
      return v;
    }
    size_t j = i;
    while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
    if (j == i) bad();
    std::string name = s.substr(i, j - i);
    i = j;
    if (name.rfind("v_", 0) == 0) return e->variable_value(name.substr(2));
    if (name.rfind("f_", 0) == 0) {
      int idx = 1;
      if (i < s.size() && s[i] == '[') { size_t k = s.find(']', i); if (k == std::string::npos) bad(); idx = atoi(s.c_str() + i + 1); i = k + 1; }
      Fix *f = e->find_fix(name.substr(2));
      if (!f) throw LammpsError("Invalid fix ID in variable formula");
      return f->compute_vector(idx - 1);
    }
    ws();
    if (i < s.size() && s[i] == '(') {   // function
      i++;
      double a = expr(), b = 0.0;
      bool two = eat(",");
      if (two) b = expr();
      if (!eat(")")) bad();
      if (name == "sqrt") { if (a < 0.0) throw LammpsError("Sqrt of negative value in variable formula"); return sqrt(a); }
      if (name == "exp") return exp(a);
      if (name == "ln") { if (a <= 0.0) throw LammpsError("Log of zero/negative value in variable formula"); return log(a); }
      if (name == "log") { if (a <= 0.0) throw LammpsError("Log of zero/negative value in variable formula"); return log10(a); }
      if (name == "abs") return fabs(a);
      if (name == "sin") return sin(a);
      if (name == "cos") return cos(a);
      if (name == "tan") return tan(a);
      if (name == "floor") return floor(a);
      if (name == "ceil") return ceil(a);
      if (name == "round") return (double)llround(a);
      if (name == "min" && two) return a < b ? a : b;
      if (name == "max" && two) return a > b ? a : b;
      throw LammpsError("Invalid math function in variable formula: " + name);
    }
    if (name == "step") return (double)e->ntimestep;
    {
      double val;
      bool isint;
      if (name.rfind("f_", 0) != 0 && e->thermo_keyword(e->last_thermo, name, val, isint)) return val;
    }
    if (name == "PI") return 3.14159265358979323846;
    throw LammpsError("Invalid thermo keyword in variable formula: " + name);
  }
};
std::string fmt_g15(double v) { char buf[64]; snprintf(buf, sizeof buf, "%.15g", v); return buf; }
}  // namespace

double Engine::evaluate(const std::string &expr) {
  Parser p(this, expr);
  double v = p.expr();
  p.ws();
  if (p.i != expr.size()) throw LammpsError("Invalid syntax in variable formula: " + expr);
  return v;
}
double Engine::variable_value(const std::string &name) {
  auto it = variables.find(name);
  if (it == variables.end()) throw LammpsError("Invalid variable name in variable formula: " + name);
  auto vi = var_info.find(name);
  if (vi != var_info.end() && vi->second.style == "equal") return evaluate(it->second);
  char *end = nullptr;
  double v = strtod(it->second.c_str(), &end);
  if (end == it->second.c_str() || *end) throw LammpsError("Variable " + name + " is not a number in variable formula");
  return v;
}

// ${name}, $x and $(formula[:format]) substitution (src/input.cpp:473-640 substitute)
std::string Engine::substitute(const std::string &line) {
  std::string out;
  char quote = 0;          // no replacement inside single / double quotes (src/input.cpp:513); print and if re-substitute
  for (size_t i = 0; i < line.size(); i++) {
    if (quote) { if (line[i] == quote) quote = 0; out += line[i]; continue; }
    if (line[i] == '"' || line[i] == '\'') { quote = line[i]; out += line[i]; continue; }
    if (line[i] != '$') { out += line[i]; continue; }
    std::string name;
    if (i + 1 < line.size() && line[i + 1] == '(') {            // immediate formula
      int depth = 0; size_t j = i + 1;
      for (; j < line.size(); j++) { if (line[j] == '(') depth++; else if (line[j] == ')' && --depth == 0) break; }
      if (j >= line.size()) throw LammpsError("Invalid immediate variable");
      std::string body = line.substr(i + 2, j - i - 2), f = "%.15g";
      size_t colon = body.rfind(':');
      if (colon != std::string::npos && body.find('%', colon) != std::string::npos) { f = body.substr(colon + 1); body = body.substr(0, colon); }
      char buf[128];
      snprintf(buf, sizeof buf, f.c_str(), evaluate(body));
      out += buf;
      i = j;
      continue;
    }
    if (i + 1 < line.size() && line[i + 1] == '{') {
      size_t j = line.find('}', i);
      if (j == std::string::npos) throw LammpsError("Invalid variable name");
      name = line.substr(i + 2, j - i - 2);
      i = j;
    } else if (i + 1 < line.size()) {
      name = line.substr(i + 1, 1);
      i++;
    }
    auto it = variables.find(name);
    if (it == variables.end()) throw LammpsError("Substitution for illegal variable " + name);
    auto vi = var_info.find(name);
    if (vi != var_info.end() && vi->second.style == "equal") out += fmt_g15(evaluate(it->second));
    else out += it->second;
  }
  return out;
}

// Input::file (src/input.cpp:181-290) incl. label / jump: `jump` re-opens a file (or SELF) and skips to the label
void Engine::file(const std::string &path_in) {
  std::string path = path_in, want_label;
  if (++file_depth > 16) { file_depth--; throw LammpsError("Too many nested levels of input scripts"); }
  struct Depth { int &d; ~Depth() { d--; } } guard{file_depth};
  for (;;) {
    std::ifstream in(path);
    if (!in) throw LammpsError("Cannot open input script " + path);
    std::vector<std::string> lines;
    std::string line, acc;
    while (std::getline(in, line)) {
      // '&' continuation (src/input.cpp:210-231)
      size_t e = line.find_last_not_of(" \t\r\n");
      if (e != std::string::npos && line[e] == '&') { acc += line.substr(0, e) + " "; continue; }
      acc += line;
      lines.push_back(acc);
      acc.clear();
    }
    if (!acc.empty()) lines.push_back(acc);
    bool jumped = false;
    for (size_t k = 0; k < lines.size(); k++) {
      if (!want_label.empty()) {            // skipping to a label: only `label X` lines are looked at
        std::vector<std::string> w = split_words(lines[k].substr(0, lines[k].find('#')));
        if (w.size() == 2 && w[0] == "label" && w[1] == want_label) want_label.clear();
        continue;
      }
      jump_pending = false;
      one(lines[k]);
      if (quit_requested) return;
      if (jump_pending) {
        jump_pending = false;
        if (jump_file != "SELF") path = jump_file;
        want_label = jump_label;
        jumped = true;
        break;
      }
    }
    if (!jumped) {
      if (!want_label.empty()) throw LammpsError("Label wasn't found in input script");
      return;
    }
  }
}

const char *Engine::one(const std::string &raw) {
  std::string line = raw;
  // strip comments outside quotes
  bool q1 = false, q2 = false;
  for (size_t i = 0; i < line.size(); i++) {
    if (line[i] == '"' && !q1) q2 = !q2;
    else if (line[i] == '\'' && !q2) q1 = !q1;
    else if (line[i] == '#' && !q1 && !q2) { line = line.substr(0, i); break; }
  }
  if (echo_screen) say(raw + "\n");
  line = substitute(line);
  std::vector<std::string> w = split_words(line);
  if (w.empty()) { last_cmd.clear(); return nullptr; }
  last_cmd = w[0];
  std::vector<std::string> arg(w.begin() + 1, w.end());
  execute(last_cmd, arg);
  return last_cmd.c_str();
}

static void set_units(Engine *e, const std::string &u) {
  // src/update.cpp:132-200
  if (u == "lj") {
    e->boltz = 1.0; e->mvv2e = 1.0; e->ftm2v = 1.0; e->nktv2p = 1.0; e->mv2d = 1.0; e->dt = 0.005; e->skin = 0.3;
    e->thermo_norm = true;
  } else if (u == "real") {
    e->boltz = 0.0019872067; e->mvv2e = 48.88821291 * 48.88821291; e->ftm2v = 1.0 / 48.88821291 / 48.88821291;
    e->nktv2p = 68568.415; e->mv2d = 1.0 / 0.602214129; e->dt = 1.0; e->skin = 2.0;
    e->thermo_norm = false;
  } else throw LammpsError("Illegal units command (MI355X engine supports lj and real)");
  e->units = u;
}

// expand "*", "n*", "*m", "n*m" type ranges (src/utils.cpp bounds)
static void bounds(const std::string &s, int nmax, int &lo, int &hi) {
  size_t star = s.find('*');
  if (star == std::string::npos) { lo = hi = inumeric(s); }
  else {
    lo = (star == 0) ? 1 : inumeric(s.substr(0, star));
    hi = (star + 1 == s.size()) ? nmax : inumeric(s.substr(star + 1));
  }
  if (lo < 1 || hi > nmax || lo > hi) throw LammpsError("Numeric index is out of bounds");
}

void Engine::execute(const std::string &cmd, std::vector<std::string> &arg) {
  auto need = [&](size_t n) { if (arg.size() < n) throw LammpsError("Illegal " + cmd + " command"); };
  if (cmd == "units") {
    need(1);
    if (box_exist) throw LammpsError("Units command after simulation box is defined");
    set_units(this, arg[0]);
  } else if (cmd == "atom_style") {
    need(1);
    if (box_exist) throw LammpsError("Atom_style command after simulation box is defined");
    if (arg[0] != "bond" && arg[0] != "molecular" && arg[0] != "atomic" && arg[0] != "angle" && arg[0] != "full")
      throw LammpsError("Unknown atom style " + arg[0]);
    atom_style = arg[0];
  } else if (cmd == "dimension") {
    need(1);
    if (inumeric(arg[0]) != 3) throw LammpsError("MI355X engine is 3d only");
  } else if (cmd == "boundary") {
    need(3);
    for (int k = 0; k < 3; k++) if (arg[k] != "p") throw LammpsError("MI355X engine supports boundary p p p only");
  } else if (cmd == "newton") {
    need(1);
    auto flag = [&](const std::string &s) {
      if (s == "on") return true;
      if (s == "off") return false;
      throw LammpsError("Illegal newton command");
    };
    if (arg.size() == 1) newton_pair = newton_bond = flag(arg[0]);
    else { newton_pair = flag(arg[0]); newton_bond = flag(arg[1]); }
  } else if (cmd == "atom_modify") {
    for (size_t i = 0; i < arg.size();) {
      if (arg[i] == "sort") {
        if (i + 3 > arg.size()) throw LammpsError("Illegal atom_modify command");
        sortfreq = inumeric(arg[i + 1]);
        if (sortfreq < 0) throw LammpsError("Illegal atom_modify command");
        i += 3;
      } else if (arg[i] == "map" || arg[i] == "first" || arg[i] == "id") {
        if (i + 2 > arg.size()) throw LammpsError("Illegal atom_modify command");
        i += 2;
      } else throw LammpsError("Illegal atom_modify command");
    }
  } else if (cmd == "special_bonds") {
    // src/force.cpp:748-826 set_special: every command starts from the defaults (lj 0 0 0, coul 0 0 0)
    need(1);
    const char *ill = "Illegal special_bonds command";
    for (int k = 1; k <= 3; k++) special_lj[k] = special_coul[k] = 0.0;
    auto set3 = [&](double *w, double a, double b, double c) { w[1] = a; w[2] = b; w[3] = c; };
    for (size_t i = 0; i < arg.size();) {
      if (arg[i] == "amber") { set3(special_lj, 0.0, 0.0, 0.5); set3(special_coul, 0.0, 0.0, 5.0 / 6.0); i++; }
      else if (arg[i] == "charmm") { set3(special_lj, 0.0, 0.0, 0.0); set3(special_coul, 0.0, 0.0, 0.0); i++; }
      else if (arg[i] == "dreiding") { set3(special_lj, 0.0, 0.0, 1.0); set3(special_coul, 0.0, 0.0, 1.0); i++; }
      else if (arg[i] == "fene") { set3(special_lj, 0.0, 1.0, 1.0); set3(special_coul, 0.0, 1.0, 1.0); i++; }
      else if (arg[i] == "lj/coul" || arg[i] == "lj" || arg[i] == "coul") {
        if (i + 4 > arg.size()) throw LammpsError(ill);
        for (int k = 1; k <= 3; k++) {
          const double w = numeric(arg[i + k]);
          if (arg[i] != "coul") special_lj[k] = w;
          if (arg[i] != "lj") special_coul[k] = w;
        }
        i += 4;
      } else if (arg[i] == "angle" || arg[i] == "dihedral") {
        if (i + 2 > arg.size() || (arg[i + 1] != "no" && arg[i + 1] != "yes")) throw LammpsError(ill);
        i += 2;
      } else throw LammpsError(ill);
    }
    for (int k = 1; k <= 3; k++)
      if (special_lj[k] < 0.0 || special_lj[k] > 1.0 || special_coul[k] < 0.0 || special_coul[k] > 1.0) throw LammpsError(ill);
    if (box_exist && natoms) { special_built = false; }
  } else if (cmd == "read_data") {
    need(1);
    read_data(arg[0]);
  } else if (cmd == "mass") {
    need(2);
    if (!box_exist) throw LammpsError("Mass command before simulation box is defined");
    int lo, hi;
    bounds(arg[0], ntypes, lo, hi);
    double m = numeric(arg[1]);
    if (m <= 0.0) throw LammpsError("Invalid mass value");
    for (int t = lo; t <= hi; t++) { mass[t] = m; mass_set[t] = 1; }
  } else if (cmd == "neighbor") {
    need(2);
    skin = numeric(arg[0]);
    if (skin < 0.0) throw LammpsError("Illegal neighbor command");
    if (arg[1] != "bin" && arg[1] != "nsq" && arg[1] != "multi") throw LammpsError("Illegal neighbor command");
  } else if (cmd == "neigh_modify") {
    for (size_t i = 0; i < arg.size();) {
      if (i + 2 > arg.size()) throw LammpsError("Illegal neigh_modify command");
      if (arg[i] == "every") { neigh_every = inumeric(arg[i + 1]); if (neigh_every <= 0) throw LammpsError("Illegal neigh_modify command"); }
      else if (arg[i] == "delay") { neigh_delay = inumeric(arg[i + 1]); if (neigh_delay < 0) throw LammpsError("Illegal neigh_modify command"); }
      else if (arg[i] == "check") {
        if (arg[i + 1] == "yes") neigh_check = 1; else if (arg[i + 1] == "no") neigh_check = 0;
        else throw LammpsError("Illegal neigh_modify command");
      } else if (arg[i] == "one" || arg[i] == "page" || arg[i] == "binsize" || arg[i] == "once" || arg[i] == "cluster") {}
      else throw LammpsError("Illegal neigh_modify command");
      i += 2;
    }
  } else if (cmd == "comm_modify") {
    for (size_t i = 0; i < arg.size();) {
      if (i + 2 > arg.size()) throw LammpsError("Illegal comm_modify command");
      if (arg[i] == "cutoff") comm_cutoff = numeric(arg[i + 1]);   // ghost cutoff: bonds use minimum image here
      else if (arg[i] == "mode" || arg[i] == "vel" || arg[i] == "group") {}
      else throw LammpsError("Illegal comm_modify command");
      i += 2;
    }
  } else if (cmd == "bond_style") {
    need(1);
    if (atom_style == "atomic") throw LammpsError("Bond_style command when no bonds allowed");
    bond_style_name = arg[0];
    bond_hybrid_styles.clear();
    for (int b = 0; b <= MAXTYPES; b++) bondtab.style[b] = 0;
    if (arg[0] == "hybrid") {
      for (size_t i = 1; i < arg.size(); i++) {
        if (arg[i] != "fene" && arg[i] != "harmonic" && arg[i] != "morse" && arg[i] != "zero") throw LammpsError("Unknown bond style " + arg[i]);
        bond_hybrid_styles.push_back(arg[i]);
      }
      if (bond_hybrid_styles.empty()) throw LammpsError("Illegal bond_style command");
    } else if (arg[0] != "fene" && arg[0] != "harmonic" && arg[0] != "morse" && arg[0] != "zero" && arg[0] != "none")
      throw LammpsError("Unknown bond style " + arg[0]);
  } else if (cmd == "bond_coeff") {
    need(1);
    if (!box_exist) throw LammpsError("Bond_coeff command before simulation box is defined");
    if (bond_style_name.empty() || bond_style_name == "none") throw LammpsError("Bond_coeff command before bond_style is defined");
    int lo, hi;
    bounds(arg[0], nbondtypes, lo, hi);
    std::string st = bond_style_name;
    size_t a0 = 1;
    if (st == "hybrid") {
      need(2);
      st = arg[1];
      if (std::find(bond_hybrid_styles.begin(), bond_hybrid_styles.end(), st) == bond_hybrid_styles.end())
        throw LammpsError("Bond coeff for hybrid has invalid style");
      a0 = 2;
    }
    for (int b = lo; b <= hi; b++) {
      if (st == "fene") {
        // K R0 epsilon sigma (src/MOLECULE/bond_fene.cpp:149-173)
        if (arg.size() != a0 + 4) throw LammpsError("Incorrect args for bond coefficients");
        bondtab.style[b] = 1;
        bondtab.p0[b] = numeric(arg[a0]); bondtab.p1[b] = numeric(arg[a0 + 1]);
        bondtab.p2[b] = numeric(arg[a0 + 2]); bondtab.p3[b] = numeric(arg[a0 + 3]);
      } else if (st == "harmonic") {
        // K r0 (src/MOLECULE/bond_harmonic.cpp:121-141)
        if (arg.size() != a0 + 2) throw LammpsError("Incorrect args for bond coefficients");
        bondtab.style[b] = 2;
        bondtab.p0[b] = numeric(arg[a0]); bondtab.p1[b] = numeric(arg[a0 + 1]);
      } else if (st == "morse") {
        // D alpha r0 (src/MOLECULE/bond_morse.cpp:130-152); evaluated by the unfused force kernel only (Engine::iterate)
        if (arg.size() != a0 + 3) throw LammpsError("Incorrect args for bond coefficients");
        bondtab.style[b] = 3;
        bondtab.p0[b] = numeric(arg[a0]); bondtab.p1[b] = numeric(arg[a0 + 1]); bondtab.p2[b] = numeric(arg[a0 + 2]);
      } else bondtab.style[b] = 0;
    }
  } else if (cmd == "angle_style") {
    // src/MOLECULE/angle_harmonic.cpp, angle_cosine.cpp (semiflexible chains); other styles are out of scope
    need(1);
    if (atom_style == "atomic" || atom_style == "bond") throw LammpsError("Angle_style command when no angles allowed");
    if (arg[0] != "harmonic" && arg[0] != "cosine" && arg[0] != "zero" && arg[0] != "none")
      throw LammpsError("Unknown angle style " + arg[0]);
    angle_style_name = arg[0];
    for (int a = 0; a <= MAXTYPES; a++) angtab.style[a] = 0;
  } else if (cmd == "angle_coeff") {
    need(1);
    if (!box_exist) throw LammpsError("Angle_coeff command before simulation box is defined");
    if (angle_style_name.empty() || angle_style_name == "none") throw LammpsError("Angle_coeff command before angle_style is defined");
    int lo, hi;
    bounds(arg[0], nangletypes, lo, hi);
    for (int a = lo; a <= hi; a++) {
      if (angle_style_name == "harmonic") {        // K theta0[degrees] (angle_harmonic.cpp:170-196: stored in radians)
        if (arg.size() != 3) throw LammpsError("Incorrect args for angle coefficients");
        angtab.style[a] = 1; angtab.k[a] = numeric(arg[1]); angtab.theta0[a] = numeric(arg[2]) / 180.0 * 3.14159265358979323846;
      } else if (angle_style_name == "cosine") {   // K (angle_cosine.cpp:124-146)
        if (arg.size() != 2) throw LammpsError("Incorrect args for angle coefficients");
        angtab.style[a] = 2; angtab.k[a] = numeric(arg[1]); angtab.theta0[a] = 0.0;
      } else angtab.style[a] = 0;
    }
  } else if (cmd == "pair_style") {
    need(1);
    pair_lj = pair_zero = false;
    if (arg[0] == "lj/cut") {
      need(2);
      pair_lj = true;
      pair_cut_global = numeric(arg[1]);
    } else if (arg[0] == "zero") {
      need(2);
      pair_zero = true;
      pair_cut_global = numeric(arg[1]);
    } else if (arg[0] != "none") throw LammpsError("Unknown pair style " + arg[0]);
    if (box_exist) {
      int nt = ntypes + 1;
      pc_eps.assign(nt * nt, 0.0); pc_sig = pc_cut = pc_eps; pc_set.assign(nt * nt, 0);
    }
  } else if (cmd == "pair_modify") {
    for (size_t i = 0; i < arg.size();) {
      if (i + 2 > arg.size()) throw LammpsError("Illegal pair_modify command");
      if (arg[i] == "shift") {
        if (arg[i + 1] == "yes") pair_shift = true; else if (arg[i + 1] == "no") pair_shift = false;
        else throw LammpsError("Illegal pair_modify command");
      } else if (arg[i] == "mix") {
        if (arg[i + 1] == "geometric") pair_mix = 0; else if (arg[i + 1] == "arithmetic") pair_mix = 1;
        else throw LammpsError("Illegal pair_modify command (MI355X engine: mix geometric|arithmetic)");
      } else if (arg[i] == "tail") { if (arg[i + 1] != "no") throw LammpsError("MI355X engine: pair_modify tail yes not supported"); }
      else throw LammpsError("Illegal pair_modify command");
      i += 2;
    }
  } else if (cmd == "pair_coeff") {
    // i j epsilon sigma [cut]  (src/pair_lj_cut.cpp:446-474)
    if (!box_exist) throw LammpsError("Pair_coeff command before simulation box is defined");
    if (!pair_lj && !pair_zero) throw LammpsError("Pair_coeff command before pair_style is defined");
    if (pair_zero) return;
    if (arg.size() < 4 || arg.size() > 5) throw LammpsError("Incorrect args for pair coefficients");
    int ilo, ihi, jlo, jhi;
    bounds(arg[0], ntypes, ilo, ihi);
    bounds(arg[1], ntypes, jlo, jhi);
    double eps = numeric(arg[2]), sig = numeric(arg[3]);
    double cut = (arg.size() == 5) ? numeric(arg[4]) : pair_cut_global;
    int nt = ntypes + 1, count = 0;
    if ((int)pc_set.size() != nt * nt) { pc_eps.assign(nt * nt, 0.0); pc_sig = pc_cut = pc_eps; pc_set.assign(nt * nt, 0); }
    for (int i = ilo; i <= ihi; i++)
      for (int j = std::max(jlo, i); j <= jhi; j++) {
        pc_eps[i * nt + j] = eps; pc_sig[i * nt + j] = sig; pc_cut[i * nt + j] = cut; pc_set[i * nt + j] = 1;
        count++;
      }
    if (count == 0) throw LammpsError("Incorrect args for pair coefficients");
  } else if (cmd == "fix") {
    // src/modify.cpp:789-947 add_fix: ID group style args
    if (arg.size() < 3) throw LammpsError("Illegal fix command");
    if (!box_exist) throw LammpsError("Fix command before simulation box is defined");
    if (find_fix(arg[0])) throw LammpsError("MI355X engine: replacing an existing fix ID is not supported: " + arg[0]);
    std::unique_ptr<Fix> f;
    const std::string &st = arg[2];
    if (st == "nve") f.reset(new FixNVE(this, arg));
    else if (st == "langevin") f.reset(new FixLangevin(this, arg));
    else if (st == "extrusion") f.reset(new FixExtrusion(this, arg));
    else if (st == "ex_load" || st == "bond/create") f.reset(new FixExLoad(this, arg));
    else if (st == "ex_unload" || st == "bond/break") f.reset(new FixExUnload(this, arg));
    else throw LammpsError("Unknown fix style " + st);
    apply_restart_state(f.get());     // a fix re-specified after read_restart continues its RNG stream (Fix::restart)
    fixes.push_back(std::move(f));
  } else if (cmd == "unfix") {
    need(1);
    auto it = std::find_if(fixes.begin(), fixes.end(), [&](const std::unique_ptr<Fix> &f) { return f->id == arg[0]; });
    if (it == fixes.end()) throw LammpsError("Could not find fix ID to delete");
    fixes.erase(it);
  } else if (cmd == "thermo") {
    need(1);
    thermo_every = inumeric(arg[0]);
    if (thermo_every < 0) throw LammpsError("Illegal thermo command");
  } else if (cmd == "thermo_style") {
    need(1);
    thermo_multi = false;
    if (arg[0] == "one") thermo_keywords = {"step", "temp", "epair", "emol", "etotal", "press"};
    else if (arg[0] == "multi") {                                  // src/thermo.cpp:70, 115-119
      thermo_keywords = {"etotal", "ke", "temp", "pe", "ebond", "eangle", "edihed", "eimp", "evdwl", "ecoul", "elong", "press"};
      thermo_multi = true;
    } else if (arg[0] == "custom") {
      if (arg.size() < 2) throw LammpsError("Illegal thermo style custom command");
      for (size_t k = 1; k < arg.size(); k++) {                    // (unknown keywords stop here: src/thermo.cpp:884-1040)
        double val;
        bool isint;
        if (arg[k].rfind("f_", 0) != 0 && !thermo_keyword(last_thermo, arg[k], val, isint))
          throw LammpsError("Unknown keyword in thermo_style custom command: " + arg[k]);
      }
      thermo_keywords.assign(arg.begin() + 1, arg.end());
    } else throw LammpsError("Illegal thermo_style command");
  } else if (cmd == "thermo_modify") {
    for (size_t i = 0; i < arg.size();) {
      if (i + 2 > arg.size()) throw LammpsError("Illegal thermo_modify command");
      if (arg[i] == "norm") thermo_norm = (arg[i + 1] == "yes");
      else if (arg[i] == "line") {                                 // src/thermo.cpp:551-556
        if (arg[i + 1] == "one") thermo_multi = false;
        else if (arg[i + 1] == "multi") thermo_multi = true;
        else throw LammpsError("Illegal thermo_modify command");
      }
      else if (arg[i] == "format") { i += 3; continue; }
      i += 2;
    }
  } else if (cmd == "timestep") {
    need(1);
    dt = numeric(arg[0]);
  } else if (cmd == "reset_timestep") {
    need(1);
    ntimestep = atol(arg[0].c_str());
    if (ntimestep < 0) throw LammpsError("Timestep must be >= 0");      // src/update.cpp:474-482
    atimestep = ntimestep;
  } else if (cmd == "run") {
    need(1);
    long nrun = atol(arg[0].c_str());
    for (size_t k = 1; k < arg.size(); k++) {
      if (arg[k] == "upto") { nrun -= ntimestep; if (nrun < 0) throw LammpsError("Run command upto value is before current timestep"); }
      else throw LammpsError("MI355X engine: run keyword " + arg[k] + " is not supported");
    }
    run(nrun);
  } else if (cmd == "write_data") {
    need(1);
    write_data(arg[0]);
  } else if (cmd == "velocity") {
    velocity(arg);
  } else if (cmd == "set") {
    set_command(arg);
  } else if (cmd == "write_restart") {
    need(1);
    write_restart(arg[0]);
  } else if (cmd == "restart") {
    need(1);
    restart_every = atol(arg[0].c_str());
    if (restart_every < 0) throw LammpsError("Illegal restart command");
    restart_a.clear(); restart_b.clear(); restart_toggle = 0;
    if (restart_every > 0) {
      if (arg.size() != 2 && arg.size() != 3) throw LammpsError("Illegal restart command");
      restart_a = arg[1];
      if (arg.size() == 3) restart_b = arg[2];
    }
  } else if (cmd == "read_restart") {
    need(1);
    if (box_exist) throw LammpsError("Cannot read_restart after simulation box is defined");   // src/read_restart.cpp:60
    read_restart(arg[0]);
  } else if (cmd == "compute") {
    // compute ID group property/local attr...   (src/compute_property_local.cpp:30-180; bond attributes only)
    need(4);
    const int cbit = group_bit(arg[1]);
    if (!cbit) throw LammpsError("Could not find compute group ID");         // src/compute.cpp:63-64
    if (arg[2] != "property/local") throw LammpsError("Unknown compute style " + arg[2]);
    std::vector<std::string> attrs(arg.begin() + 3, arg.end());
    for (auto &a : attrs)
      if (a != "btype" && a != "batom1" && a != "batom2")
        throw LammpsError("MI355X engine: compute property/local supports btype batom1 batom2 (got " + a + ")");
    computes_local[arg[0]] = attrs;
    computes_local_bit[arg[0]] = cbit;
  } else if (cmd == "uncompute") {
    need(1);
    if (!computes_local.erase(arg[0])) throw LammpsError("Could not find uncompute ID");
    computes_local_bit.erase(arg[0]);
  } else if (cmd == "dump") {
    // dump ID group style N file args   (src/dump.cpp:60-170, dump_custom.cpp:60-250, dump_local.cpp:40-130)
    need(5);
    const int dbit = group_bit(arg[1]);
    if (!dbit) throw LammpsError("Could not find dump group ID");            // src/dump.cpp:69-70
    for (auto &dp : dumps) if (dp.id == arg[0]) throw LammpsError("Reuse of dump ID");
    Dump dp;
    dp.groupbit = dbit;
    dp.id = arg[0]; dp.style = arg[2]; dp.every = atol(arg[3].c_str()); dp.path = arg[4];
    if (dp.every <= 0) throw LammpsError("Invalid dump frequency");
    static const char *atom_cols[] = {"id", "mol", "type", "mass", "x", "y", "z", "xs", "ys", "zs", "xu", "yu", "zu", "ix", "iy",
                                      "iz", "vx", "vy", "vz", "fx", "fy", "fz"};
    if (dp.style == "atom") {
      if (arg.size() != 5) throw LammpsError("Illegal dump atom command");
      dp.cols = {"id", "type", "xs", "ys", "zs"};
    } else if (dp.style == "custom") {
      if (arg.size() < 6) throw LammpsError("Illegal dump custom command");
      for (size_t k = 5; k < arg.size(); k++) {
        bool ok = false;
        for (auto c : atom_cols) ok = ok || arg[k] == c;
        if (!ok) throw LammpsError("MI355X engine: dump custom attribute " + arg[k] + " is not supported");
        dp.cols.push_back(arg[k]);
      }
    } else if (dp.style == "local") {
      if (arg.size() < 6) throw LammpsError("Illegal dump local command");
      for (size_t k = 5; k < arg.size(); k++) {
        const std::string &a = arg[k];
        if (a != "index") {
          size_t lb = a.find('['), rb = a.find(']');
          if (a.compare(0, 2, "c_") != 0 || lb == std::string::npos || rb == std::string::npos)
            throw LammpsError("Invalid attribute in dump local command");
          std::string cid = a.substr(2, lb - 2);
          if (!computes_local.count(cid)) throw LammpsError("Could not find dump local compute ID");
          int col = atoi(a.substr(lb + 1, rb - lb - 1).c_str());
          if (col < 1 || col > (int)computes_local[cid].size()) throw LammpsError("Dump local compute vector is accessed out-of-range");
        }
        dp.cols.push_back(a);
      }
    } else if (dp.style == "dcd") {
      // src/dump_dcd.cpp:60-95: no extra arguments, one file for all snapshots, atoms in ID order
      if (arg.size() != 5) throw LammpsError("Illegal dump dcd command");
      if (dp.path.find('*') != std::string::npos || dp.path.find('%') != std::string::npos)
        throw LammpsError("Invalid dump dcd filename");
    } else throw LammpsError("Unknown dump style " + dp.style);
    dumps.push_back(dp);
  } else if (cmd == "dump_modify") {
    need(1);
    Dump *dp = nullptr;
    for (auto &q : dumps) if (q.id == arg[0]) dp = &q;
    if (!dp) throw LammpsError("Cound not find dump_modify ID");      // (sic) src/output.cpp:708
    for (size_t k = 1; k < arg.size(); k += 2) {
      if (k + 1 >= arg.size()) throw LammpsError("Illegal dump_modify command");
      if (arg[k] == "every") { dp->every = atol(arg[k + 1].c_str()); if (dp->every <= 0) throw LammpsError("Illegal dump_modify command"); }
      else if (arg[k] == "sort") { if (arg[k + 1] != "id" && arg[k + 1] != "off") throw LammpsError("MI355X engine: dump_modify sort id|off only (rows are always written in ID order)"); }
      else if (arg[k] == "label") dp->label = arg[k + 1];
      else if (arg[k] == "unwrap") {                                             // src/dump_dcd.cpp:262-272
        if (dp->style != "dcd") throw LammpsError("Illegal dump_modify command");
        if (arg[k + 1] == "yes") dp->unwrap = true; else if (arg[k + 1] == "no") dp->unwrap = false;
        else throw LammpsError("Illegal dump_modify command");
      }
      else throw LammpsError("MI355X engine: dump_modify " + arg[k] + " is not supported");
    }
  } else if (cmd == "undump") {
    need(1);
    bool found = false;
    for (size_t k = 0; k < dumps.size(); k++)
      if (dumps[k].id == arg[0]) { if (dumps[k].fp) fclose(dumps[k].fp); dumps.erase(dumps.begin() + k); found = true; break; }
    if (!found) throw LammpsError("Could not find undump ID");
  } else if (cmd == "variable") {
    // variable name style args   (src/variable.cpp:82-520: index, loop, string, equal, delete)
    need(2);
    const std::string &name = arg[0], &st = arg[1];
    if (st == "delete") { variables.erase(name); var_info.erase(name); return; }
    need(3);
    if (st == "index") {
      if (variables.count(name)) return;                          // already defined (incl. command-line -var): ignored
      VarInfo vi; vi.style = st; vi.values.assign(arg.begin() + 2, arg.end());
      variables[name] = vi.values[0]; var_info[name] = vi;
    } else if (st == "loop") {
      if (variables.count(name)) return;
      // loop N | loop N pad | loop N1 N2 | loop N1 N2 pad
      long n1 = 1, n2 = 0; bool pad = false;
      std::vector<std::string> a(arg.begin() + 2, arg.end());
      if (!a.empty() && a.back() == "pad") { pad = true; a.pop_back(); }
      if (a.size() == 1) n2 = atol(a[0].c_str());
      else if (a.size() == 2) { n1 = atol(a[0].c_str()); n2 = atol(a[1].c_str()); }
      else throw LammpsError("Illegal variable command");
      if (n2 < n1 || n1 < 0) throw LammpsError("Illegal variable command");
      VarInfo vi; vi.style = st;
      size_t width = std::to_string(n2).size();
      for (long v = n1; v <= n2; v++) {
        std::string sv = std::to_string(v);
        if (pad && sv.size() < width) sv = std::string(width - sv.size(), '0') + sv;
        vi.values.push_back(sv);
      }
      variables[name] = vi.values[0]; var_info[name] = vi;
    } else if (st == "string") {
      VarInfo vi; vi.style = st; vi.values = {arg[2]};
      variables[name] = arg[2]; var_info[name] = vi;
    } else if (st == "equal") {
      std::string f;
      for (size_t k = 2; k < arg.size(); k++) f += arg[k];
      VarInfo vi; vi.style = st; vi.values = {f};
      variables[name] = f; var_info[name] = vi;
    } else throw LammpsError("MI355X engine: variable style " + st + " not supported");
  } else if (cmd == "next") {
    // src/variable.cpp:560-700: advance every listed variable; an exhausted one is deleted and the next jump is skipped
    need(1);
    for (auto &name : arg) {
      auto vi = var_info.find(name);
      if (vi == var_info.end()) {
        if (!variables.count(name)) throw LammpsError("Invalid variable in next command");
        variables.erase(name); jump_skip = true;                 // a -var command-line value has no successor
        continue;
      }
      if (vi->second.style != "index" && vi->second.style != "loop") throw LammpsError("Invalid variable style with next command");
      if (++vi->second.which >= vi->second.values.size()) { variables.erase(name); var_info.erase(vi); jump_skip = true; }
      else variables[name] = vi->second.values[vi->second.which];
    }
  } else if (cmd == "run_style") {
    need(1);
    if (arg[0] == "verlet") respa_levels = 0;
    else if (arg[0] == "respa") {
      // Respa::Respa (src/respa.cpp:47-262): N, N-1 loop factors, then keyword/level pairs (levels 1-based in the script)
      const std::string ill = "Illegal run_style respa command";
      std::vector<std::string> a(arg.begin() + 1, arg.end());
      if (a.empty()) throw LammpsError(ill);
      const int nl = inumeric(a[0]);
      if (nl < 1) throw LammpsError("Respa levels must be >= 1");
      if (nl > 8) throw LammpsError("MI355X engine: at most 8 respa levels");
      if ((int)a.size() < nl) throw LammpsError(ill);
      int loop[8] = {1, 1, 1, 1, 1, 1, 1, 1};
      for (int k = 1; k < nl; k++) { loop[k - 1] = inumeric(a[k]); if (loop[k - 1] <= 0) throw LammpsError(ill); }
      int lb = -1, la = -1, ld = -1, li = -1, lp = -1, lk = -1;
      for (size_t k = (size_t)nl; k < a.size(); k += 2) {
        if (k + 2 > a.size()) throw LammpsError(ill);
        const int lev = inumeric(a[k + 1]) - 1;
        if (a[k] == "bond") lb = lev; else if (a[k] == "angle") la = lev; else if (a[k] == "dihedral") ld = lev;
        else if (a[k] == "improper") li = lev; else if (a[k] == "pair") lp = lev; else if (a[k] == "kspace") lk = lev;
        else if (a[k] == "inner" || a[k] == "middle" || a[k] == "outer" || a[k] == "hybrid")
          throw LammpsError("MI355X engine: run_style respa " + a[k] + " (split pair forces) is not supported");
        else throw LammpsError(ill);
      }
      if (lb == -1) lb = 0;                    // :169-177 defaults
      if (la == -1) la = lb;
      if (ld == -1) ld = la;
      if (li == -1) li = ld;
      if (lp == -1) lp = nl - 1;
      if (lk == -1) lk = lp;
      if (la < lb || ld < la || li < ld || lp < li || lk < lp || lb < 0 || lk >= nl)          // :218-224
        throw LammpsError("Invalid order of forces within respa levels");
      respa_levels = nl;
      for (int k = 0; k < 8; k++) respa_loop[k] = loop[k];
      respa_loop[nl - 1] = 1;
      respa_level_bond = lb;
      respa_level_pair = lp;
      respa_level_angle = la;
      std::string msg = "Respa levels:\n";     // :198-214
      for (int k = 0; k < nl; k++) {
        msg += "  " + std::to_string(k + 1) + " =";
        if (lb == k) msg += " bond";
        if (la == k) msg += " angle";
        if (ld == k) msg += " dihedral";
        if (li == k) msg += " improper";
        if (lp == k) msg += " pair";
        if (lk == k) msg += " kspace";
        msg += "\n";
      }
      say(msg);
    } else throw LammpsError("MI355X engine: run_style " + arg[0] + " is not supported (verlet, respa)");
  } else if (cmd == "label") {
    need(1);
  } else if (cmd == "jump") {
    need(1);
    if (jump_skip) { jump_skip = false; return; }
    if (file_depth == 0) throw LammpsError("MI355X engine: jump is only available inside an input script");
    jump_file = arg[0];
    jump_label = arg.size() > 1 ? arg[1] : "";
    jump_pending = true;
  } else if (cmd == "include") {
    need(1);
    file(arg[0]);
  } else if (cmd == "if") {
    // if boolean then t1 t2 ... elif boolean f1 ... else e1 ...   (src/input.cpp:851-986)
    need(3);
    if (arg[1] != "then") throw LammpsError("Illegal if command");
    auto truth = [&](const std::string &b0) {
      const std::string b = substitute(b0);                       // src/input.cpp:842, :901 (the Boolean may be quoted)
      try { return evaluate(b) != 0.0; }
      catch (LammpsError &) {
        for (const char *op : {"==", "!="}) {                    // string comparison
          size_t k = b.find(op);
          if (k == std::string::npos) continue;
          auto trim = [](std::string t) { size_t x = t.find_first_not_of(" \t"), y = t.find_last_not_of(" \t"); return x == std::string::npos ? std::string() : t.substr(x, y - x + 1); };
          bool eq = trim(b.substr(0, k)) == trim(b.substr(k + 2));
          return (op[0] == '=') ? eq : !eq;
        }
        throw;
      }
    };
    std::vector<std::string> todo;
    bool done = false, cond = truth(arg[0]);
    size_t k = 2;
    for (;;) {
      size_t first = k;
      while (k < arg.size() && arg[k] != "elif" && arg[k] != "else") k++;
      if (first == k) throw LammpsError("Illegal if command");
      if (cond && !done) { todo.assign(arg.begin() + first, arg.begin() + k); done = true; }
      if (k == arg.size()) break;
      if (arg[k] == "elif") {
        if (k + 1 >= arg.size()) throw LammpsError("Illegal if command");
        cond = done ? false : truth(arg[k + 1]);
        k += 2;
      } else { cond = true; k++; }
    }
    for (auto &c : todo) { one(c); if (jump_pending || quit_requested) break; }
  } else if (cmd == "quit") {
    quit_requested = true;
  } else if (cmd == "log") {
    need(1);
    if (logfile) { fclose(logfile); logfile = nullptr; }
    if (arg[0] != "none") {
      logfile = fopen(arg[0].c_str(), (arg.size() > 1 && arg[1] == "append") ? "a" : "w");
      if (!logfile) throw LammpsError("Cannot open logfile " + arg[0]);
    }
  } else if (cmd == "echo") {
    need(1);
    echo_screen = (arg[0] == "screen" || arg[0] == "both");
  } else if (cmd == "print") {
    need(1);
    say(substitute(arg[0]) + "\n");                              // src/input.cpp:1093
  } else if (cmd == "timer") {
    // src/timer.cpp:230-300 modify_params: off | loop | normal | full | sync | nosync (timeout / every: no wall-time limit here)
    for (size_t k = 0; k < arg.size(); k++) {
      const std::string &a = arg[k];
      if (a == "off") timer_level = 0;
      else if (a == "loop") timer_level = 1;
      else if (a == "normal") timer_level = 2;
      else if (a == "full") timer_level = 3;
      else if (a == "sync") timer_sync = true;
      else if (a == "nosync") timer_sync = false;
      else if ((a == "timeout" || a == "every") && k + 1 < arg.size()) k++;
      else throw LammpsError("Illegal timer command");
    }
  } else if (cmd == "processors" || cmd == "package" ||
             cmd == "suffix" || cmd == "velocity_zero") {
  } else if (cmd == "group") {
    group_command(arg);
  } else if (cmd == "region") {
    region_command(arg);
  } else if (cmd == "clear") {
    throw LammpsError("MI355X engine: clear is not supported; open a new instance");
  } else {
    throw LammpsError("Unknown command: " + cmd);   // src/input.cpp:352-353
  }
}

// ---------------------------------------------------------------------------------------------
// read_data (src/read_data.cpp): header keywords incl. "extra bond per atom" (:1112) and
// "extra special per atom" (:1124); sections Masses, Atoms (bond/molecular: id mol type x y z [ix iy iz];
// full: id mol type q x y z [..]), Velocities, Bonds; bond coefficient sections are skipped.
// ---------------------------------------------------------------------------------------------
void Engine::read_data(const std::string &path) {
  if (box_exist) throw LammpsError("MI355X engine: read_data can be used once per instance");
  std::ifstream in(path);
  if (!in) throw LammpsError("Cannot open file " + path);
  std::string line;
  std::getline(in, line);   // title
  long nb_hdr = 0, nang_hdr = 0;
  int na = 0;
  bool have[3] = {false, false, false};
  static const char *sections[] = {"Masses", "Atoms", "Velocities", "Bonds", "Angles", "Dihedrals", "Impropers",
                                   "Pair Coeffs", "PairIJ Coeffs", "Bond Coeffs", "Angle Coeffs", "Dihedral Coeffs",
                                   "Improper Coeffs", "BondBond Coeffs", "BondAngle Coeffs"};
  auto strip = [](std::string s) {
    size_t h = s.find('#');
    if (h != std::string::npos) s = s.substr(0, h);
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return std::string();
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
  };
  auto is_section = [&](const std::string &s) -> const char * {
    for (const char *sec : sections) if (s == sec) return sec;
    return nullptr;
  };
  std::string section;
  // header
  while (std::getline(in, line)) {
    std::string s = strip(line);
    if (s.empty()) continue;
    if (is_section(s)) { section = s; break; }
    std::vector<std::string> w = split_words(s);
    auto ends = [&](const char *suffix) {
      std::string suf = suffix;
      return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
    };
    if (ends("atoms")) na = inumeric(w[0]);
    else if (ends("bonds")) nb_hdr = atol(w[0].c_str());
    else if (ends("angles")) nang_hdr = atol(w[0].c_str());
    else if (ends("dihedrals") || ends("impropers")) {}
    else if (ends("atom types")) ntypes = inumeric(w[0]);
    else if (ends("bond types")) nbondtypes = inumeric(w[0]);
    else if (ends("angle types")) nangletypes = inumeric(w[0]);
    else if (ends("dihedral types") || ends("improper types")) {}
    else if (ends("extra bond per atom")) extra_bond = inumeric(w[0]);
    else if (ends("extra special per atom")) extra_special = inumeric(w[0]);
    else if (ends("extra angle per atom")) extra_angle = inumeric(w[0]);
    else if (ends("extra dihedral per atom") || ends("extra improper per atom")) {}
    else if (ends("xlo xhi")) { box.lo[0] = numeric(w[0]); box.hi[0] = numeric(w[1]); have[0] = true; }
    else if (ends("ylo yhi")) { box.lo[1] = numeric(w[0]); box.hi[1] = numeric(w[1]); have[1] = true; }
    else if (ends("zlo zhi")) { box.lo[2] = numeric(w[0]); box.hi[2] = numeric(w[1]); have[2] = true; }
    else if (ends("xy xz yz")) throw LammpsError("MI355X engine: triclinic boxes are not supported");
    else throw LammpsError("Unknown identifier in data file: " + s);
  }
  if (!have[0] || !have[1] || !have[2]) throw LammpsError("Box bounds are missing in data file");
  for (int d = 0; d < 3; d++) {
    if (box.lo[d] >= box.hi[d]) throw LammpsError("Box bounds are invalid or missing");
    box.prd[d] = box.hi[d] - box.lo[d];
    box.half[d] = 0.5 * box.prd[d];
    box.iprd[d] = 1.0 / box.prd[d];
  }
  natoms = na;
  box_exist = true;
  mass.assign(ntypes + 1, 0.0);
  mass_set.assign(ntypes + 1, 0);
  x.assign(3 * (size_t)natoms, 0.0); v = f = x;
  type.assign(natoms, 0); molecule.assign(natoms, 0); image.assign(3 * (size_t)natoms, 0);
  crank.resize(natoms);
  for (int i = 0; i < natoms; i++) crank[i] = -1;
  std::vector<int> b_t, b_1, b_2, g_t, g_1, g_2, g_3;
  bool atoms_read = false;
  int file_rank = 0;
  bool full = (atom_style == "full"), has_mol = (atom_style != "atomic");
  // sections
  while (!section.empty()) {
    std::string cur = section;
    section.clear();
    while (std::getline(in, line)) {
      std::string s = strip(line);
      if (s.empty()) continue;
      if (is_section(s)) { section = s; break; }
      std::vector<std::string> w = split_words(s);
      if (cur == "Masses") {
        if (w.size() < 2) throw LammpsError("Invalid mass line in data file");
        int t = inumeric(w[0]);
        if (t < 1 || t > ntypes) throw LammpsError("Invalid type for mass set");
        mass[t] = numeric(w[1]);
        if (mass[t] <= 0.0) throw LammpsError("Invalid mass value");
        mass_set[t] = 1;
      } else if (cur == "Atoms") {
        size_t base = 1 + (has_mol ? 1 : 0) + 1 + (full ? 1 : 0);   // id [mol] type [q]
        if (w.size() != base + 3 && w.size() != base + 6) throw LammpsError("Incorrect atom format in data file");
        int id = inumeric(w[0]);
        if (id < 1 || id > natoms) throw LammpsError("MI355X engine: atom IDs must be 1..natoms");
        int i = id - 1;
        if (crank[i] >= 0) throw LammpsError("Duplicate atom IDs exist");
        if (has_mol) molecule[i] = inumeric(w[1]);
        type[i] = inumeric(w[has_mol ? 2 : 1]);
        if (type[i] < 1 || type[i] > ntypes) throw LammpsError("Invalid atom type in Atoms section of data file");
        for (int d = 0; d < 3; d++) x[3 * i + d] = numeric(w[base + d]);
        if (w.size() == base + 6) for (int d = 0; d < 3; d++) image[3 * i + d] = inumeric(w[base + 3 + d]);
        crank[i] = file_rank++;   // local order of the reference = file order (1 rank)
        atoms_read = true;
      } else if (cur == "Velocities") {
        if (w.size() < 4) throw LammpsError("Incorrect velocity format in data file");
        int id = inumeric(w[0]);
        if (id < 1 || id > natoms) throw LammpsError("Invalid atom ID in Velocities section of data file");
        for (int d = 0; d < 3; d++) v[3 * (id - 1) + d] = numeric(w[1 + d]);
      } else if (cur == "Bonds") {
        if (w.size() != 4) throw LammpsError("Incorrect format of Bonds section in data file");
        int bt = inumeric(w[1]), a1 = inumeric(w[2]), a2 = inumeric(w[3]);
        if (a1 <= 0 || a1 > natoms || a2 <= 0 || a2 > natoms || a1 == a2)
          throw LammpsError("Invalid atom ID in Bonds section of data file");
        if (bt <= 0 || bt > nbondtypes) throw LammpsError("Invalid bond type in Bonds section of data file");
        b_t.push_back(bt); b_1.push_back(a1); b_2.push_back(a2);
      } else if (cur == "Angles") {          // src/atom.cpp:1290-1353
        if (w.size() != 5) throw LammpsError("Incorrect format of Angles section in data file");
        int at = inumeric(w[1]), a1 = inumeric(w[2]), a2 = inumeric(w[3]), a3 = inumeric(w[4]);
        if (a1 <= 0 || a1 > natoms || a2 <= 0 || a2 > natoms || a3 <= 0 || a3 > natoms || a1 == a2 || a1 == a3 || a2 == a3)
          throw LammpsError("Invalid atom ID in Angles section of data file");
        if (at <= 0 || at > nangletypes) throw LammpsError("Invalid angle type in Angles section of data file");
        g_t.push_back(at); g_1.push_back(a1); g_2.push_back(a2); g_3.push_back(a3);
      }  // other sections are read and ignored
    }
  }
  if (!atoms_read || file_rank != natoms) throw LammpsError("Did not assign all atoms correctly");
  if ((long)b_t.size() != nb_hdr) throw LammpsError("Bonds assigned incorrectly");
  // apply PBC to read-in coordinates is done at setup (Domain::pbc); remap like read_data's domain->remap
  for (int i = 0; i < natoms; i++)
    for (int d = 0; d < 3; d++) {
      double &c = x[3 * i + d];
      while (c < box.lo[d]) { c += box.prd[d]; image[3 * i + d]--; }
      while (c >= box.hi[d]) { c -= box.prd[d]; image[3 * i + d]++; }
    }
  // bonds stored on both atoms: the LE fixes require newton_bond off semantics (SURVEY §0 req. 1);
  // this engine always stores each bond with both of its atoms (src/atom.cpp:1269 newton_bond == 0)
  nbonds = (long)b_t.size();
  std::vector<int> cnt(natoms, 0);
  for (size_t b = 0; b < b_t.size(); b++) { cnt[b_1[b] - 1]++; cnt[b_2[b] - 1]++; }
  int mx = 0;
  for (int c : cnt) mx = std::max(mx, c);
  bpa = std::max(1, mx + extra_bond);
  if (bpa > MAXBPA) throw LammpsError("MI355X engine: too many bonds per atom");
  num_bond.assign(natoms, 0);
  bond_type.assign((size_t)natoms * bpa, 0);
  bond_atom.assign((size_t)natoms * bpa, 0);
  for (size_t b = 0; b < b_t.size(); b++) {
    int m = b_1[b] - 1;
    bond_type[(size_t)m * bpa + num_bond[m]] = b_t[b]; bond_atom[(size_t)m * bpa + num_bond[m]] = b_2[b]; num_bond[m]++;
    m = b_2[b] - 1;
    bond_type[(size_t)m * bpa + num_bond[m]] = b_t[b]; bond_atom[(size_t)m * bpa + num_bond[m]] = b_1[b]; num_bond[m]++;
  }
  // angles: with atom2 first, then atom1, then atom3 (newton_bond off: all three store it, src/atom.cpp:1319-1350)
  if ((long)g_t.size() != nang_hdr) throw LammpsError("Angles assigned incorrectly");
  nangles = (long)g_t.size();
  apa = 0;
  if (nangletypes > MAXTYPES) throw LammpsError("MI355X engine handles at most " + std::to_string(MAXTYPES) + " angle types");
  if (nangletypes > 0 && atom_style != "bond" && atom_style != "atomic") {
    std::vector<int> ca(natoms, 0);
    for (size_t k = 0; k < g_t.size(); k++) { ca[g_1[k] - 1]++; ca[g_2[k] - 1]++; ca[g_3[k] - 1]++; }
    int ma = 0;
    for (int c : ca) ma = std::max(ma, c);
    apa = std::max(1, ma + extra_angle);
    num_angle.assign(natoms, 0);
    angle_type.assign((size_t)natoms * apa, 0); angle_a1 = angle_a2 = angle_a3 = angle_type;
    for (size_t k = 0; k < g_t.size(); k++) {
      const int order[3] = {g_2[k], g_1[k], g_3[k]};
      for (int q = 0; q < 3; q++) {
        const int m = order[q] - 1;
        const size_t c = (size_t)m * apa + num_angle[m]++;
        angle_type[c] = g_t[k]; angle_a1[c] = g_1[k]; angle_a2[c] = g_2[k]; angle_a3[c] = g_3[k];
      }
    }
  } else if (!g_t.empty()) throw LammpsError("Angles section in data file for an atom_style without angles");
  int nt = ntypes + 1;
  pc_eps.assign(nt * nt, 0.0); pc_sig = pc_cut = pc_eps; pc_set.assign(nt * nt, 0);
  special_built = false;
  build_special();
  host_current = true;
  dev_current = false;
  char buf[256];
  snprintf(buf, sizeof buf, "  orthogonal box = (%g %g %g) to (%g %g %g)\n  %d atoms\n  %ld bonds\n  %d = max bonds/atom\n",
           box.lo[0], box.lo[1], box.lo[2], box.hi[0], box.hi[1], box.hi[2], natoms, nbonds, mx);
  say(buf);
}

// Special::build (src/special.cpp:55-): set semantics of onetwo/onethree/onefour + dedup + combine;
// 1-3 / 1-4 are only built when their weights differ from 1.0 (:103-131)
void Engine::build_special() {
  // src/special.cpp:97-131: the lists beyond a level are built unless both the lj and the coul weights there are 1.0
  bool do13 = !(special_lj[2] == 1.0 && special_coul[2] == 1.0 && special_lj[3] == 1.0 && special_coul[3] == 1.0),
       do14 = do13 && !(special_lj[3] == 1.0 && special_coul[3] == 1.0);
  std::vector<std::vector<int>> l12(natoms), l13(natoms), l14(natoms);
  auto has = [](const std::vector<int> &a, int val) { return std::find(a.begin(), a.end(), val) != a.end(); };
  for (int i = 0; i < natoms; i++)
    for (int m = 0; m < num_bond[i]; m++) {
      int u = bond_atom[(size_t)i * bpa + m];
      if (u != i + 1 && !has(l12[i], u)) l12[i].push_back(u);
    }
  if (do13)
    for (int i = 0; i < natoms; i++)
      for (int j : l12[i])
        for (int k : l12[j - 1])
          if (k != i + 1 && !has(l12[i], k) && !has(l13[i], k)) l13[i].push_back(k);
  if (do14)
    for (int i = 0; i < natoms; i++)
      for (int j : l13[i])
        for (int k : l12[j - 1])
          if (k != i + 1 && !has(l12[i], k) && !has(l13[i], k) && !has(l14[i], k)) l14[i].push_back(k);
  size_t maxall = 0;
  for (int i = 0; i < natoms; i++) maxall = std::max(maxall, l12[i].size() + l13[i].size() + l14[i].size());
  maxspecial = std::max<int>(1, (int)maxall + extra_special);
  if (maxspecial > MS_MAX) throw LammpsError("MI355X engine: more than " + std::to_string(MS_MAX) + " special neighbors per atom");
  nspecial.assign(3 * (size_t)natoms, 0);
  special.assign((size_t)natoms * maxspecial, 0);
  for (int i = 0; i < natoms; i++) {
    int *sp = &special[(size_t)i * maxspecial];
    int n = 0;
    for (int u : l12[i]) sp[n++] = u;
    nspecial[3 * i] = n;
    for (int u : l13[i]) sp[n++] = u;
    nspecial[3 * i + 1] = n;
    for (int u : l14[i]) sp[n++] = u;
    nspecial[3 * i + 2] = n;
  }
  special_built = true;
  dev_current = false;
  char buf[160];
  snprintf(buf, sizeof buf, "Finding 1-2 1-3 1-4 neighbors ...\n  special bond factors lj:    %-8g %-8g %-8g\n  %d = max # of special neighbors\n",
           special_lj[1], special_lj[2], special_lj[3], (int)maxall);
  say(buf);
}

// ---------------------------------------------------------------------------------------------
// velocity group-ID create|set|scale|zero   (src/velocity.cpp:52-139 command, :162-405 create, :409-580 set,
// :586-625 scale, :704-728 zero, :733-830 rescale / zero_momentum / zero_rotation, :836-905 options)
// Host-side, once per script: the master copies are in tag order, the sums run in the reference's local order
// (`crank`) so that the result agrees with a 1-rank reference run to the last bits that order can influence.
// RanPark = src/random_park.cpp:41-127 (Park-Miller minimal standard, polar gaussian, coordinate hash for loop geom).
// ---------------------------------------------------------------------------------------------
namespace {
struct RanPark {
  int seed, save = 0;
  double second = 0.0;
  explicit RanPark(int s) : seed(s) {}
  double uniform() {
    int k = seed / 127773;
    seed = 16807 * (seed - k * 127773) - 2836 * k;
    if (seed < 0) seed += 2147483647;
    return (1.0 / 2147483647) * seed;
  }
  double gaussian() {
    if (save) { save = 0; return second; }
    double v1, v2, rsq;
    do {
      v1 = 2.0 * uniform() - 1.0;
      v2 = 2.0 * uniform() - 1.0;
      rsq = v1 * v1 + v2 * v2;
    } while (rsq >= 1.0 || rsq == 0.0);
    double fac = sqrt(-2.0 * log(rsq) / rsq);
    second = v1 * fac;
    save = 1;
    return v2 * fac;
  }
  void reset(int ibase, const double *coord) {   // Jenkins one-at-a-time over the seed and the 24 coordinate bytes
    unsigned int hash = 0;
    auto eat = [&](const void *p, int n) {
      const char *s = (const char *)p;
      for (int i = 0; i < n; i++) { hash += s[i]; hash += (hash << 10); hash ^= (hash >> 6); }
    };
    eat(&ibase, sizeof(int));
    eat(coord, 3 * sizeof(double));
    hash += (hash << 3); hash ^= (hash >> 11); hash += (hash << 15);
    seed = hash & 0x7ffffff;
    if (!seed) seed = 1;
    for (int i = 0; i < 5; i++) uniform();
    save = 0;
  }
};
}  // namespace

void Engine::velocity(std::vector<std::string> &arg) {
  if (arg.size() < 2) throw LammpsError("Illegal velocity command");
  if (!box_exist) throw LammpsError("Velocity command before simulation box is defined");
  if (natoms == 0) throw LammpsError("Velocity command with no atoms existing");
  const int gbit = group_bit(arg[0]);
  if (!gbit) throw LammpsError("Could not find velocity group ID");      // src/velocity.cpp:64-65
  const std::string style = arg[1];
  size_t nfix;
  if (style == "create") nfix = 4;
  else if (style == "set") nfix = 5;
  else if (style == "scale") nfix = 3;
  else if (style == "zero") nfix = 3;
  else if (style == "ramp") throw LammpsError("MI355X engine: velocity ramp is not supported");
  else throw LammpsError("Illegal velocity command");
  if (arg.size() < nfix) throw LammpsError("Illegal velocity command");
  int dist = 0, sum = 0, mom = 1, rot = 0, loop = 0;   // defaults: src/velocity.cpp:69-76
  for (size_t k = nfix; k < arg.size(); k += 2) {
    if (k + 1 >= arg.size()) throw LammpsError("Illegal velocity command");
    const std::string &key = arg[k], &val = arg[k + 1];
    auto yesno = [&]() { if (val == "yes") return 1; if (val == "no") return 0; throw LammpsError("Illegal velocity command"); };
    if (key == "dist") { if (val == "uniform") dist = 0; else if (val == "gaussian") dist = 1; else throw LammpsError("Illegal velocity command"); }
    else if (key == "sum") sum = yesno();
    else if (key == "mom") mom = yesno();
    else if (key == "rot") rot = yesno();
    else if (key == "loop") { if (val == "all") loop = 0; else if (val == "local") loop = 1; else if (val == "geom") loop = 2; else throw LammpsError("Illegal velocity command"); }
    else if (key == "units") { if (val != "box" && val != "lattice") throw LammpsError("Illegal velocity command"); }
    else if (key == "temp" || key == "bias" || key == "rigid") throw LammpsError("MI355X engine: velocity " + key + " is not supported");
    else throw LammpsError("Illegal velocity command");
  }
  for (int t = 1; t <= ntypes; t++)
    if (!mass_set[t]) throw LammpsError("Not all per-type masses are set");   // src/atom.cpp check_mass
  download();   // the run before this command may have left the newest state on the device

  // the reference's local order (1 rank): order[r] = tag-1 of the bead at local index r
  std::vector<int> order(natoms);
  for (int i = 0; i < natoms; i++) order[crank[i]] = i;
  auto m_of = [&](int i) { return mass[type[i]]; };
  // the command acts on the members of its group (`mask[i] & groupbit` throughout src/velocity.cpp); sums, the centre of mass
  // and the temperature (a compute temp on the same group: dof = 3 * members - 3) run over them
  auto in = [&](int i) { return gbit == 1 || (!gmask.empty() && (gmask[i] & gbit)); };
  long members = 0;
  for (int i = 0; i < natoms; i++) members += in(i) ? 1 : 0;
  auto temperature = [&]() {   // compute temp, group all: src/compute_temp.cpp:60-101, extra_dof = 3 (src/compute.cpp:91)
    double t = 0.0;
    for (int r = 0; r < natoms; r++) {
      int i = order[r];
      if (!in(i)) continue;
      t += (v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]) * m_of(i);
    }
    double dof = 3.0 * members - 3.0;
    double tfactor = dof > 0.0 ? mvv2e / (dof * boltz) : 0.0;
    return t * tfactor;
  };
  auto masstotal = [&]() { double m = 0.0; for (int r = 0; r < natoms; r++) if (in(order[r])) m += m_of(order[r]); return m; };
  auto zero_momentum = [&]() {   // src/velocity.cpp:756-780, src/group.cpp:1125-1163
    if (members == 0) throw LammpsError("Cannot zero momentum of no atoms");
    double mt = masstotal(), p[3] = {0, 0, 0};
    for (int r = 0; r < natoms; r++) { int i = order[r]; if (!in(i)) continue; double mo = m_of(i); for (int k = 0; k < 3; k++) p[k] += v[3 * i + k] * mo; }
    if (mt > 0.0) for (int k = 0; k < 3; k++) p[k] /= mt;
    for (int i = 0; i < natoms; i++) if (in(i)) for (int k = 0; k < 3; k++) v[3 * i + k] -= p[k];
  };
  auto zero_rotation = [&]() {   // src/velocity.cpp:786-830, src/group.cpp:1018-1062, :1426-1459, :1583-1625, :1682-1728
    if (members == 0) throw LammpsError("Cannot zero momentum of no atoms");
    double mt = masstotal(), xcm[3] = {0, 0, 0}, L[3] = {0, 0, 0}, I[3][3] = {{0}};
    auto unwrap = [&](int i, double *u) { for (int k = 0; k < 3; k++) u[k] = x[3 * i + k] + image[3 * i + k] * box.prd[k]; };
    double u[3];
    for (int r = 0; r < natoms; r++) { int i = order[r]; if (!in(i)) continue; unwrap(i, u); for (int k = 0; k < 3; k++) xcm[k] += u[k] * m_of(i); }
    if (mt > 0.0) for (int k = 0; k < 3; k++) xcm[k] /= mt;
    for (int r = 0; r < natoms; r++) {
      int i = order[r];
      if (!in(i)) continue;
      unwrap(i, u);
      double dx = u[0] - xcm[0], dy = u[1] - xcm[1], dz = u[2] - xcm[2], mo = m_of(i);
      L[0] += mo * (dy * v[3 * i + 2] - dz * v[3 * i + 1]);
      L[1] += mo * (dz * v[3 * i] - dx * v[3 * i + 2]);
      L[2] += mo * (dx * v[3 * i + 1] - dy * v[3 * i]);
      I[0][0] += mo * (dy * dy + dz * dz); I[1][1] += mo * (dx * dx + dz * dz); I[2][2] += mo * (dx * dx + dy * dy);
      I[0][1] -= mo * dx * dy; I[1][2] -= mo * dy * dz; I[0][2] -= mo * dx * dz;
    }
    I[1][0] = I[0][1]; I[2][1] = I[1][2]; I[2][0] = I[0][2];
    double det = I[0][0] * I[1][1] * I[2][2] + I[0][1] * I[1][2] * I[2][0] + I[0][2] * I[1][0] * I[2][1] -
                 I[0][0] * I[1][2] * I[2][1] - I[0][1] * I[1][0] * I[2][2] - I[2][0] * I[1][1] * I[0][2];
    if (!(det > 1.0e-6)) throw LammpsError("MI355X engine: velocity rot/zero angular needs a non-singular inertia tensor");
    double inv[3][3];
    inv[0][0] = I[1][1] * I[2][2] - I[1][2] * I[2][1];
    inv[0][1] = -(I[0][1] * I[2][2] - I[0][2] * I[2][1]);
    inv[0][2] = I[0][1] * I[1][2] - I[0][2] * I[1][1];
    inv[1][0] = -(I[1][0] * I[2][2] - I[1][2] * I[2][0]);
    inv[1][1] = I[0][0] * I[2][2] - I[0][2] * I[2][0];
    inv[1][2] = -(I[0][0] * I[1][2] - I[0][2] * I[1][0]);
    inv[2][0] = I[1][0] * I[2][1] - I[1][1] * I[2][0];
    inv[2][1] = -(I[0][0] * I[2][1] - I[0][1] * I[2][0]);
    inv[2][2] = I[0][0] * I[1][1] - I[0][1] * I[1][0];
    double w[3];
    for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) inv[a][b] /= det; }
    for (int a = 0; a < 3; a++) w[a] = inv[a][0] * L[0] + inv[a][1] * L[1] + inv[a][2] * L[2];
    for (int i = 0; i < natoms; i++) {
      if (!in(i)) continue;
      unwrap(i, u);
      double dx = u[0] - xcm[0], dy = u[1] - xcm[1], dz = u[2] - xcm[2];
      v[3 * i] -= w[1] * dz - w[2] * dy;
      v[3 * i + 1] -= w[2] * dx - w[0] * dz;
      v[3 * i + 2] -= w[0] * dy - w[1] * dx;
    }
  };
  auto rescale = [&](double t_old, double t_new) {
    if (t_old == 0.0) throw LammpsError("Attempting to rescale a 0.0 temperature");
    double factor = sqrt(t_new / t_old);
    for (int i = 0; i < natoms; i++) if (in(i)) for (int k = 0; k < 3; k++) v[3 * i + k] *= factor;
  };
  auto num = [&](const std::string &s) {
    char *end = nullptr;
    double val = strtod(s.c_str(), &end);
    if (end == s.c_str() || *end) throw LammpsError("Expected floating point parameter instead of '" + s + "' in input script or data file");
    return val;
  };

  if (style == "create") {
    double t_desired = num(arg[2]);
    int seed = atoi(arg[3].c_str());
    if (seed <= 0) throw LammpsError("Illegal velocity create command");
    std::vector<double> vhold;
    if (sum) vhold = v;
    auto draw3 = [&](RanPark &rn, double *o) {
      if (dist == 0) for (int k = 0; k < 3; k++) o[k] = rn.uniform() - 0.5;
      else for (int k = 0; k < 3; k++) o[k] = rn.gaussian();
    };
    double o[3];
    if (loop == 0) {          // loop all: one stream walked in ID order
      RanPark rn(seed);
      for (int i = 0; i < natoms; i++) {
        draw3(rn, o);                       // (a triple for every ID, assigned to the members: src/velocity.cpp:279-300)
        if (!in(i)) continue;
        double factor = 1.0 / sqrt(m_of(i));
        for (int k = 0; k < 3; k++) v[3 * i + k] = o[k] * factor;
      }
    } else if (loop == 1) {   // loop local, as on rank 0 of a 1-rank run: seed + me, 100 warm-up draws, local order
      RanPark rn(seed);
      for (int k = 0; k < 100; k++) rn.uniform();
      for (int r = 0; r < natoms; r++) {
        int i = order[r];
        if (!in(i)) continue;               // (only members draw: :315-331)
        draw3(rn, o);
        double factor = 1.0 / sqrt(m_of(i));
        for (int k = 0; k < 3; k++) v[3 * i + k] = o[k] * factor;
      }
    } else {                  // loop geom: stream re-seeded from each bead's coordinates
      RanPark rn(1);
      for (int i = 0; i < natoms; i++) {
        if (!in(i)) continue;
        rn.reset(seed, &x[3 * i]);
        draw3(rn, o);
        double factor = 1.0 / sqrt(m_of(i));
        for (int k = 0; k < 3; k++) v[3 * i + k] = o[k] * factor;
      }
    }
    if (mom) zero_momentum();
    if (rot) zero_rotation();
    rescale(temperature(), t_desired);
    if (sum) for (int i = 0; i < natoms; i++) if (in(i)) for (int k = 0; k < 3; k++) v[3 * i + k] += vhold[3 * i + k];
  } else if (style == "set") {
    for (int k = 0; k < 3; k++) {
      const std::string &s = arg[2 + k];
      if (s.rfind("v_", 0) == 0) throw LammpsError("MI355X engine: velocity set with variables is not supported");
      if (s == "NULL") continue;
      double val = num(s);
      for (int i = 0; i < natoms; i++) { if (!in(i)) continue; if (sum) v[3 * i + k] += val; else v[3 * i + k] = val; }
    }
  } else if (style == "scale") {
    double t_desired = num(arg[2]);
    rescale(temperature(), t_desired);
  } else {   // zero
    if (arg[2] == "linear") zero_momentum();
    else if (arg[2] == "angular") zero_rotation();
    else throw LammpsError("Illegal velocity command");
  }
  host_current = true;
  dev_current = false;   // the next run uploads the new velocities
}

// ---------------------------------------------------------------------------------------------
// set style ID keyword values ...   (src/set.cpp:60-620 command, :626-700 selection, :706-1000 set, :1013-1041 setrandom)
// styles atom | type | mol (ranges N, N*M, *M, N*, *) and group all; keywords type, type/fraction, mol, x y z, vx vy vz,
// image.  Host-side like the reference; used for marking barrier (CTCF) beads and the like.
// ---------------------------------------------------------------------------------------------
// group ID style args (src/group.cpp:81-535): type | id | molecule with values and ranges a:b[:stride], union, subtract,
// intersect, empty.  Static masks (no `dynamic`), no region / variable styles; a group cannot be deleted.
int Engine::group_bit(const std::string &name) const {
  for (size_t k = 0; k < group_names.size(); k++) if (group_names[k] == name) return 1 << k;
  return 0;
}
// region ID style args [side in|out] [units box|lattice]   (src/region.cpp:305-420 options, region_block.cpp:29-100,
// region_sphere.cpp:29-75, region_cylinder.cpp:31-140, region_union.cpp, region_intersect.cpp).  Static regions only: `move`,
// `rotate`, `open` and variable parameters are refused.  `units lattice` (the default) scales by the lattice spacing, which is
// 1.0 without a `lattice` command (this engine has none): the numbers are box units either way.
void Engine::region_command(std::vector<std::string> &arg) {
  if (arg.size() < 2) throw LammpsError("Illegal region command");
  const std::string &id = arg[0];
  if (arg[1] == "delete") {
    if (arg.size() != 2) throw LammpsError("Illegal region command");
    if (!regions.erase(id)) throw LammpsError("Delete region ID does not exist");
    return;
  }
  if (regions.count(id)) throw LammpsError("Reuse of region ID");
  Region r;
  r.style = arg[1];
  const double BIG = 1.0e20;
  auto bound = [&](const std::string &t, int dim, bool upper) {      // INF / EDGE (region_block.cpp:35-90)
    if (t == "INF" || t == "EDGE") {
      if (!box_exist) throw LammpsError("Cannot use region INF or EDGE when box does not exist");
      if (t == "INF") return upper ? BIG : -BIG;
      return upper ? box.hi[dim] : box.lo[dim];
    }
    if (t.rfind("v_", 0) == 0) throw LammpsError("MI355X engine: region parameters as variables are not supported");
    return numeric(t);
  };
  size_t k = 2;
  auto need = [&](size_t n) { if (k + n > arg.size()) throw LammpsError("Illegal region " + r.style + " command"); };
  if (r.style == "block") {
    need(6);
    for (int q = 0; q < 6; q++) r.p[q] = bound(arg[k + q], q / 2, q & 1);
    k += 6;
    if (r.p[0] > r.p[1] || r.p[2] > r.p[3] || r.p[4] > r.p[5]) throw LammpsError("Illegal region block command");
  } else if (r.style == "sphere") {
    need(4);
    for (int q = 0; q < 4; q++) { if (arg[k + q].rfind("v_", 0) == 0) throw LammpsError("MI355X engine: region parameters as variables are not supported"); r.p[q] = numeric(arg[k + q]); }
    k += 4;
    if (r.p[3] < 0.0) throw LammpsError("Illegal region sphere command");
  } else if (r.style == "cylinder") {
    need(6);
    if (arg[k] != "x" && arg[k] != "y" && arg[k] != "z") throw LammpsError("Illegal region cylinder command");
    r.axis = arg[k][0];
    for (int q = 0; q < 3; q++) { if (arg[k + 1 + q].rfind("v_", 0) == 0) throw LammpsError("MI355X engine: region parameters as variables are not supported"); r.p[q] = numeric(arg[k + 1 + q]); }
    const int dim = r.axis - 'x';
    r.p[3] = bound(arg[k + 4], dim, false);
    r.p[4] = bound(arg[k + 5], dim, true);
    k += 6;
    if (r.p[2] <= 0.0) throw LammpsError("Illegal region cylinder command");
    if (r.p[3] > r.p[4]) throw LammpsError("Illegal region cylinder command");
  } else if (r.style == "union" || r.style == "intersect") {
    need(1);
    const int n = inumeric(arg[k]);
    if (n < 2) throw LammpsError("Illegal region command");
    k++;
    need((size_t)n);
    for (int q = 0; q < n; q++) {
      if (!regions.count(arg[k + q])) throw LammpsError("Region " + r.style + " region ID does not exist");
      r.sub.push_back(arg[k + q]);
    }
    k += n;
  } else throw LammpsError("Unknown region style " + r.style);
  while (k < arg.size()) {
    if (k + 2 > arg.size()) throw LammpsError("Illegal region command");
    if (arg[k] == "side") {
      if (arg[k + 1] == "in") r.interior = true;
      else if (arg[k + 1] == "out") r.interior = false;
      else throw LammpsError("Illegal region command");
    } else if (arg[k] == "units") {
      if (arg[k + 1] != "box" && arg[k + 1] != "lattice") throw LammpsError("Illegal region command");
    } else if (arg[k] == "move" || arg[k] == "rotate" || arg[k] == "open")
      throw LammpsError("MI355X engine: region keyword " + arg[k] + " is not supported (static regions only)");
    else throw LammpsError("Illegal region command");
    k += 2;
  }
  regions[id] = r;
}
bool Engine::region_match(const Region &r, double px, double py, double pz) const {
  bool inside;
  if (r.style == "block") inside = px >= r.p[0] && px <= r.p[1] && py >= r.p[2] && py <= r.p[3] && pz >= r.p[4] && pz <= r.p[5];
  else if (r.style == "sphere") {
    const double dx = px - r.p[0], dy = py - r.p[1], dz = pz - r.p[2];
    inside = sqrt(dx * dx + dy * dy + dz * dz) <= r.p[3];                      // region_sphere.cpp:134-143
  } else if (r.style == "cylinder") {
    const double a = r.axis == 'x' ? px : r.axis == 'y' ? py : pz;
    const double d1 = (r.axis == 'x' ? py : px) - r.p[0], d2 = (r.axis == 'z' ? py : pz) - r.p[1];
    inside = sqrt(d1 * d1 + d2 * d2) <= r.p[2] && a >= r.p[3] && a <= r.p[4];  // region_cylinder.cpp:251-277
  } else {
    // union: inside any sub-region's match; intersect: all of them (region_union.cpp / region_intersect.cpp ::inside)
    inside = r.style == "intersect";
    for (auto &name : r.sub) {
      auto it = regions.find(name);
      if (it == regions.end()) throw LammpsError("Region " + r.style + " region ID does not exist");      // (deleted since)
      const bool m = region_match(it->second, px, py, pz);
      if (r.style == "union") inside = inside || m; else inside = inside && m;
    }
  }
  return !(inside ^ r.interior);
}

void Engine::group_command(std::vector<std::string> &arg) {
  if (!box_exist) throw LammpsError("Group command before simulation box is defined");
  if (arg.size() < 2) throw LammpsError("Illegal group command");
  const std::string &name = arg[0], &style = arg[1];
  if (name == "all") throw LammpsError("Cannot change the group all");     // (src/group.cpp: "all" is fixed)
  if (style == "variable" || style == "dynamic" || style == "static" || style == "include")
    throw LammpsError("MI355X engine: group style " + style + " is not supported");
  download();
  if (gmask.empty()) gmask.assign(natoms, 1);
  int bit = group_bit(name);
  if (style == "delete" || style == "clear") {        // src/group.cpp:103-150
    if (!bit) throw LammpsError("Could not find group " + style + " group ID");
    if (style == "delete") {
      for (auto &f : fixes) if (f->groupbit == bit) throw LammpsError("Cannot delete group currently used by a fix");
      for (auto &c : computes_local_bit) if (c.second == bit) throw LammpsError("Cannot delete group currently used by a compute");
      for (auto &dp : dumps) if (dp.groupbit == bit) throw LammpsError("Cannot delete group currently used by a dump");
    }
    for (int i = 0; i < natoms; i++) gmask[i] &= ~bit;
    if (style == "delete")                               // (the slot is free for the next new group: Group::find_unused)
      for (size_t k = 1; k < group_names.size(); k++) if ((1 << k) == bit) group_names[k].clear();
    dev_current = false;
    return;
  }
  if (!bit) {
    size_t slot = 0;
    for (size_t k = 1; k < group_names.size() && !slot; k++) if (group_names[k].empty()) slot = k;
    if (!slot) {
      if (group_names.size() >= 31) throw LammpsError("Too many groups");
      group_names.push_back(name);
      slot = group_names.size() - 1;
    } else group_names[slot] = name;
    bit = 1 << slot;
  }
  auto inum = [&](const std::string &t) {
    char *end; long v = strtol(t.c_str(), &end, 10);
    if (end == t.c_str() || *end) throw LammpsError("Expected integer parameter instead of '" + t + "' in input script or data file");
    return v;
  };
  if (style == "type" || style == "id" || style == "molecule") {
    if (arg.size() < 3) throw LammpsError("Illegal group command");
    if (style == "molecule" && molecule.empty()) throw LammpsError("Group molecule command requires atom attribute molecule");
    if (arg[2] == "<" || arg[2] == ">" || arg[2] == "<=" || arg[2] == ">=" || arg[2] == "==" || arg[2] == "!=" || arg[2] == "<>") {
      // one comparison against a bound, `<>` = between two bounds, inclusive (src/group.cpp:200-280)
      const std::string &op = arg[2];
      if (arg.size() != (op == "<>" ? 5u : 4u)) throw LammpsError("Illegal group command");
      const long b1 = inum(arg[3]), b2 = op == "<>" ? inum(arg[4]) : 0;
      for (int i = 0; i < natoms; i++) {
        const long v = style == "type" ? type[i] : style == "id" ? i + 1 : molecule[i];
        const bool in = op == "<" ? v < b1 : op == ">" ? v > b1 : op == "<=" ? v <= b1 : op == ">=" ? v >= b1 : op == "==" ? v == b1 :
                        op == "!=" ? v != b1 : (v >= b1 && v <= b2);
        if (in) gmask[i] |= bit;
      }
      dev_current = false;
      return;
    }
    for (size_t k = 2; k < arg.size(); k++) {
      long lo, hi, stride = 1;
      const std::string &t = arg[k];
      size_t c1 = t.find(':');
      if (c1 == std::string::npos) lo = hi = inum(t);
      else {
        size_t c2 = t.find(':', c1 + 1);
        lo = inum(t.substr(0, c1));
        hi = inum(c2 == std::string::npos ? t.substr(c1 + 1) : t.substr(c1 + 1, c2 - c1 - 1));
        if (c2 != std::string::npos) stride = inum(t.substr(c2 + 1));
        if (stride < 1 || lo > hi) throw LammpsError("Illegal range increment value");
      }
      for (int i = 0; i < natoms; i++) {
        const long v = style == "type" ? type[i] : style == "id" ? i + 1 : molecule[i];
        if (v >= lo && v <= hi && (v - lo) % stride == 0) gmask[i] |= bit;
      }
    }
  } else if (style == "union" || style == "subtract" || style == "intersect") {
    if (arg.size() < 3) throw LammpsError("Illegal group command");
    std::vector<int> bits;
    for (size_t k = 2; k < arg.size(); k++) {
      const int b = group_bit(arg[k]);
      if (!b) throw LammpsError("Group ID does not exist");
      bits.push_back(b);
    }
    for (int i = 0; i < natoms; i++) {
      bool in;
      if (style == "union") { in = false; for (int b : bits) in = in || (gmask[i] & b); }
      else if (style == "intersect") { in = true; for (int b : bits) in = in && (gmask[i] & b); }
      else { in = gmask[i] & bits[0]; for (size_t k = 1; k < bits.size(); k++) in = in && !(gmask[i] & bits[k]); }
      if (in) gmask[i] |= bit;       // (an existing group is added to, as in the reference)
    }
  } else if (style == "region") {      // the atoms inside the region NOW (stored coordinates): src/group.cpp:174-186
    if (arg.size() != 3) throw LammpsError("Illegal group command");
    auto it = regions.find(arg[2]);
    if (it == regions.end()) throw LammpsError("Group region ID does not exist");
    for (int i = 0; i < natoms; i++)
      if (region_match(it->second, x[3 * (size_t)i], x[3 * (size_t)i + 1], x[3 * (size_t)i + 2])) gmask[i] |= bit;
  } else if (style == "empty") {
  } else throw LammpsError("Illegal group command");
  dev_current = false;               // the masks travel with the next upload
}

void Engine::set_command(std::vector<std::string> &arg) {
  if (!box_exist) throw LammpsError("Set command before simulation box is defined");
  if (natoms == 0) throw LammpsError("Set command with no atoms existing");
  if (arg.size() < 3) throw LammpsError("Illegal set command");
  download();
  const std::string &style = arg[0], &id = arg[1];
  std::vector<char> select(natoms, 0);
  auto range = [&](long nmax, long &lo, long &hi) {
    size_t star = id.find('*');
    auto num = [&](const std::string &t) { char *end; long v = strtol(t.c_str(), &end, 10); if (end == t.c_str() || *end) throw LammpsError("Invalid range string: " + id); return v; };
    if (star == std::string::npos) lo = hi = num(id);
    else { lo = (star == 0) ? 1 : num(id.substr(0, star)); hi = (star + 1 == id.size()) ? nmax : num(id.substr(star + 1)); }
    if (lo < 1 || hi > nmax || lo > hi) throw LammpsError("Invalid range string: " + id);
  };
  long lo, hi;
  if (style == "atom") { range(2147483647L, lo, hi); for (int i = 0; i < natoms; i++) select[i] = (i + 1 >= lo && i + 1 <= hi); }
  else if (style == "type") { range(ntypes, lo, hi); for (int i = 0; i < natoms; i++) select[i] = (type[i] >= lo && type[i] <= hi); }
  else if (style == "mol") {
    if (molecule.empty()) throw LammpsError("Cannot use set mol with no molecule IDs defined");
    range(2147483647L, lo, hi);
    for (int i = 0; i < natoms; i++) select[i] = (molecule[i] >= lo && molecule[i] <= hi);
  } else if (style == "group") {
    const int bit = group_bit(id);
    if (!bit) throw LammpsError("Could not find set group ID");
    for (int i = 0; i < natoms; i++) select[i] = bit == 1 || (!gmask.empty() && (gmask[i] & bit));
  } else if (style == "region") {      // src/set.cpp:671-678
    auto it = regions.find(id);
    if (it == regions.end()) throw LammpsError("Set region ID does not exist");
    for (int i = 0; i < natoms; i++) select[i] = region_match(it->second, x[3 * (size_t)i], x[3 * (size_t)i + 1], x[3 * (size_t)i + 2]);
  } else throw LammpsError("Illegal set command");
  auto need = [&](size_t k, size_t n) { if (k + n > arg.size()) throw LammpsError("Illegal set command"); };
  auto fnum = [&](const std::string &t) {
    if (t.rfind("v_", 0) == 0) return variable_value(t.substr(2));
    char *end; double v = strtod(t.c_str(), &end);
    if (end == t.c_str() || *end) throw LammpsError("Expected floating point parameter instead of '" + t + "' in input script or data file");
    return v;
  };
  long count = 0;
  size_t k = 2;
  while (k < arg.size()) {
    const std::string &kw = arg[k];
    count = 0;
    if (kw == "type") {
      need(k, 2);
      int t = (int)fnum(arg[k + 1]);
      if (t <= 0 || t > ntypes) throw LammpsError("Invalid value in set command");
      for (int i = 0; i < natoms; i++) if (select[i]) { type[i] = t; count++; }
      k += 2;
    } else if (kw == "type/fraction") {
      need(k, 4);
      int t = (int)fnum(arg[k + 1]);
      double fraction = fnum(arg[k + 2]);
      int seed = (int)fnum(arg[k + 3]);
      if (t <= 0 || t > ntypes) throw LammpsError("Invalid value in set command");
      if (fraction < 0.0 || fraction > 1.0) throw LammpsError("Invalid value in set command");
      if (seed <= 0) throw LammpsError("Invalid random number seed in set command");
      RanPark rp(1);
      for (int i = 0; i < natoms; i++)
        if (select[i]) {
          rp.reset(seed, &x[3 * (size_t)i]);                      // src/set.cpp:1034-1040: generator re-seeded from the coordinates
          if (rp.uniform() > fraction) continue;
          type[i] = t; count++;
        }
      k += 4;
    } else if (kw == "mol") {
      need(k, 2);
      if (molecule.empty()) throw LammpsError("Cannot set this attribute for this atom style");
      int m = (int)fnum(arg[k + 1]);
      if (m < 0) throw LammpsError("Invalid value in set command");
      for (int i = 0; i < natoms; i++) if (select[i]) { molecule[i] = m; count++; }
      k += 2;
    } else if (kw == "x" || kw == "y" || kw == "z" || kw == "vx" || kw == "vy" || kw == "vz") {
      need(k, 2);
      double val = fnum(arg[k + 1]);
      const bool vel = kw[0] == 'v';
      const int d = kw.back() - 'x';
      for (int i = 0; i < natoms; i++) if (select[i]) { (vel ? v : x)[3 * (size_t)i + d] = val; count++; }
      k += 2;
    } else if (kw == "image") {
      need(k, 4);
      for (int d = 0; d < 3; d++) {
        if (arg[k + 1 + d] == "NULL") continue;
        int val = (int)fnum(arg[k + 1 + d]);
        for (int i = 0; i < natoms; i++) if (select[i]) image[3 * (size_t)i + d] = val;
      }
      for (int i = 0; i < natoms; i++) if (select[i]) count++;
      k += 4;
    } else throw LammpsError("MI355X engine: set keyword " + kw + " is not supported");
    char buf[96];
    snprintf(buf, sizeof buf, "Setting atom values ...\n  %ld settings made for %s\n", count, kw.c_str());
    say(buf);
  }
  host_current = true;
  dev_current = false;
}

// write_data (src/write_data.cpp): header, Masses, Atoms (with image flags), Velocities, Bonds (each once)
void Engine::write_data(const std::string &path) {
  download();
  FILE *fp = fopen(path.c_str(), "w");
  if (!fp) throw LammpsError("Cannot open data file " + path);
  fprintf(fp, "LAMMPS data file via write_data, MI355X engine, timestep = %ld\n\n", ntimestep);
  fprintf(fp, "%d atoms\n%d atom types\n%ld bonds\n%d bond types\n", natoms, ntypes, nbonds, nbondtypes);
  if (apa) fprintf(fp, "%ld angles\n%d angle types\n", nangles, nangletypes);
  if (apa && extra_angle) fprintf(fp, "%d extra angle per atom\n", extra_angle);
  if (extra_bond) fprintf(fp, "%d extra bond per atom\n", extra_bond);
  if (extra_special) fprintf(fp, "%d extra special per atom\n", extra_special);
  fprintf(fp, "\n%.17g %.17g xlo xhi\n%.17g %.17g ylo yhi\n%.17g %.17g zlo zhi\n\nMasses\n\n", box.lo[0], box.hi[0], box.lo[1],
          box.hi[1], box.lo[2], box.hi[2]);
  for (int t = 1; t <= ntypes; t++) fprintf(fp, "%d %.17g\n", t, mass[t]);
  fprintf(fp, "\nAtoms # %s\n\n", atom_style.c_str());
  for (int i = 0; i < natoms; i++) {
    if (atom_style == "atomic")
      fprintf(fp, "%d %d %.17g %.17g %.17g %d %d %d\n", i + 1, type[i], x[3 * i], x[3 * i + 1], x[3 * i + 2], image[3 * i],
              image[3 * i + 1], image[3 * i + 2]);
    else
      fprintf(fp, "%d %d %d %.17g %.17g %.17g %d %d %d\n", i + 1, molecule[i], type[i], x[3 * i], x[3 * i + 1], x[3 * i + 2],
              image[3 * i], image[3 * i + 1], image[3 * i + 2]);
  }
  fprintf(fp, "\nVelocities\n\n");
  for (int i = 0; i < natoms; i++) fprintf(fp, "%d %.17g %.17g %.17g\n", i + 1, v[3 * i], v[3 * i + 1], v[3 * i + 2]);
  if (nbonds) {
    fprintf(fp, "\nBonds\n\n");
    long k = 0;
    for (int i = 0; i < natoms; i++)
      for (int m = 0; m < num_bond[i]; m++) {
        int u = bond_atom[(size_t)i * bpa + m];
        if (i + 1 < u) fprintf(fp, "%ld %d %d %d\n", ++k, bond_type[(size_t)i * bpa + m], i + 1, u);
      }
  }
  if (apa && nangles) {        // one row per angle: the copy its central atom stores (src/atom_vec.cpp pack_angle, newton_bond off)
    fprintf(fp, "\nAngles\n\n");
    long k = 0;
    for (int i = 0; i < natoms; i++)
      for (int m = 0; m < num_angle[i]; m++) {
        const size_t c = (size_t)i * apa + m;
        if (angle_a2[c] == i + 1) fprintf(fp, "%ld %d %d %d %d\n", ++k, angle_type[c], angle_a1[c], angle_a2[c], angle_a3[c]);
      }
  }
  fclose(fp);
}


// ---------------------------------------------------------------------------------------------
// dumps: text snapshots in the reference's formats (header_item: src/dump_custom.cpp:510-527, dump_local.cpp:255-279;
// values "%d" / "%g" separated by one blank, no trailing blank: dump_custom.cpp:160-170, 280-304).  Rows are in
// atom-ID order, which is the reference's order at 1 rank with `atom_modify sort 0 0` (or `dump_modify sort id`).
// ---------------------------------------------------------------------------------------------
bool Engine::dump_due(long step) const {
  for (auto &dp : dumps) if (step % dp.every == 0 && dp.last != step) return true;
  return false;
}
void Engine::write_dumps(long step) {
  if (!dump_due(step)) return;
  host_current = false;       // mid-run: the device holds the state of this step
  download();                 // collective when decomposed; every rank then holds the whole system
  host_current = false;
  for (auto &dp : dumps) {
    if (step % dp.every != 0 || dp.last == step) continue;
    dp.last = step;
    if (rank != 0) continue;
    FILE *fp = dp.fp;
    size_t star = dp.path.find('*');
    if (star != std::string::npos) {      // one file per snapshot (src/dump.cpp:590-610)
      std::string name = dp.path.substr(0, star) + std::to_string(step) + dp.path.substr(star + 1);
      fp = fopen(name.c_str(), "w");
    } else if (!fp) fp = dp.fp = fopen(dp.path.c_str(), "w");
    if (!fp) throw LammpsError("Cannot open dump file " + dp.path);
    if (dp.style == "dcd") {
      // CHARMM/NAMD DCD as DumpDCD writes it (src/dump_dcd.cpp:127-200 unit cell, :278-303 frame, :307-357 header):
      // Fortran records (length word before and after), "CORD", float32 coordinates, one x / y / z record per snapshot
      if (dp.fp == nullptr) fp = dp.fp = fopen(dp.path.c_str(), "wb+");
      if (!fp) throw LammpsError("Cannot open dump file " + dp.path);
      auto w32 = [&](uint32_t val) { fwrite(&val, 4, 1, fp); };
      std::vector<int> rows;                                   // the group's atoms in ID order (src/dump_dcd.cpp:69,200)
      for (int i = 0; i < natoms; i++) if (dp.groupbit == 1 || (!gmask.empty() && (gmask[i] & dp.groupbit))) rows.push_back(i);
      const int nrows_dcd = (int)rows.size();
      if (dp.nframes == 0) {
        w32(84); fwrite("CORD", 4, 1, fp);
        w32(0); w32((uint32_t)step); w32((uint32_t)dp.every); w32((uint32_t)step);
        for (int k = 0; k < 5; k++) w32(0);
        float fdt = (float)dt; fwrite(&fdt, 4, 1, fp);
        w32(1);
        for (int k = 0; k < 8; k++) w32(0);
        w32(24); w32(84);
        w32(164); w32(2);
        char title[81];
        memset(title, 0, sizeof title); strncpy(title, "Written by LAMMPS", 80); title[79] = 0; fwrite(title, 80, 1, fp);
        memset(title, ' ', 80); memcpy(title, "REMARKS Created by the MI355X engine", 36); fwrite(title, 80, 1, fp);
        w32(164);
        w32(4); w32((uint32_t)nrows_dcd); w32(4);
      }
      double dim[6] = {box.prd[0], 0.0, box.prd[1], 0.0, 0.0, box.prd[2]};
      w32(48); fwrite(dim, 8, 6, fp); w32(48);
      std::vector<float> c(nrows_dcd);
      for (int d = 0; d < 3; d++) {
        for (int r = 0; r < nrows_dcd; r++) {
          const size_t i = (size_t)rows[r];
          c[r] = (float)(dp.unwrap ? x[3 * i + d] + image[3 * i + d] * box.prd[d] : x[3 * i + d]);
        }
        w32((uint32_t)(nrows_dcd * 4)); fwrite(c.data(), 4, nrows_dcd, fp); w32((uint32_t)(nrows_dcd * 4));
      }
      dp.nframes++;
      fseek(fp, 8, SEEK_SET); w32((uint32_t)dp.nframes);      // NFILE
      fseek(fp, 20, SEEK_SET); w32((uint32_t)step);            // NSTEP
      fseek(fp, 0, SEEK_END);
      fflush(fp);
      continue;
    }
    auto member = [&](int i, int bit) { return bit == 1 || (!gmask.empty() && (gmask[i] & bit)); };
    long nrows = 0;
    for (int i = 0; i < natoms; i++) nrows += member(i, dp.groupbit) ? 1 : 0;
    std::vector<std::array<int, 3>> bonds;     // (type, atom1, atom2) as compute property/local lists them
    if (dp.style == "local") {
      int cbit = 1;                            // rows come from the computes (all columns of one dump: equal counts, dump_local.cpp:284-325)
      for (auto &a : dp.cols) if (a != "index") { cbit = computes_local_bit.at(a.substr(2, a.find('[') - 2)); break; }
      // newton_bond off storage: every bond sits with both atoms, listed once from the lower ID
      // (src/compute_property_local.cpp:420-470)
      for (int i = 0; i < natoms; i++)
        for (int m = 0; m < num_bond[i]; m++) {
          int bt = bond_type[(size_t)i * bpa + m], j = bond_atom[(size_t)i * bpa + m];
          if (bt == 0 || i + 1 > j) continue;
          if (!member(i, cbit) || !member(j - 1, cbit)) continue;       // (both atoms in the compute's group: :477-480)
          bonds.push_back({bt, i + 1, j});
        }
      nrows = (long)bonds.size();
    }
    fprintf(fp, "ITEM: TIMESTEP\n%ld\n", step);
    fprintf(fp, "ITEM: NUMBER OF %s\n%ld\n", dp.style == "local" ? dp.label.c_str() : "ATOMS", nrows);
    fprintf(fp, "ITEM: BOX BOUNDS pp pp pp\n");
    for (int k = 0; k < 3; k++) fprintf(fp, "%-1.16e %-1.16e\n", box.lo[k], box.hi[k]);
    std::string columns;
    for (size_t k = 0; k < dp.cols.size(); k++) columns += (k ? " " : "") + dp.cols[k];
    fprintf(fp, "ITEM: %s %s\n", dp.style == "local" ? dp.label.c_str() : "ATOMS", columns.c_str());
    const size_t nc = dp.cols.size();
    if (dp.style == "local") {
      struct Col { int kind; int attr; };   // kind 0 = index, 1 = compute column
      std::vector<Col> cc;
      for (auto &a : dp.cols) {
        if (a == "index") { cc.push_back({0, 0}); continue; }
        size_t lb = a.find('['), rb = a.find(']');
        auto &attrs = computes_local.at(a.substr(2, lb - 2));
        const std::string &at = attrs[atoi(a.substr(lb + 1, rb - lb - 1).c_str()) - 1];
        cc.push_back({1, at == "btype" ? 0 : at == "batom1" ? 1 : 2});
      }
      for (long r = 0; r < nrows; r++)
        for (size_t k = 0; k < nc; k++) {
          if (cc[k].kind == 0) fprintf(fp, "%ld", r + 1);
          else fprintf(fp, "%g", (double)bonds[r][cc[k].attr]);     // compute values are doubles
          fputc(k + 1 < nc ? ' ' : '\n', fp);
        }
    } else {
      for (int i = 0; i < natoms; i++) {
        if (!member(i, dp.groupbit)) continue;
        for (size_t k = 0; k < nc; k++) {
          const std::string &c = dp.cols[k];
          const double *xi = &x[3 * (size_t)i];
          const int *im = &image[3 * (size_t)i];
          if (c == "id") fprintf(fp, "%d", i + 1);
          else if (c == "mol") fprintf(fp, "%d", molecule[i]);
          else if (c == "type") fprintf(fp, "%d", type[i]);
          else if (c == "mass") fprintf(fp, "%g", mass[type[i]]);
          else if (c == "ix" || c == "iy" || c == "iz") fprintf(fp, "%d", im[c[1] - 'x']);
          else if (c.size() == 1) fprintf(fp, "%g", xi[c[0] - 'x']);
          else if (c[1] == 's') fprintf(fp, "%g", (xi[c[0] - 'x'] - box.lo[c[0] - 'x']) / box.prd[c[0] - 'x']);
          else if (c[1] == 'u') fprintf(fp, "%g", xi[c[0] - 'x'] + im[c[0] - 'x'] * box.prd[c[0] - 'x']);
          else if (c[0] == 'v') fprintf(fp, "%g", v[3 * (size_t)i + (c[1] - 'x')]);
          else fprintf(fp, "%g", f[3 * (size_t)i + (c[1] - 'x')]);
          fputc(k + 1 < nc ? ' ' : '\n', fp);
        }
      }
    }
    if (star != std::string::npos) fclose(fp); else fflush(fp);
  }
}


// ---------------------------------------------------------------------------------------------
// restart files.  The reference's write_restart / read_restart (src/write_restart.cpp, read_restart.cpp) keep the
// system, the force-field coefficients and per-fix state, but neither fix langevin nor the USER-LE fixes save
// their RanMars streams, so a restarted reference run is a different trajectory.  This format (engine-specific,
// binary, host byte order) also keeps the stream positions: `run A; write_restart; [new process] read_restart;
// fix ...; run B` is bit-identical to `run A; run B` in one process.  As in the reference, fixes, thermo and dump
// settings are not stored and have to be re-specified; a fix with a saved ID and the same style picks its state up.
// ---------------------------------------------------------------------------------------------
namespace {
struct Wr {
  FILE *fp;
  template <class T> void pod(const T &v) { fwrite(&v, sizeof(T), 1, fp); }
  void str(const std::string &s) { uint64_t n = s.size(); pod(n); fwrite(s.data(), 1, n, fp); }
  template <class T> void vec(const std::vector<T> &v) { uint64_t n = v.size(); pod(n); if (n) fwrite(v.data(), sizeof(T), n, fp); }
};
struct Rd {
  FILE *fp;
  template <class T> void pod(T &v) { if (fread(&v, sizeof(T), 1, fp) != 1) throw LammpsError("Restart file is truncated"); }
  void str(std::string &s) { uint64_t n; pod(n); if (n > (1u << 20)) throw LammpsError("Restart file is not a lammps_le_amd restart file"); s.resize(n); if (n && fread(&s[0], 1, n, fp) != n) throw LammpsError("Restart file is truncated"); }
  template <class T> void vec(std::vector<T> &v) { uint64_t n; pod(n); if (n > (1ull << 36)) throw LammpsError("Restart file is truncated"); v.resize(n); if (n && fread(v.data(), sizeof(T), n, fp) != n) throw LammpsError("Restart file is truncated"); }
};
const char RESTART_MAGIC[] = "LAMMPS_LE_AMD restart 2";
template <class T> void blob_put(std::vector<unsigned char> &b, const T &v) { const unsigned char *p = (const unsigned char *)&v; b.insert(b.end(), p, p + sizeof(T)); }
template <class T> void blob_get(const std::vector<unsigned char> &b, size_t &off, T &v) { if (off + sizeof(T) > b.size()) throw LammpsError("Restart file: bad fix state"); memcpy(&v, &b[off], sizeof(T)); off += sizeof(T); }
}  // namespace

void le_rng_download(DeviceState &d, int slot, RanMarsInt &r);

// Output::write_restart (src/output.cpp:360-420): one file per due step (`root.step`, or `*` replaced by the step), or two
// files written in turn
void Engine::write_periodic_restart(long step) {
  std::string path;
  if (!restart_b.empty()) { path = restart_toggle ? restart_b : restart_a; restart_toggle ^= 1; }
  else {
    size_t star = restart_a.find('*');
    path = (star == std::string::npos) ? restart_a + "." + std::to_string(step)
                                        : restart_a.substr(0, star) + std::to_string(step) + restart_a.substr(star + 1);
  }
  host_current = false;       // mid-run: the device holds the state of this step
  write_restart(path);
  host_current = false;
}

void Engine::write_restart(const std::string &path) {
  if (!box_exist) throw LammpsError("Write_restart command before simulation box is defined");   // src/write_restart.cpp:62
  download();
  // the LE fixes' generators live on the device while a run has used them
  int slot = 0;
  for (auto &f : fixes) {
    if (!f->force_reneighbor) continue;
    if (auto *a = dynamic_cast<FixExtrusion *>(f.get())) { if (a->rng_on_device) le_rng_download(*dev, slot, a->rng); }
    else if (auto *b = dynamic_cast<FixExLoad *>(f.get())) { if (b->rng_on_device) le_rng_download(*dev, slot, b->rng); }
    else if (auto *c = dynamic_cast<FixExUnload *>(f.get())) { if (c->rng_on_device) le_rng_download(*dev, slot, c->rng); }
    slot++;
  }
  if (rank != 0) return;
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) throw LammpsError("Cannot open restart file " + path);
  Wr w{fp};
  w.str(RESTART_MAGIC);
  w.str(units); w.str(atom_style);
  w.pod(dt); w.pod(skin); w.pod(neigh_every); w.pod(neigh_delay); w.pod(neigh_check); w.pod(newton_pair); w.pod(newton_bond);
  w.pod(sortfreq); w.pod(nextsort); w.pod(special_lj); w.pod(special_coul); w.pod(comm_cutoff); w.pod(ntimestep); w.pod(box);
  w.pod(natoms); w.pod(ntypes); w.pod(nbondtypes); w.pod(extra_bond); w.pod(extra_special); w.pod(bpa); w.pod(maxspecial);
  w.pod(nbonds); w.pod(special_built);
  w.vec(mass); w.vec(mass_set); w.vec(x); w.vec(v); w.vec(type); w.vec(image); w.vec(molecule); w.vec(num_bond);
  w.vec(bond_type); w.vec(bond_atom); w.vec(nspecial); w.vec(special); w.vec(crank);
  w.pod(pair_lj); w.pod(pair_zero); w.pod(pair_cut_global); w.pod(pair_shift); w.pod(pair_mix);
  w.vec(pc_eps); w.vec(pc_sig); w.vec(pc_cut); w.vec(pc_set);
  w.str(bond_style_name);
  uint64_t nh = bond_hybrid_styles.size(); w.pod(nh);
  for (auto &h : bond_hybrid_styles) w.str(h);
  w.pod(bondtab);
  uint64_t nf = 0;
  for (auto &f : fixes) if (f->style == "langevin" || f->force_reneighbor) nf++;
  w.pod(nf);
  for (auto &f : fixes) {
    std::vector<unsigned char> b;
    if (auto *l = dynamic_cast<FixLangevin *>(f.get())) { blob_put(b, l->seed); blob_put(b, l->draws); }
    else if (auto *a = dynamic_cast<FixExtrusion *>(f.get())) { blob_put(b, a->rng); blob_put(b, a->last_break); }
    else if (auto *c = dynamic_cast<FixExLoad *>(f.get())) { blob_put(b, c->rng); blob_put(b, c->last_create); blob_put(b, c->total_create); }
    else if (auto *u = dynamic_cast<FixExUnload *>(f.get())) { blob_put(b, u->rng); blob_put(b, u->last_break); blob_put(b, u->total_break); }
    else continue;
    w.str(f->id); w.str(f->style); w.vec(b);
  }
  // optional tagged sections behind the round-2 layout (a reader without them stops at the end of the file)
  if (!gmask.empty()) {        // groups: names in bit order + masks (as the reference's restart files carry them, src/group.cpp)
    w.str("GROUPS");
    uint64_t ng = group_names.size(); w.pod(ng);
    for (auto &g : group_names) w.str(g);
    w.vec(gmask);
  }
  if (apa) {                   // angles: tables, style and coefficients
    w.str("ANGLES");
    w.pod(nangletypes); w.pod(extra_angle); w.pod(apa); w.pod(nangles);
    w.vec(num_angle); w.vec(angle_type); w.vec(angle_a1); w.vec(angle_a2); w.vec(angle_a3);
    w.str(angle_style_name); w.pod(angtab);
  }
  fclose(fp);
}

void Engine::read_restart(const std::string &path) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) throw LammpsError("Cannot open restart file " + path);
  Rd r{fp};
  try {
    std::string magic;
    r.str(magic);
    if (magic != RESTART_MAGIC) throw LammpsError("Restart file is not a lammps_le_amd restart file");
    r.str(units); r.str(atom_style);
    r.pod(dt); r.pod(skin); r.pod(neigh_every); r.pod(neigh_delay); r.pod(neigh_check); r.pod(newton_pair); r.pod(newton_bond);
    r.pod(sortfreq); r.pod(nextsort); r.pod(special_lj); r.pod(special_coul); r.pod(comm_cutoff); r.pod(ntimestep); r.pod(box);
    r.pod(natoms); r.pod(ntypes); r.pod(nbondtypes); r.pod(extra_bond); r.pod(extra_special); r.pod(bpa); r.pod(maxspecial);
    r.pod(nbonds); r.pod(special_built);
    r.vec(mass); r.vec(mass_set); r.vec(x); r.vec(v); r.vec(type); r.vec(image); r.vec(molecule); r.vec(num_bond);
    r.vec(bond_type); r.vec(bond_atom); r.vec(nspecial); r.vec(special); r.vec(crank);
    r.pod(pair_lj); r.pod(pair_zero); r.pod(pair_cut_global); r.pod(pair_shift); r.pod(pair_mix);
    r.vec(pc_eps); r.vec(pc_sig); r.vec(pc_cut); r.vec(pc_set);
    r.str(bond_style_name);
    uint64_t nh; r.pod(nh);
    bond_hybrid_styles.resize(nh);
    for (auto &h : bond_hybrid_styles) r.str(h);
    r.pod(bondtab);
    uint64_t nf; r.pod(nf);
    restart_fix_state.clear();
    for (uint64_t k = 0; k < nf; k++) {
      std::string id, style;
      std::vector<unsigned char> b;
      r.str(id); r.str(style); r.vec(b);
      b.insert(b.begin(), style.begin(), style.end());
      b.insert(b.begin() + style.size(), (unsigned char)0);
      restart_fix_state[id] = b;
    }
    group_names = {"all"}; gmask.clear();
    atime = 0.0; atimestep = 0;      // (the reference's restart files do not carry the simulation time either)
    nangletypes = 0; extra_angle = 0; apa = 0; nangles = 0;
    num_angle.clear(); angle_type.clear(); angle_a1.clear(); angle_a2.clear(); angle_a3.clear(); angle_style_name.clear();
    for (;;) {                  // optional tagged sections
      int c = fgetc(fp);
      if (c == EOF) break;
      ungetc(c, fp);
      std::string tag;
      r.str(tag);
      if (tag == "GROUPS") {
        uint64_t ng; r.pod(ng);
        if (ng < 1 || ng > 31) throw LammpsError("Restart file is inconsistent");
        group_names.resize(ng);
        for (auto &g : group_names) r.str(g);
        r.vec(gmask);
      } else if (tag == "ANGLES") {
        r.pod(nangletypes); r.pod(extra_angle); r.pod(apa); r.pod(nangles);
        r.vec(num_angle); r.vec(angle_type); r.vec(angle_a1); r.vec(angle_a2); r.vec(angle_a3);
        r.str(angle_style_name); r.pod(angtab);
      } else throw LammpsError("Restart file holds a section this build does not know: " + tag);
    }
  } catch (...) { fclose(fp); throw; }
  fclose(fp);
  if (x.size() != 3 * (size_t)natoms || num_bond.size() != (size_t)natoms) throw LammpsError("Restart file is inconsistent");
  if ((!gmask.empty() && gmask.size() != (size_t)natoms) || (apa && num_angle.size() != (size_t)natoms)) throw LammpsError("Restart file is inconsistent");
  f.assign(3 * (size_t)natoms, 0.0);
  box_exist = true;
  host_current = true;
  dev_current = false;
  char buf[160];
  snprintf(buf, sizeof buf, "Reading restart file ...\n  %d atoms\n  %ld bonds\n", natoms, nbonds);
  say(buf);
}

void Engine::apply_restart_state(Fix *f) {
  auto it = restart_fix_state.find(f->id);
  if (it == restart_fix_state.end()) return;
  const std::vector<unsigned char> &b = it->second;
  std::string style((const char *)b.data());
  size_t off = style.size() + 1;
  if (style == f->style) {
    if (auto *l = dynamic_cast<FixLangevin *>(f)) {
      int seed; uint64_t draws;
      blob_get(b, off, seed); blob_get(b, off, draws);
      if (seed == l->seed) { l->draws = draws; l->dev_ready = false; }   // another seed = a new stream, as in the reference
    } else if (auto *a = dynamic_cast<FixExtrusion *>(f)) { blob_get(b, off, a->rng); blob_get(b, off, a->last_break); }
    else if (auto *c = dynamic_cast<FixExLoad *>(f)) { blob_get(b, off, c->rng); blob_get(b, off, c->last_create); blob_get(b, off, c->total_create); }
    else if (auto *u = dynamic_cast<FixExUnload *>(f)) { blob_get(b, off, u->rng); blob_get(b, off, u->last_break); blob_get(b, off, u->total_break); }
  }
  restart_fix_state.erase(it);
}

}  // namespace lmp_le
