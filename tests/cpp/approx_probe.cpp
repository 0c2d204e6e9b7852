// approx_probe — csrc/approx.cpp on its own (host only: g++ on graph.cpp + approx.cpp, no HIP): reads a patch from stdin, prints the
// analysis as JSON.  tests/test_approx.py drives it, one case per structure the error budget has to get right.
//   cfg <sample_rate> <buffer_size> <channels> | mod <type> | field <m> <f> <value> | ov <m> <f> <n> <v...> | step <m> <ch> <i> <state> <value>
//   wave <m> <n> <v...> | conn <src> <port> <sink> <port> | exact
#include <cmath>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "../../s-rack_amd/csrc/approx.hpp"

using namespace srack;

static void num(std::ostream& o, double x)
{
    if (std::isinf(x)) o << (x > 0 ? "1e999" : "-1e999");
    else if (std::isnan(x)) o << "null";
    else o << x;
}

int main()
{
    Graph g;
    std::vector<VoiceOverride> ov;
    bool exact = false;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string op;
        if (!(in >> op)) continue;
        if (op == "cfg") in >> g.cfg.sample_rate >> g.cfg.buffer_size >> g.cfg.channels;
        else if (op == "mod") { int t; in >> t; if (g.add_module(t) < 0) { std::fprintf(stderr, "add_module: %s\n", last_error()); return 2; } }
        else if (op == "field") { int m, f; double v; in >> m >> f >> v; if (g.set_field(m, f, v) < 0) { std::fprintf(stderr, "set_field: %s\n", last_error()); return 2; } }
        else if (op == "ov") { VoiceOverride o; size_t n; in >> o.module >> o.field >> n; o.values.resize(n); for (double& v : o.values) in >> v; ov.push_back(o); }
        else if (op == "step") { int m, ch, i, st, v; in >> m >> ch >> i >> st >> v; g.set_step(m, ch, i, st, v); }
        else if (op == "wave") { int m; size_t n; in >> m >> n; std::vector<float> w(n); for (float& v : w) in >> v; g.set_wave(m, w.data(), (uint32_t)n, 48000.0f); }
        else if (op == "conn") { int a, ap, b, bp; in >> a >> ap >> b >> bp; if (g.connect(a, ap, b, bp) < 0) { std::fprintf(stderr, "connect: %s\n", last_error()); return 2; } }
        else if (op == "exact") exact = true;
    }
    g.make_plan();
    std::vector<char> live;
    std::vector<uint32_t> port_live;
    int self_loop = -1;
    if (audible(g, live, port_live, &self_loop) != 0) { std::fprintf(stderr, "self loop at %d\n", self_loop); return 3; }
    const ApproxPlan P = plan_approximations(g, live, port_live, ov, exact);
    std::ostream& o = std::cout;
    o.precision(9);
    auto flags = [&](const char* name, const std::vector<char>& v) {
        o << "\"" << name << "\": [";
        for (size_t i = 0; i < v.size(); i++) o << (i ? ", " : "") << (int)v[i];
        o << "], ";
    };
    o << "{";
    flags("osc_exact", P.osc_exact);
    flags("exact_blep", P.exact_blep);
    flags("literal", P.literal);
    flags("sine_loose", P.sine_loose);
    flags("nonlin_loose", P.nonlin_loose);
    flags("saw_fixed", P.saw_fixed);
    flags("live", live);
    o << "\"port_live\": [";
    for (size_t i = 0; i < port_live.size(); i++) o << (i ? ", " : "") << port_live[i];
    o << "], ";
    o << "\"exact_patch\": " << (P.exact_patch ? "true" : "false") << ", \"why\": \"" << P.why << "\", \"bound\": ";
    num(o, P.bound);
    for (int which = 0; which < 2; which++) {
        const auto& t = which ? P.gain : P.mag;
        o << ", \"" << (which ? "gain" : "mag") << "\": [";
        for (size_t m = 0; m < t.size(); m++) {
            o << (m ? ", [" : "[");
            for (size_t p = 0; p < t[m].size(); p++) { if (p) o << ", "; num(o, t[m][p]); }
            o << "]";
        }
        o << "]";
    }
    o << ", \"sweeps\": [";
    for (size_t m = 0; m < g.modules.size(); m++) o << (m ? ", " : "") << (wire_sweeps(g, (int)m) ? 1 : 0);
    o << "]}\n";
    return 0;
}
