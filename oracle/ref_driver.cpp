// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Drives the *real* reference implementation (header-only, compiled from the sources where they
// lie under /root/reference/include; nothing of the reference is copied into this repository) on a
// netlist given as a "pe-deck" text file (format: oracle/README.md).  Built by oracle/Makefile into
// oracle/_ref/ref_driver (git-ignored).  Two uses:
//   1. golden vectors for tests/golden/ (scripts/make_golden.py runs it);
//   2. the "reference" CPU baseline timed by bench.py on the GPU box's host cores.
//
// Reference entry points exercised (all public members of phy_engine::circult):
//   prepare()          include/phy_engine/circuits/circuit.h:468
//   update_tr_step()   include/phy_engine/circuits/circuit.h:363
//   solve_once()       include/phy_engine/circuits/circuit.h:987
//   analyze()          include/phy_engine/circuits/circuit.h:179
// The Newton loop around solve_once() is re-stated here (counted_solve) only to COUNT iterations;
// `--check-analyze` proves it bit-identical to circult::analyze() on a second copy of the netlist.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/model/models/linear/capacitor.h>
#include <phy_engine/model/models/linear/inductor.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/VAC.h>
#include <phy_engine/model/models/linear/IDC.h>
#include <phy_engine/model/models/linear/IAC.h>
#include <phy_engine/model/models/linear/VCCS.h>
#include <phy_engine/model/models/linear/VCVS.h>
#include <phy_engine/model/models/linear/CCCS.h>
#include <phy_engine/model/models/linear/CCVS.h>
#include <phy_engine/model/models/linear/op_amp.h>
#include <phy_engine/model/models/linear/transformer.h>
#include <phy_engine/model/models/linear/coupled_inductors.h>
#include <phy_engine/model/models/controller/switch.h>
#include <phy_engine/model/models/controller/relay.h>
#include <phy_engine/model/models/linear/transformer_center_tap.h>
#include <phy_engine/model/models/generator/sawtooth.h>
#include <phy_engine/model/models/generator/square.h>
#include <phy_engine/model/models/generator/pulse.h>
#include <phy_engine/model/models/generator/triangle.h>
#include <phy_engine/model/models/non-linear/nmosfet.h>
#include <phy_engine/model/models/non-linear/pmosfet.h>
#include <phy_engine/model/models/non-linear/BJT_NPN.h>
#include <phy_engine/model/models/non-linear/BJT_PNP.h>
#include <phy_engine/model/models/non-linear/PN_junction.h>
#include <phy_engine/model/models/non-linear/full_bridge_rectifier.h>
#include <phy_engine/netlist/impl.h>

namespace pe = ::phy_engine;

struct deck_line
{
    std::string kind;
    std::vector<long> nodes;
    std::vector<double> par;
};

struct deck
{
    long n_nodes{};
    std::vector<deck_line> dev;
};

static int n_pins_of(std::string const& k)
{
    if(k == "XCT") return 5;
    if(k == "RELAY") return 4;
    if(k == "FBR" || k == "VCCS" || k == "VCVS" || k == "CCCS" || k == "CCVS" || k == "OPAMP" || k == "XFMR" || k == "KL") return 4;
    if(k == "NMOS" || k == "PMOS" || k == "NPN" || k == "PNP") return 3;
    return 2;
}

static bool load_deck(char const* path, deck& d)
{
    std::ifstream f(path);
    if(!f) return false;
    std::string line;
    while(std::getline(f, line))
    {
        if(line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string k;
        ss >> k;
        if(k == "nodes")
        {
            ss >> d.n_nodes;
            continue;
        }
        deck_line dl;
        dl.kind = k;
        int np = n_pins_of(k);
        for(int i = 0; i < np; ++i)
        {
            long v;
            ss >> v;
            dl.nodes.push_back(v);
        }
        std::string tok;
        while(ss >> tok) dl.par.push_back(std::strtod(tok.c_str(), nullptr));
        d.dev.push_back(std::move(dl));
    }
    return true;
}

static bool build(pe::circult& c, deck const& d)
{
    auto& nl = c.get_netlist();
    std::vector<pe::model::node_t*> nodes(d.n_nodes + 1, nullptr);
    nodes[0] = &nl.ground_node;
    for(long i = 1; i <= d.n_nodes; ++i) nodes[i] = &create_node(nl);
    for(auto const& l: d.dev)
    {
        pe::model::model_base* m{};
        auto P = [&](size_t i, double def) { return i < l.par.size() ? l.par[i] : def; };
        if(l.kind == "R") m = add_model(nl, pe::model::resistance{.r = P(0, 10.0)}).mod;
        else if(l.kind == "C") m = add_model(nl, pe::model::capacitor{.m_kZimag = P(0, 1e-5)}).mod;
        else if(l.kind == "L") m = add_model(nl, pe::model::inductor{.m_kZimag = P(0, 1e-5)}).mod;
        else if(l.kind == "VDC") m = add_model(nl, pe::model::VDC{.V = P(0, 5.0)}).mod;
        else if(l.kind == "VAC") m = add_model(nl, pe::model::VAC{.m_Vp = P(0, 5.0), .m_omega = P(1, 50.0), .m_phase = P(2, 0.0)}).mod;
        else if(l.kind == "IDC") m = add_model(nl, pe::model::IDC{.I = P(0, 1.0)}).mod;
        else if(l.kind == "D")
        {
            pe::model::PN_junction pn{};
            pn.Is = P(0, pn.Is);
            pn.N = P(1, pn.N);
            pn.Isr = P(2, pn.Isr);
            pn.Nr = P(3, pn.Nr);
            pn.Temp = P(4, pn.Temp);
            pn.Ibv = P(5, pn.Ibv);
            pn.Bv = P(6, pn.Bv);
            pn.Bv_set = P(7, 1.0) != 0.0;
            pn.Area = P(8, pn.Area);
            pn.tt = P(9, pn.tt);
            m = add_model(nl, std::move(pn)).mod;
        }
        else if(l.kind == "FBR") m = add_model(nl, pe::model::full_bridge_rectifier{}).mod;
        else if(l.kind == "IAC") m = add_model(nl, pe::model::IAC{.m_Ip = P(0, 1.0), .m_omega = P(1, 50.0), .m_phase = P(2, 0.0)}).mod;
        else if(l.kind == "VCCS") m = add_model(nl, pe::model::VCCS{.m_g = P(0, 1.0)}).mod;
        else if(l.kind == "VCVS") m = add_model(nl, pe::model::VCVS{.m_mu = P(0, 1.0)}).mod;
        else if(l.kind == "CCCS") m = add_model(nl, pe::model::CCCS{.m_alpha = P(0, 1.0)}).mod;
        else if(l.kind == "CCVS") m = add_model(nl, pe::model::CCVS{.m_r = P(0, 1.0)}).mod;
        else if(l.kind == "OPAMP") m = add_model(nl, pe::model::op_amp{.mu = P(0, 1e5)}).mod;
        else if(l.kind == "XFMR") m = add_model(nl, pe::model::transformer{.n = P(0, 1.0)}).mod;
        else if(l.kind == "SW") m = add_model(nl, pe::model::single_pole_switch{.cut_through = P(0, 0.0) != 0.0}).mod;
        else if(l.kind == "SAW") m = add_model(nl, pe::model::sawtooth_gen{.Vh = P(0, 5.0), .Vl = P(1, 0.0), .freq = P(2, 1e3), .phase = P(3, 0.0)}).mod;
        else if(l.kind == "SQR")
            m = add_model(nl, pe::model::square_gen{.Vh = P(0, 5.0), .Vl = P(1, 0.0), .freq = P(2, 1e3), .duty = P(3, 0.5), .phase = P(4, 0.0)}).mod;
        else if(l.kind == "PULSE")
            m = add_model(nl, pe::model::pulse_gen{.Vh = P(0, 5.0), .Vl = P(1, 0.0), .freq = P(2, 1e3), .duty = P(3, 0.5), .phase = P(4, 0.0), .tr = P(5, 0.0), .tf = P(6, 0.0)}).mod;
        else if(l.kind == "TRI") m = add_model(nl, pe::model::triangle_gen{.Vh = P(0, 5.0), .Vl = P(1, 0.0), .freq = P(2, 1e3), .phase = P(3, 0.0)}).mod;
        else if(l.kind == "NMOS") m = add_model(nl, pe::model::nmosfet{.Kp = P(0, 1e-3), .lambda = P(1, 0.0), .Vth = P(2, 1.0)}).mod;
        else if(l.kind == "PMOS") m = add_model(nl, pe::model::pmosfet{.Kp = P(0, 1e-3), .lambda = P(1, 0.0), .Vth = P(2, 1.0)}).mod;
        else if(l.kind == "NPN")
            m = add_model(nl, pe::model::BJT_NPN{.Is = P(0, 1e-16), .N = P(1, 1.0), .BetaF = P(2, 100.0), .Temp = P(3, 27.0), .Area = P(4, 1.0)}).mod;
        else if(l.kind == "PNP")
            m = add_model(nl, pe::model::BJT_PNP{.Is = P(0, 1e-16), .N = P(1, 1.0), .BetaF = P(2, 100.0), .Temp = P(3, 27.0), .Area = P(4, 1.0)}).mod;
        else if(l.kind == "RELAY")
        {
            pe::model::relay r{};
            r.Von = P(0, 5.0);
            r.Voff = P(1, 3.0);
            m = add_model(nl, std::move(r)).mod;
        }
        else if(l.kind == "XCT") m = add_model(nl, pe::model::transformer_center_tap{.n_total = P(0, 1.0)}).mod;
        else if(l.kind == "KL") m = add_model(nl, pe::model::coupled_inductors{.L1 = P(0, 1e-3), .L2 = P(1, 1e-3), .k = P(2, 0.99)}).mod;
        else
        {
            std::fprintf(stderr, "ref_driver: unknown device kind %s\n", l.kind.c_str());
            return false;
        }
        for(size_t p = 0; p < l.nodes.size(); ++p)
        {
            long id = l.nodes[p];
            if(id < 0) continue;  // unconnected pin
            if(id > d.n_nodes) return false;
            add_to_node(nl, *m, p, *nodes[id]);
        }
    }
    return true;
}

// Newton loop of circult::solve() (circuit.h:892-985) with an iteration counter.
static int counted_solve(pe::circult& c)
{
    if(c.at == pe::analyze_type::AC || !c.has_nonlinear_device()) return c.solve_once() ? 1 : -1;
    double const v_abstol{c.env.V_eps_max > 0.0 ? c.env.V_eps_max : 1e-6};
    double const v_reltol{c.env.V_epsr_max > 0.0 ? c.env.V_epsr_max : 1e-3};
    double const i_abstol{c.env.I_eps_max > 0.0 ? c.env.I_eps_max : 1e-12};
    double const i_reltol{c.env.I_epsr_max > 0.0 ? c.env.I_epsr_max : v_reltol};
    std::vector<std::complex<double>> pv(c.node_counter), pi(c.size_t_to_branch_p.size());
    for(int iter = 0; iter < 64; ++iter)
    {
        for(size_t i = 0; i < c.node_counter; ++i) pv[i] = c.size_t_to_node_p[i]->node_information.an.voltage;
        for(size_t i = 0; i < pi.size(); ++i) pi[i] = c.size_t_to_branch_p[i]->current;
        if(!c.solve_once()) return -1;
        bool conv = true;
        for(size_t i = 0; i < c.node_counter && conv; ++i)
        {
            auto const vn = c.size_t_to_node_p[i]->node_information.an.voltage;
            double const tol = v_abstol + v_reltol * std::max(std::abs(vn), std::abs(pv[i]));
            if(std::abs(vn - pv[i]) > tol) conv = false;
        }
        for(size_t i = 0; i < pi.size() && conv; ++i)
        {
            auto const in = c.size_t_to_branch_p[i]->current;
            double const tol = i_abstol + i_reltol * std::max(std::abs(in), std::abs(pi[i]));
            if(std::abs(in - pi[i]) > tol) conv = false;
        }
        if(conv)
        {
            if(c.at == pe::analyze_type::OP || c.at == pe::analyze_type::DC || c.at == pe::analyze_type::TROP)
            {
                for(auto& blk: c.nl.models)
                    for(auto m{blk.begin}; m != blk.curr; ++m)
                        if(m->type == pe::model::model_type::normal && m->ptr) (void)m->ptr->save_op();
            }
            return iter + 1;
        }
    }
    return -2;  // not converged in 64
}

static std::vector<double> snapshot(pe::circult const& c)
{
    auto x = c.capture_solution_vector();
    std::vector<double> r(x.size());
    for(size_t i = 0; i < x.size(); ++i) r[i] = x[i].real();
    return r;
}

static void usage()
{
    std::fprintf(stderr,
                 "usage: ref_driver <deck> --analysis TR|DC|OP|TROP [--dt X --steps K] [--gmin G] [--ropen R]\n"
                 "                  [--snap s1,s2,..|all] [--out prefix] [--dump-mna] [--check-analyze]\n"
                 "       ref_driver <deck> --bench --dt X --steps K [--warmup W] [--gmin G]\n");
}

int main(int argc, char** argv)
{
    if(argc < 3)
    {
        usage();
        return 2;
    }
    char const* deck_path = argv[1];
    std::string analysis = "TR", out = "ref_out", snaps, omegas;
    double dt = 0.0, gmin = 0.0, ropen = 0.0;
    long steps = 0, warmup = 1;
    bool bench = false, dump_mna = false, check_analyze = false;
    for(int i = 2; i < argc; ++i)
    {
        std::string a = argv[i];
        auto next = [&]() -> char const* { return (i + 1 < argc) ? argv[++i] : ""; };
        if(a == "--analysis") analysis = next();
        else if(a == "--dt") dt = std::strtod(next(), nullptr);
        else if(a == "--steps") steps = std::strtol(next(), nullptr, 10);
        else if(a == "--warmup") warmup = std::strtol(next(), nullptr, 10);
        else if(a == "--gmin") gmin = std::strtod(next(), nullptr);
        else if(a == "--ropen") ropen = std::strtod(next(), nullptr);
        else if(a == "--snap") snaps = next();
        else if(a == "--omegas") omegas = next();
        else if(a == "--out") out = next();
        else if(a == "--bench") bench = true;
        else if(a == "--dump-mna") dump_mna = true;
        else if(a == "--check-analyze") check_analyze = true;
        else
        {
            usage();
            return 2;
        }
    }

    deck d;
    if(!load_deck(deck_path, d))
    {
        std::fprintf(stderr, "ref_driver: cannot read %s\n", deck_path);
        return 2;
    }

    pe::analyze_type at = pe::analyze_type::TR;
    if(analysis == "DC") at = pe::analyze_type::DC;
    else if(analysis == "OP") at = pe::analyze_type::OP;
    else if(analysis == "TROP") at = pe::analyze_type::TROP;
    else if(analysis == "AC") at = pe::analyze_type::AC;
    else if(analysis == "ACOP") at = pe::analyze_type::ACOP;

    pe::circult c{};
    c.set_analyze_type(at);
    c.env.g_min = gmin;
    if(ropen > 0.0) c.env.r_open = ropen;
    if(!build(c, d)) return 2;

    using clk = std::chrono::steady_clock;

    if(bench)
    {
        // protocol of benchmark/0001.models/100000_random_links_cpu.cpp:202-228:
        // build -> prepare -> warm-up -> timed loop; here the loop is the TR loop of circuit.h:242-254.
        c.analyzer_setting.tr.t_step = dt;
        c.prepare();
        long iters = 0;
        for(long s = 0; s < warmup; ++s)
        {
            c.update_tr_step(dt);
            c.tr_duration += dt;
            if(counted_solve(c) < 0) return 1;
        }
        auto t0 = clk::now();
        for(long s = 0; s < steps; ++s)
        {
            c.update_tr_step(dt);
            c.tr_duration += dt;
            int it = counted_solve(c);
            if(it < 0) return 1;
            iters += it;
        }
        auto t1 = clk::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        std::printf("{\"steps\": %ld, \"newton_iters\": %ld, \"seconds\": %.9g, \"steps_per_s\": %.9g, \"newton_iters_per_s\": %.9g, \"rows\": %zu}\n",
                    steps,
                    iters,
                    sec,
                    steps / sec,
                    iters / sec,
                    c.node_counter + c.branch_counter);
        return 0;
    }

    // which steps to snapshot
    std::vector<long> snap_steps;
    bool snap_all = (snaps == "all");
    if(!snap_all && !snaps.empty())
    {
        std::istringstream ss(snaps);
        std::string t;
        while(std::getline(ss, t, ',')) snap_steps.push_back(std::strtol(t.c_str(), nullptr, 10));
    }
    auto want = [&](long s)
    {
        if(snap_all) return true;
        for(long v: snap_steps)
            if(v == s) return true;
        return false;
    };

    std::vector<std::vector<double>> snap_x;
    std::vector<long> snap_at;
    std::vector<int> newton_per_step;
    int fail_step = -1;

    if(at == pe::analyze_type::AC || at == pe::analyze_type::ACOP)
    {
        // the reference's own analyze() per frequency point (prepare; OP solve when non-linear / ACOP; AC solve_once):
        // snapshot = [Re x ; Im x]
        std::istringstream ss(omegas);
        std::string t;
        long idx = 0;
        while(std::getline(ss, t, ','))
        {
            c.analyzer_setting.ac.sweep = pe::analyzer::AC::sweep_type::single;
            c.analyzer_setting.ac.omega = std::strtod(t.c_str(), nullptr);
            bool const ok = c.analyze();
            newton_per_step.push_back(ok ? 1 : -1);
            if(!ok)
            {
                fail_step = static_cast<int>(idx);
                break;
            }
            auto const xc = c.capture_solution_vector();
            std::vector<double> x(2 * xc.size());
            for(size_t i = 0; i < xc.size(); ++i)
            {
                x[i] = xc[i].real();
                x[xc.size() + i] = xc[i].imag();
            }
            snap_x.push_back(std::move(x));
            snap_at.push_back(idx++);
        }
    }
    else
    {
    c.analyzer_setting.tr.t_step = dt;
    c.analyzer_setting.tr.t_stop = dt * static_cast<double>(steps);
    c.prepare();

    if(at == pe::analyze_type::DC || at == pe::analyze_type::OP)
    {
        int it = counted_solve(c);
        newton_per_step.push_back(it);
        if(it < 0) fail_step = 0;
        snap_x.push_back(snapshot(c));
        snap_at.push_back(0);
    }
    else
    {
        if(at == pe::analyze_type::TROP)
        {
            int it = counted_solve(c);
            newton_per_step.push_back(it);
            if(it < 0) fail_step = 0;
            if(want(0))
            {
                snap_x.push_back(snapshot(c));
                snap_at.push_back(0);
            }
            c.at = pe::analyze_type::TR;
        }
        for(long s = 1; s <= steps && fail_step < 0; ++s)
        {
            c.update_tr_step(dt);
            auto const prev = c.tr_duration;
            c.tr_duration = prev + dt;
            int it = counted_solve(c);
            newton_per_step.push_back(it);
            if(it < 0)
            {
                c.tr_duration = prev;
                fail_step = static_cast<int>(s);
                break;
            }
            if(want(s))
            {
                snap_x.push_back(snapshot(c));
                snap_at.push_back(s);
            }
        }
    }
    }

    size_t const rows = c.node_counter + c.branch_counter;

    int analyze_equal = -1;
    if(check_analyze && fail_step < 0 && (at == pe::analyze_type::TR || at == pe::analyze_type::DC || at == pe::analyze_type::OP))
    {
        pe::circult c2{};
        c2.set_analyze_type(at);
        c2.env.g_min = gmin;
        if(ropen > 0.0) c2.env.r_open = ropen;
        build(c2, d);
        // run analyze() one step at a time (t_stop = dt) so FP accumulation of the loop bound cannot change the count
        bool ok = true;
        if(at == pe::analyze_type::TR)
        {
            c2.analyzer_setting.tr.t_step = dt;
            c2.analyzer_setting.tr.t_stop = dt * 0.5;  // exactly one step per analyze() call
            for(long s = 0; s < steps && ok; ++s) ok = c2.analyze();
        }
        else { ok = c2.analyze(); }
        auto xa = snapshot(c2);
        auto xb = snapshot(c);
        analyze_equal = ok && xa.size() == xb.size() && std::memcmp(xa.data(), xb.data(), xa.size() * sizeof(double)) == 0;
    }

    {
        std::string p = out + ".bin";
        FILE* f = std::fopen(p.c_str(), "wb");
        if(!f) return 2;
        for(auto const& x: snap_x) std::fwrite(x.data(), sizeof(double), x.size(), f);
        std::fclose(f);
    }
    if(dump_mna)
    {
        // values of the LAST solve_once (pattern + values + rhs), CSR, real parts
        std::string p = out + ".mna.txt";
        FILE* f = std::fopen(p.c_str(), "w");
        std::fprintf(f, "%zu\n", rows);
        for(size_t r = 0; r < rows; ++r)
            for(auto const& [col, v]: c.mna.A[r]) std::fprintf(f, "A %zu %zu %.17g\n", r, static_cast<size_t>(col), v.real());
        for(auto const& [r, v]: c.mna.Z) std::fprintf(f, "Z %zu %.17g\n", static_cast<size_t>(r), v.real());
        std::fclose(f);
    }
    {
        std::string p = out + ".json";
        FILE* f = std::fopen(p.c_str(), "w");
        std::fprintf(f, "{\"rows\": %zu, \"nodes\": %zu, \"branches\": %zu, \"analysis\": \"%s\", \"dt\": %.17g, \"steps\": %ld, \"gmin\": %.17g,\n",
                     rows, c.node_counter, c.branch_counter, analysis.c_str(), dt, steps, gmin);
        std::fprintf(f, " \"fail_step\": %d, \"analyze_bit_equal\": %d, \"t_end\": %.17g,\n \"snap_steps\": [", fail_step, analyze_equal, c.tr_duration);
        for(size_t i = 0; i < snap_at.size(); ++i) std::fprintf(f, "%s%ld", i ? "," : "", snap_at[i]);
        std::fprintf(f, "],\n \"newton_iters\": [");
        for(size_t i = 0; i < newton_per_step.size(); ++i) std::fprintf(f, "%s%d", i ? "," : "", newton_per_step[i]);
        std::fprintf(f, "]}\n");
        std::fclose(f);
    }
    return 0;
}
