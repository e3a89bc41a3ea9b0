// tests/cpp/overlay_batch.cpp -- host-stamp overlay on a BATCH of instances (include/pe_hip.h: PE_HIP_OVERLAY_INSTANCE), straight
// through the C ABI.  exit 0 = pass.
//
// Circuit: V (1 branch) -- R (device table, per-instance value) -- node 2 -- { cubic conductor i = g v + k v^3 (host hook, k per
// instance) || capacitor with its OWN trapezoidal companion (host hook: history per instance) } -- ground.  The overlay models keep
// state per instance, as a plug-in model's object would (junction voltages, companion histories): the callback switches on
// PE_HIP_OVERLAY_INSTANCE.  One operating point + 6 transient steps on three instances at once must equal the same three instances run
// one at a time with batch = 1 -- to the last bit: the per-instance arithmetic does not depend on the batch (same kernels, same
// geometry: the split / host-driven schedule whenever an overlay is present).
// Round 4: small-signal AC on the batch as well (PE_HIP_OVERLAY_AC per instance behind PE_HIP_OVERLAY_INSTANCE -- the reference runs every
// model's iterate_ac in its AC loop, circuit.h:389-431): an AC current source drives node 2, the host models stamp their small-signal
// admittance g + 3 k v_op^2 + j omega C at each instance's own operating point; phasors of the batch = phasors of the single runs bit for
// bit, and equal to I / (1/R + Y) in closed form.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include <pe_hip.h>

namespace
{
    struct host_models
    {
        std::vector<double> k;                 // cubic coefficient per instance
        std::vector<double> hist, gprev, vlast;  // companion state of the host capacitor per instance
        double g{1e-3}, cap{2e-9};
        int current{0};
        int instance_events{0};
        int ac_events{0};
        double dt_now{0.0};
    };

    // cells: (1,1) of node 2 (0-based row 1); rhs row 1
    int hook(void* user, int event, int mode, double t, double dt, double const* x, double* a, double* b)
    {
        auto& m = *static_cast<host_models*>(user);
        (void)t;
        if(event == PE_HIP_OVERLAY_INSTANCE)
        {
            if(mode < 0 || mode >= static_cast<int>(m.k.size())) return 1;
            m.current = mode;
            ++m.instance_events;
            return 0;
        }
        int const i = m.current;
        if(event == PE_HIP_OVERLAY_STEP)
        {
            // step_changed_tr of a capacitor (capacitor.h:106-128): Ieq <- -(g + g_prev) v_prev - Ieq
            double const v_prev = x[1], g_new = 2.0 * m.cap / dt;
            m.hist[i] = -(g_new + m.gprev[i]) * v_prev - m.hist[i];
            m.gprev[i] = g_new;
            m.dt_now = dt;
            return 0;
        }
        if(event == PE_HIP_OVERLAY_ITERATE)
        {
            double const v = x[1];
            // cubic conductor linearised at v: i = (g + 3 k v^2) v' - 2 k v^3
            double const gd = m.g + 3.0 * m.k[i] * v * v, ieq = -2.0 * m.k[i] * v * v * v;
            bool const tr = mode == PE_HIP_MODE_TR;
            a[0] = gd + (tr ? m.gprev[i] : 0.0);
            b[0] = -ieq - (tr ? m.hist[i] : 0.0);
            m.vlast[i] = v;
            return 0;
        }
        if(event == PE_HIP_OVERLAY_AC)
        {
            // iterate_ac of the two host models at the operating point x (t carries omega): one cell, a = [re | im]
            double const v = x[1];
            a[0] = m.g + 3.0 * m.k[i] * v * v;
            a[1] = t * m.cap;
            b[0] = b[1] = 0.0;
            ++m.ac_events;
            return 0;
        }
        return 0;  // CONVERGED: accept
    }

    bool check(int rc, pe_hip_engine* h, char const* what)
    {
        if(rc == PE_HIP_OK) return true;
        std::fprintf(stderr, "%s: rc %d: %s\n", what, rc, pe_hip_last_error(h));
        return false;
    }

    // runs instances [first, first + count) as one batch; returns x of every instance after the DC point and after each TR step
    constexpr double ac_omega = 2.0e5, ac_amp = 1.0e-3;
    bool run(int first, int count, std::vector<double> const& r_all, std::vector<double> const& k_all, std::vector<std::vector<double>>& out, int& instance_events,
             std::vector<std::vector<double>>& phasors)
    {
        pe_hip_engine* h{};
        if(!check(pe_hip_create(0, &h), nullptr, "create")) return false;
        host_models m;
        m.k.assign(k_all.begin() + first, k_all.begin() + first + count);
        m.hist.assign(count, 0.0);
        m.gprev.assign(count, 0.0);
        m.vlast.assign(count, 0.0);
        int const rows1[1] = {1}, cols1[1] = {1}, rhs1[1] = {1};
        double const rep[1] = {1e-3};
        if(!check(pe_hip_set_overlay(h, 1, rows1, cols1, rep, 1, rhs1, /*nonlinear=*/1, &hook, &m), h, "set_overlay")) return false;
        int const vn[2] = {1, 0}, vb[1] = {0}, rn[2] = {1, 2}, in[2] = {0, 2};
        double const vpar[1] = {3.0}, ipar[3] = {ac_amp, ac_omega, 0.0};  // (IAC: nothing in OP / DC, Ip sin(omega t) in TR, the phasor Ip in AC)
        std::vector<double> rpar(r_all.begin() + first, r_all.begin() + first + count);  // [batch][1][1]
        pe_hip_device_table tabs[3]{};
        tabs[0] = {PE_HIP_VDC, 1, vn, vb, vpar, 0};
        tabs[1] = {PE_HIP_R, 1, rn, nullptr, rpar.data(), 1};
        tabs[2] = {PE_HIP_IAC, 1, in, nullptr, ipar, 0};
        pe_hip_options opt{};
        opt.g_min = 1e-12;
        if(!check(pe_hip_set_options(h, &opt), h, "set_options")) return false;
        if(!check(pe_hip_load_circuit(h, 2, 1, count, 3, tabs), h, "load_circuit")) return false;
        pe_hip_run_stats st{};
        auto snap = [&]
        {
            std::vector<double> x(static_cast<size_t>(3) * count);
            if(!check(pe_hip_get_solution(h, 0, count, x.data()), h, "get_solution")) return false;
            for(int b = 0; b < count; ++b) out[first + b].insert(out[first + b].end(), x.begin() + 3 * b, x.begin() + 3 * b + 3);
            return true;
        };
        if(!check(pe_hip_analyze_dc(h, PE_HIP_MODE_DC, &st), h, "analyze_dc") || st.n_failed) return false;
        if(!snap()) return false;
        {
            // small-signal point at the operating point just solved: phasors [re(3) | im(3)] per instance
            if(!check(pe_hip_analyze_ac(h, ac_omega, &st), h, "analyze_ac")) return false;
            std::vector<double> re(static_cast<size_t>(3) * count), im(static_cast<size_t>(3) * count);
            if(!check(pe_hip_get_solution_ac(h, 0, count, re.data(), im.data()), h, "get_solution_ac")) return false;
            for(int b = 0; b < count; ++b)
            {
                phasors[first + b].assign(re.begin() + 3 * b, re.begin() + 3 * b + 3);
                phasors[first + b].insert(phasors[first + b].end(), im.begin() + 3 * b, im.begin() + 3 * b + 3);
            }
            if(m.ac_events != count) { std::fprintf(stderr, "PE_HIP_OVERLAY_AC: %d calls for %d instances\n", m.ac_events, count); return false; }
        }
        for(int s = 0; s < 6; ++s)
        {
            if(!check(pe_hip_analyze_tr(h, 1e-7, 1, &st), h, "analyze_tr") || st.n_failed) return false;
            if(!snap()) return false;
        }
        instance_events = m.instance_events;
        pe_hip_destroy(h);
        return true;
    }
}  // namespace

int main()
{
    std::vector<double> const r{1000.0, 1500.0, 700.0}, k{0.0, 2e-4, 8e-4};
    std::vector<std::vector<double>> batched(3), single(3), ph3(3), ph1(3);
    int ev3 = 0, ev1 = 0;
    if(!run(0, 3, r, k, batched, ev3, ph3)) return 1;
    for(int b = 0; b < 3; ++b)
        if(!run(b, 1, r, k, single, ev1, ph1)) return 2;
    if(ev3 == 0 || ev1 != 0)
    {
        std::fprintf(stderr, "PE_HIP_OVERLAY_INSTANCE: %d events in the batch of 3 (expected some), %d with batch = 1 (expected none)\n", ev3, ev1);
        return 3;
    }
    for(int b = 0; b < 3; ++b)
    {
        if(batched[b].size() != 21 || single[b].size() != 21) return 4;
        for(size_t i = 0; i < 21; ++i)
            if(std::memcmp(&batched[b][i], &single[b][i], sizeof(double)) != 0)
            {
                std::fprintf(stderr, "instance %d value %zu: batch %.17g, alone %.17g\n", b, i, batched[b][i], single[b][i]);
                return 5;
            }
        // the operating point solves V = v + R (g v + k v^3): checked against a scalar Newton on the host
        double v = 1.0;
        for(int it = 0; it < 60; ++it) v -= (v + r[b] * (1e-3 * v + k[b] * v * v * v) - 3.0) / (1.0 + r[b] * (1e-3 + 3.0 * k[b] * v * v));
        if(std::fabs(batched[b][1] - v) > 1e-6 + 1e-3 * std::fabs(v) * 1e-3)
        {
            std::fprintf(stderr, "instance %d operating point %.12g, expected %.12g\n", b, batched[b][1], v);
            return 6;
        }
    }
    // AC with host-stamped models on the batch: bit for bit the single runs, and the closed form v2 = I / (1/R + g + 3 k v^2 + j omega C)
    for(int b = 0; b < 3; ++b)
    {
        if(ph3[b].size() != 6 || ph1[b].size() != 6) return 8;
        if(std::memcmp(ph3[b].data(), ph1[b].data(), 6 * sizeof(double)) != 0)
        {
            std::fprintf(stderr, "instance %d AC phasor: batch (%.17g, %.17g), alone (%.17g, %.17g)\n", b, ph3[b][1], ph3[b][4], ph1[b][1], ph1[b][4]);
            return 9;
        }
        double const v = batched[b][1], yr = 1.0 / r[b] + 1e-3 + 3.0 * k[b] * v * v + 1e-12, yi = ac_omega * 2e-9, den = yr * yr + yi * yi;
        double const wre = ac_amp * yr / den, wim = -ac_amp * yi / den;
        if(std::fabs(ph3[b][1] - wre) > 1e-9 * std::fabs(wre) + 1e-15 || std::fabs(ph3[b][4] - wim) > 1e-9 * std::fabs(wim) + 1e-15)
        {
            std::fprintf(stderr, "instance %d AC phasor (%.12g, %.12g), expected (%.12g, %.12g)\n", b, ph3[b][1], ph3[b][4], wre, wim);
            return 10;
        }
    }
    if(ph3[0][1] == ph3[1][1] || ph3[1][1] == ph3[2][1]) return 11;
    // the instances differ (the batch did not collapse onto instance 0)
    if(batched[0][1] == batched[1][1] || batched[1][1] == batched[2][1]) return 7;
    std::printf("overlay on a batch of 3 = three runs of one: bit for bit; v2 = %.9f %.9f %.9f V\n", batched[0][1], batched[1][1], batched[2][1]);
    return 0;
}
