// pe_engine_policy.cpp -- launch geometry and symbolic analysis of the resident circuit: which schedule (resident kernel / split), how
// many wavefronts, parts, pivots and LDS doubles by batch size (measured sweeps under profiles/), the PHY_ENGINE_HIP_* tuning knobs
// (INTEGRATION.md), the analysis with its LDS-fit escalation, and the upload of its tables.
#include "pe_engine_internal.hpp"
#include "pe_top_plan.hpp"

using namespace pe_eng;

namespace pe_eng PE_ENG_HIDDEN
{
    // multi-workgroup schedule (one launch per phase and tree level) instead of the single resident kernel
    // Large circuits always: the per-phase kernels fit their register budgets (the factor kernel spills 48 B / lane at 128 VGPRs,
    // the resident kernel 580), which outweighs ~35 launches and one host round trip per Newton iteration once an iteration
    // takes milliseconds.  Small circuits stay in the resident kernel (a time step is microseconds there).
    bool split_launch(pe_hip_engine const* h)
    {
        int const v = knob(h, "SPLIT", -1);  // knob: 1 = always split, 0 = never (resident kernel, one part)
        if(v == 1) return true;
        if(v == 0) return false;
        if(h->overlay_fn && (h->hc.n_ov_a || h->hc.n_ov_b)) return true;  // host-stamped models: the host drives the Newton loop
        if(h->careful) return true;  // an inaccurate solve was detected: the host-driven loop refines / re-matches
        return h->V.n_parts > 1 || h->hc.rows >= 3000;
    }

    // uploads symbolic arrays + allocates per-instance factor storage into `pool`, fills the symbolic part of V
    // test knob: choose the launch geometry as if the batch had this many instances
    int geometry_batch(pe_hip_engine const* h, int batch)
    {
        int const v = knob(h, "GEOMETRY_BATCH", 0);
        return v > 0 ? v : batch;
    }


    // LDS panel need of a front in the panel layout (pe_front.hpp: L panel m x p + U panel p x u, odd leading dimensions)
    long long panel_need(int p, int u) { return static_cast<long long>(pe::pe_ld(p + u)) * p + static_cast<long long>(pe::pe_ld(p)) * u; }
    // a front the ordinary launches cannot take: more pivots than their staged blocks hold, or panels beyond their LDS share
    // (0: they can; 3: it fits half a CU's LDS -- the 8-wavefront launch; 2: it needs the 16-wavefront launch with a CU's whole LDS)
    int lds_class(pe::Symbolic const& S, pe::SymbolicOptions const& so, int s)
    {
        if(S.f_kind[s] == 0 || (S.f_p[s] <= so.max_pivots && panel_need(S.f_p[s], S.f_u[s]) <= so.panel_doubles)) return 0;
        return (so.panel_doubles_mid > 0 && panel_need(S.f_p[s], S.f_u[s]) <= so.panel_doubles_mid) ? 3 : 2;
    }

    // The reference's contract for its solver is "return false, never corrupt" (circuit.h:1517).  A front whose LDS layout -- fixed by
    // build_assembly_lists against the cap of the launch class its level was GIVEN -- exceeds the dynamic LDS of the launch that will
    // actually RUN it would read zeros and drop writes beyond the workgroup's allocation on gfx950 (scripts/lds_oob_probe.hip) without
    // any fault.  So the plan is checked where it is made: every wave front against a wavefront's slot, every cooperative front of a
    // part against the parts' launch, every top front against the launch for_each_top_launch (pe_top_plan.hpp, the launcher's own
    // plan) puts its level on.  A mismatch refuses the load with PE_HIP_ERR_INTERNAL.
    static int check_lds_plan(pe_hip_engine* h, pe::Symbolic const& S, pe::DevView const& V, std::vector<int> const& need, int batch)
    {
        auto refuse = [&](int s, char const* what, long long have)
        {
            return fail(h, PE_HIP_ERR_INTERNAL,
                        "launch plan: front " + std::to_string(s) + " (order " + std::to_string(S.f_p[s] + S.f_u[s]) + ", " + std::to_string(S.f_p[s]) + " pivots, layout " +
                            std::to_string(S.f_mode[s]) + ") needs " + std::to_string(need[static_cast<size_t>(s)]) + " doubles of LDS, " + what + " has " + std::to_string(have));
        };
        for(int s = 0; s < S.nfronts; ++s)
        {
            if(S.f_kind[s] == 0 && need[static_cast<size_t>(s)] > V.lds_slot) return refuse(s, "a wavefront's slot", V.lds_slot);
            if(S.f_kind[s] == 1 && need[static_cast<size_t>(s)] > V.lds_doubles - 2) return refuse(s, "the parts' launch", V.lds_doubles - 2);
        }
        int rc = PE_HIP_OK;
        pe::for_each_top_launch(V, batch, V.high_occupancy && V.n_waves == 4, V.mid_top_limit,
                                [&](pe::TopLaunch const& t)
                                {
                                    for(int l = t.level; l < t.level + t.nlev && rc == PE_HIP_OK; ++l)
                                        for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1] && rc == PE_HIP_OK; ++k)
                                            if(int const s = S.top_list[k]; need[static_cast<size_t>(s)] > t.lds_doubles - 2)
                                                rc = refuse(s, t.kind == 1 ? "its 16-wavefront launch" : (t.kind == 0 ? "its 4-wavefront launch" : "its 8-wavefront launch"), t.lds_doubles - 2);
                                });
        return rc;
    }

    int upload_symbolic(pe_hip_engine* h, Pool& pool, pe::Symbolic& S, pe::SymbolicOptions const& so, pe::DevView& V, int batch)
    {
        V.nfronts = S.nfronts;
        HIPCHK(h, pool.upload(V.f_col0, S.f_col0));
        HIPCHK(h, pool.upload(V.f_p, S.f_p));
        HIPCHK(h, pool.upload(V.f_u, S.f_u));
        HIPCHK(h, pool.upload(V.f_rows_ptr, S.f_rows_ptr));
        HIPCHK(h, pool.upload(V.f_rows, S.f_rows));
        HIPCHK(h, pool.upload(V.f_child_ptr, S.f_child_ptr));
        HIPCHK(h, pool.upload(V.f_child, S.f_child));
        HIPCHK(h, pool.upload(V.f_wstack, S.f_wstack));
        HIPCHK(h, pool.upload(V.f_wpar, S.f_wpar));
        HIPCHK(h, pool.upload(V.f_rel, S.f_rel));
        HIPCHK(h, pool.upload(V.f_inv_off, S.f_inv_off));
        HIPCHK(h, pool.upload(V.f_cnp, S.f_cnp));
        HIPCHK(h, pool.upload(V.f_inv, S.f_inv));
        HIPCHK(h, pool.upload(V.f_bmask, S.f_bmask));
        HIPCHK(h, pool.upload(V.f_asm_ptr, S.f_asm_ptr));
        HIPCHK(h, pool.upload(V.asm_slot, S.asm_slot));
        HIPCHK(h, pool.upload(V.asm_pos, S.asm_pos));
        HIPCHK(h, pool.upload(V.f_lptr, S.f_lptr));
        HIPCHK(h, pool.upload(V.f_uptr, S.f_uptr));
        HIPCHK(h, pool.upload(V.f_sptr, S.f_sptr));
        HIPCHK(h, pool.upload(V.row_src, S.row_src));
        {
            std::vector<int> dst(S.row_src.size(), 0);
            for(size_t k = 0; k < S.row_src.size(); ++k) dst[static_cast<size_t>(S.row_src[k])] = static_cast<int>(k);
            HIPCHK(h, pool.upload(V.row_dst, dst));
            V.row_dyn = nullptr;  // (set with the x-dependent row list of the resident circuit, ensure_symbolic)
        }
        HIPCHK(h, pool.upload(V.col_src, S.col_src));
        HIPCHK(h, pool.upload(V.wave_ptr, S.wave_ptr));
        HIPCHK(h, pool.upload(V.wave_list, S.wave_list));
        HIPCHK(h, pool.upload(V.coop_ptr, S.coop_ptr));
        HIPCHK(h, pool.upload(V.coop_list, S.coop_list));
        HIPCHK(h, pool.upload(V.top_ptr, S.top_ptr));
        HIPCHK(h, pool.upload(V.top_list, S.top_list));
        V.n_parts = S.n_parts;
        V.n_top_levels = static_cast<int>(S.top_ptr.size()) - 1;
        if(V.n_top_levels > 64) return fail(h, PE_HIP_ERR_INTERNAL, "assembly tree has more than 64 top levels");
        for(int l = 0; l < V.n_top_levels; ++l) V.top_cnt[l] = S.top_ptr[l + 1] - S.top_ptr[l];
        V.n_waves = so.n_waves;
        V.high_occupancy = so.shared_cu;
        V.wave_m = so.wave_m;
        V.max_m = std::max(S.max_m, 1);
        V.max_p = so.max_pivots;
        V.wave_p = so.wave_p;
        // a wavefront's slot holds its fronts whole (order <= wave_m, odd leading dimension) + the right-hand-side column -- or, with an
        // explicit wave_slot, the panels of the larger ones
        V.lds_slot = so.wave_slot > 0 ? static_cast<int>(so.wave_slot) : (pe::pe_ld(so.wave_m) + 1) * so.wave_m;
        V.lds_wave_stage = so.wave_m * so.wave_p;              // a wavefront stages the whole m x p panel of its (small) fronts
        V.lds_coop_stage = std::max(V.max_p * V.max_p, std::min(64, V.max_m) * V.max_p);
        // (top fronts regrouped against a CU's LDS may carry more pivots than the fronts of the parts: regroup_wide_top)
        int max_p_top = V.max_p;
        for(int s = 0; s < S.nfronts; ++s)
            if(S.f_kind[s] == 2) max_p_top = std::max(max_p_top, S.f_p[s]);
        V.lds_top_stage = std::max(max_p_top * max_p_top, std::min(64, V.max_m) * max_p_top);
        V.lds_bstack_off = so.wave_m + V.lds_wave_stage + 64;  // t[m] + staged block + partial sums of one wavefront,
        V.lds_sslot = V.lds_bstack_off + std::max(1, S.wave_stack);  // + the backward stack (the solved vectors along one path of a wave subtree)
        V.lds_wave_stage_b = so.wave_p * so.wave_p;            // backward pass: U11 only (front_backward_lean)
        V.lds_bstack_off_b = so.wave_m + V.lds_wave_stage_b + 64;
        V.lds_sslot_b = V.lds_bstack_off_b + std::max(1, S.wave_stack);
        {
            long long need = static_cast<long long>(so.n_waves) * V.lds_slot;
            need = std::max(need, so.panel_doubles + so.panel_reserve);
            need = std::max(need, static_cast<long long>(so.n_waves) * V.lds_sslot);
            need = std::max(need, static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64);
            V.lds_doubles = static_cast<int>(need + 2);
            // the triangular-solve kernels of the split schedule need far less: more of their workgroups fit a CU
            long long const need_solve = std::max(static_cast<long long>(so.n_waves) * V.lds_sslot,
                                                  static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64);
            V.lds_solve_doubles = static_cast<int>(need_solve + 2);
            V.lds_solve_b_doubles = static_cast<int>(std::max(static_cast<long long>(so.n_waves) * V.lds_sslot_b,
                                                              static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64) + 2);
            V.lds_solve_top_doubles = static_cast<int>(std::max<long long>(need_solve, static_cast<long long>(V.max_m) + V.lds_top_stage + so.n_waves * 64) + 2);
        }
        V.factor_doubles = std::max<long long>(S.factor_doubles, 1);
        V.arena_doubles = std::max<long long>(S.arena_doubles, 1);
        // the LDS caps are fixed now: layout of every front + the assembly lists that go with it
        // top levels that leave most CUs without a workgroup run ONE 16-wavefront workgroup per front (k_m2_factor_top_wide): always in
        // the one-workgroup-per-CU geometry (few instances), and on the under-filled levels near the root of a sweep (fronts x instances
        // <= CUs + 25 %).  Such a workgroup owns its CU's LDS: whole-front layout up to order ~141, chain links continued in LDS.
        {
            bool const wide_knob = knob(h, "WIDE_TOP", 1) != 0;
            bool const chain_lds = knob(h, "TOP_CHAIN_LDS", 1) != 0;  // developer knob: 0 = round 2's layout of the top fronts
            long long const whole_cu = h->lds_limit / 8 - 160 - 8;
            // (the population rule counts the instances the GEOMETRY was chosen for -- knob GEOMETRY_BATCH: the host emulation runs the
            //  launch plan of a large sweep on one instance, tests/test_host_logic.py)
            int const gbatch = geometry_batch(h, batch);
            int const wide_wgs = std::max(0, knob(h, "TOP_WIDE_WGS", 320));  // (CUs + 25 %; developer knob for A/B runs)
            for(int l = 0; l < 64; ++l) V.top_wide[l] = (l < V.n_top_levels && wide_knob && (!V.high_occupancy || V.top_cnt[l] * gbatch <= wide_wgs)) ? 1 : 0;
            // levels that hold a front formed against a CU's whole LDS (regroup_wide_top): wide whatever their population, 2 = every
            // front of the level is laid out against the larger cap
            // ... 3 = fronts formed against half a CU's LDS at a level that is not wide by its population: the 8-wavefront launch
            V.lds_mid_doubles = static_cast<int>(std::max<long long>(V.lds_doubles, (h->lds_limit / 8 - 160) / 2 - 8));
            for(int l = 0; l < V.n_top_levels; ++l)
            {
                int need = 0;
                for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k)
                    if(int const c = lds_class(S, so, S.top_list[k]); c != 0) need = (need == 2 || c == 2) ? 2 : 3;
                if(need != 0) V.top_wide[l] = (need == 2 || V.top_wide[l] != 0) ? 2 : 3;
            }
            V.lds_top_doubles = chain_lds ? static_cast<int>(std::max<long long>(V.lds_doubles, whole_cu)) : V.lds_doubles;
            if(!pe::build_assembly_lists(S, V.lds_slot, V.lds_doubles - 2, chain_lds ? V.top_wide : nullptr, V.lds_top_doubles - 2, V.lds_mid_doubles - 2))
                return fail(h, PE_HIP_ERR_INTERNAL, "symbolic analysis: " + S.error);
        }
        HIPCHK(h, pool.upload(V.f_mode, S.f_mode));
        HIPCHK(h, pool.upload(V.f_keep, S.f_keep));
        // launch-shape knobs of this engine travel in the view (pe_kernels.hip reads no environment)
        V.mid_top_limit = knob(h, "MID_TOP", 512);
        V.ew_grid = std::max(0, knob(h, "EW_GRID", 0));
        V.quad_lds_pad = std::max(0, knob(h, "QUAD_LDS", 0));
        V.top_run_any_class = knob(h, "TEST_OLD_TOP_RUNS", 0) != 0 ? 1 : 0;
        // LDS guard: what every front's layout occupies, checked here against the launch that will run it (a host loop over the
        // fronts) and once more on the device against the LDS the launch really received (k_m2_factor_top*: flag bit 3)
        {
            std::vector<int> need(static_cast<size_t>(S.nfronts), 0);
            for(int s = 0; s < S.nfronts; ++s) need[static_cast<size_t>(s)] = static_cast<int>(pe::front_lds_need(S.f_mode[s], S.f_p[s], S.f_u[s]));
            if(int const rc = check_lds_plan(h, S, V, need, batch); rc != PE_HIP_OK) return rc;
            HIPCHK(h, pool.upload(V.f_need, need));
        }
        HIPCHK(h, pool.upload(V.gl_ptr, S.gl_ptr));
        HIPCHK(h, pool.upload(V.gl_rptr, S.gl_rptr));
        HIPCHK(h, pool.upload(V.gl_sptr, S.gl_sptr));
        HIPCHK(h, pool.upload(V.gl_dst, S.gl_dst));
        HIPCHK(h, pool.upload(V.gl_cnt, S.gl_cnt));
        HIPCHK(h, pool.upload(V.gl_src, S.gl_src));
        HIPCHK(h, pool.alloc(V.zero, 1));
        // lane-group kernel of the wave fronts (pe_quad.hpp): its tables; V.q_list / V.n_quads follow the `active` mask (upload_active)
        V.quad = 0;
        V.quad_back = 0;
        V.n_mid = 0;
        if(S.quad)
        {
            HIPCHK(h, pool.upload(V.q_prog, S.q_prog));
            HIPCHK(h, pool.upload(V.q_lists, S.q_lists));
            HIPCHK(h, pool.upload(V.q_lane, S.q_lane));
            HIPCHK(h, pool.upload(V.q_bprog, S.q_bprog));
            HIPCHK(h, pool.upload(V.q2_prog, S.q2_prog));
            HIPCHK(h, pool.upload(V.q2_lists, S.q2_lists));
            HIPCHK(h, pool.upload(V.q2_lane, S.q2_lane));
            HIPCHK(h, pool.upload(V.f_kind, S.f_kind));
            HIPCHK(h, pool.upload(V.f_quad, S.f_quad));
            V.n_mid = S.n_mid;
            V.q_zero_off = S.q_zero_off;
            // LDS stack of a quad: slot 0 of an instance's stack holds a zero, the stride puts the four instances on different banks
            V.q_lds_stride = S.q_lds_doubles > 0 ? (S.q_lds_doubles + 1 + 31) / 32 * 32 + 8 : 0;
            // a quad addresses its four instances by 32-bit byte offsets from the first one: every per-instance array must leave room
            // for at least one instance inside 4 GiB (else the wave fronts fall back to the per-instance path of factor_part)
            long long const stride = 8 * std::max({static_cast<long long>(S.nnzA), V.factor_doubles, V.arena_doubles, static_cast<long long>(S.n)});
            V.quad = stride < (1ll << 31) ? (knob(h, "QUAD", 1) | 1) : 0;
            // the same fronts' backward pass on the lane-group kernel -- from 192 instances on: at 128 (one of its wavefronts per SIMD) the
            // per-instance backward pass of the parts is 1.3 % faster per iteration, at 256 they are equal (profiles/r03_ab_runs.log ab22)
            // (round 4: with the ancestors' unknowns fetched by two loads + row broadcasts the lane-group backward kernel wins at 128 instances
            //  too -- 1.378 against 1.403 ms per iteration, profiles/r04_ab_runs.log -- : on wherever the lane-group kernel is)
            V.quad_back = (V.quad && knob(h, "QUAD_BACK", 1) != 0) ? 1 : 0;
        }
        HIPCHK(h, pool.alloc(V.factor, static_cast<size_t>(V.factor_doubles) * batch));
        HIPCHK(h, pool.alloc(V.arena, static_cast<size_t>(V.arena_doubles) * batch));
        return PE_HIP_OK;
    }

    // launch geometry -> symbolic limits: 8 wavefronts per workgroup, panels / wave slots carved from the LDS limit
    int env_int0(char const* name, int def)
    {
        char const* v = std::getenv(name);
        return v && *v ? std::atoi(v) : def;
    }
    // a tuning knob of THIS engine: pe_hip_set_knob() first, then the process environment (PHY_ENGINE_HIP_<name>), then the default
    int knob(pe_hip_engine const* h, char const* name, int def)
    {
        if(h)
            if(auto it = h->knobs.find(name); it != h->knobs.end()) return it->second;
        return env_int0((std::string("PHY_ENGINE_HIP_") + name).c_str(), def);
    }

    pe::SymbolicOptions symbolic_options(pe_hip_engine const* h, int batch_in, int rows, int panel_reserve, int force_resident)
    {
        int const batch = geometry_batch(h, batch_in);
        pe::SymbolicOptions so{};
        // Workgroup geometry by batch size (measured on MI355X, profiles/ and scripts/sweep_split_*.sh).
        // Large circuits run the split schedule (one launch per phase): from ~100 instances on, 256-thread workgroups at four per
        // CU with every instance cut into 4 (8, 16) parts -- >= 1024 workgroups for the low-register kernels; fewer instances keep one
        // big workgroup per CU and more parts.  Small circuits run the resident kernel: geometry by the batch alone.
        bool const large = rows >= 3000 && knob(h, "SPLIT", -1) != 0;
        bool const four_per_cu = large ? batch >= 96 : batch >= 768;
        if(four_per_cu)
        {
            so.n_waves = 4;
            so.wave_m = 45;       // one wavefront takes fronts up to order 45: whole in its 10 KB slot up to 35, the larger ones in the
            so.wave_slot = (pe::pe_ld(35) + 1) * 35;  // panel layout (their panels fit the same slot) -- a third of what used to be
                                  // cooperative fronts leaves the barrier-synchronised phase (-1.3 % per iteration at 1 024 instances, -1.5 % at 128)
            so.wave_p = 16;       // (small staged blocks: the backward kernel of the split schedule then fits 8 workgroups per CU)
            so.absorb_m = 35;
            so.max_pivots = 32;
        }
        else if(batch >= 384)
        {
            so.n_waves = 8;
            so.wave_m = 32;
            so.max_pivots = 32;
        }
        else
        {
            so.n_waves = 8;
            so.wave_m = 56;
            so.wave_p = 20;
            so.max_pivots = 48;
        }
        if(large)
        {
            // (re-swept after the larger wave-front class: 128 instances 16 parts 1.65 ms per iteration against 1.68 with 8 and 1.77 with 12;
            //  256 instances 8 parts 2.74 against 2.81 with 4; 512 and 1 024 instances stay at 4)
            // (after the top fronts were regrouped against their launch's LDS the top got cheaper: 512 instances now prefer 8 parts, 4.47 against
            //  4.55 ms per iteration; 1 024 stay at 4 -- 8 parts take 0.65 ms off the dominant pair and put 0.6 ms on the top: profiles/sweep_r03_b128.log)
            so.n_parts = batch >= 768 ? 4 : (batch >= 192 ? 8 : (batch >= 96 ? 16 : std::clamp(256 / std::max(1, batch), 1, 48)));
            so.part_cut = 1.0;
            so.nd_leaf = 10;  // finer dissection: fewer, better-shaped fronts on big meshes (-3.6 % per iteration on M10k, profiles/sweep_r02_leaf.log);
                              // small circuits keep 24 (their whole graph is one minimum-degree leaf, as validated by every golden)
        }
        // tuning knobs (PHY_ENGINE_HIP_* family, SURVEY.md 5 "Config / flags")
        auto env_int = [h](char const* name, int def) { return knob(h, name + 15, def); };  // (name without its PHY_ENGINE_HIP_ prefix)
        // the wave fronts of a large sweep run four instances per wavefront on the lane-group kernel (pe_quad.hpp): fronts of order
        // <= 32 with <= 16 pivots; larger ones stay with the cooperative phase
        so.quad = (four_per_cu && large && env_int("PHY_ENGINE_HIP_QUAD", 1) != 0) ? 1 : 0;
        if(so.quad)
        {
            // Amalgamation re-swept WITH the lane-group kernel (profiles/sweep_r03_amalgamation.log): a front that absorption grows past
            // order 32 drops out of the quad class, and with it every ancestor inside its wave subtree.  Absorbing only up to order 32 and
            // forcing last-child merges only up to 4 pivots (8 before) leaves 933 of 995 wave fronts to the lane-group kernel on M10k (706
            // of 814 before), 5 % fewer stored factor entries: launch pair -2.3 %, steps/s +1.8 % at 1 024 instances (three interleaved runs).
            // (128 instances -- 16 parts -- do not gain: 31.8 k against 32.0 k steps/s; 256: +2 %.  From 192 instances on.)
            if(batch >= 192)
            {
                so.absorb_m = 32;
                so.relax_small = 4;
            }
            // (the wave-front class keeps the geometry above: wave fronts that do not qualify for the lane-group kernel -- order 33..45, or
            //  above one -- stay with the per-instance wave phase, which is cheaper for them than the cooperative phase)
            // The MID launch (order 33..64 on the lane-group scheme, pe_quad.hpp) measured slower than the cooperative phase and is no
            // longer selectable: since the hybrid wave phase a MID front may sit above a per-instance wave front, which the parts kernel
            // factors AFTER the MID launch -- the MID front then assembles last iteration's update matrix (Newton still converges, one
            // iteration later: seen as 3.14 instead of 2.48 iterations per step on M10k).  Re-measured on the round's final tree before
            // retiring it (scripts/r3_mid.sh): it takes 0.69 ms out of k_m2_factor_parts and costs 1.95 ms.  PHY_ENGINE_HIP_MID is ignored.
            so.quad_mid = 0;
            // update matrices whose parent follows in the same list could stay on an LDS stack: 8 wavefronts per CU (two per SIMD at this
            // kernel's register count) share the 160 KB -> 600 doubles per instance of a quad
            // (measured slower than the arena for the fronts it applies to, pe_quad.hpp PE_QUAD_LDS_STACK: off unless asked for)
            so.quad_lds_doubles = std::max(0, env_int("PHY_ENGINE_HIP_QUAD_STACK", 0));
        }
        so.n_waves = std::clamp(env_int("PHY_ENGINE_HIP_WAVES", so.n_waves), 1, PE_THREADS / 64);
        so.wave_m = std::max(1, env_int("PHY_ENGINE_HIP_WAVE_M", so.wave_m));
        so.wave_p = std::max(1, env_int("PHY_ENGINE_HIP_WAVE_P", so.wave_p));
        so.absorb_m = std::max(1, env_int("PHY_ENGINE_HIP_ABSORB_M", so.absorb_m));
        so.nd_leaf = std::max(2, env_int("PHY_ENGINE_HIP_ND_LEAF", so.nd_leaf));
        so.relax_zero_frac = 0.01 * std::clamp(env_int("PHY_ENGINE_HIP_RELAX_X100", static_cast<int>(so.relax_zero_frac * 100.0 + 0.5)), 0, 100);
        so.relax_small = std::max(1, env_int("PHY_ENGINE_HIP_RELAX_SMALL", so.relax_small));
        so.cut_factor = 0.1 * std::max(1, env_int("PHY_ENGINE_HIP_CUT_X10", static_cast<int>(so.cut_factor * 10.0)));
        so.max_pivots = std::clamp(env_int("PHY_ENGINE_HIP_MAX_PIVOTS", so.max_pivots), 1, 64);  // (the triangular solves keep one pivot per lane)
        so.n_parts = std::clamp(env_int("PHY_ENGINE_HIP_PARTS", so.n_parts), 1, 64);
        if(env_int("PHY_ENGINE_HIP_SPLIT", -1) == 0) so.n_parts = 1;  // the resident kernel handles one part per instance
        so.part_cut = 0.1 * std::max(1, env_int("PHY_ENGINE_HIP_PART_CUT_X10", static_cast<int>(so.part_cut * 10.0)));
        so.wave_p = std::min(so.wave_p, so.max_pivots);
        // LDS share of one workgroup: the 128-VGPR kernels keep 16 wavefronts per CU resident (16 / n_waves workgroups)
        bool const shared_cu = four_per_cu || batch >= 384;  // 128-VGPR kernels, 16 wavefronts per CU
        int const resident = force_resident > 0 ? force_resident : std::clamp(env_int("PHY_ENGINE_HIP_RESIDENT", shared_cu ? std::max(1, 16 / so.n_waves) : 1), 1, 8);
        so.shared_cu = (resident > 1) ? 1 : 0;
        long long const lds_doubles = (h->lds_limit / 8 - 160) / resident - 8;  // minus the static LDS of __syncthreads_or & co.
        // a wavefront's slot holds whole fronts of order <= wave_m (pe_front.hpp, FULL mode)
        if(int const ws = env_int("PHY_ENGINE_HIP_WAVE_SLOT", 0); ws > 0) so.wave_slot = ws;  // tuning knob: slot smaller than wave_m needs whole
        if(so.wave_slot > 0 && so.n_waves * so.wave_slot > lds_doubles) so.wave_slot = 0;
        if(so.wave_slot == 0)
            while(static_cast<long long>(so.n_waves) * so.wave_m * (pe::pe_ld(so.wave_m) + 1) > lds_doubles && so.wave_m > 8) --so.wave_m;
        so.wave_p = std::min(so.wave_p, so.wave_m);
        so.absorb_m = std::min(so.absorb_m, so.wave_m);
        // large (panel-mode) fronts keep room behind the panels for the right-hand-side column (m doubles) and their
        // children's staged inverse maps
        so.panel_doubles = std::max<long long>(lds_doubles - panel_reserve, lds_doubles / 2);
        so.panel_reserve = lds_doubles - so.panel_doubles;
        return so;
    }

    // Second pass of the analysis for the split schedule: the top levels that run one 16-wavefront workgroup per front anyway (one
    // workgroup per CU: k_m2_factor_top_wide -- every top level of a single circuit, the levels with fronts x instances <= CUs + 25 %
    // of a sweep) were cut into links that fit the ORDINARY workgroup's LDS share and pivot limit: at 128 instances the root
    // separator of the 10k mesh and its two children are eleven levels of links with 3..32 pivots and orders 86..150, each link a
    // front's fixed work and a trip of its Schur block through HBM.  With the unknowns of those levels marked, the analysis forms
    // their fronts against a CU's whole LDS and 64 pivots, in links of equal length; the factorisation is the same elimination in
    // fewer, larger steps.  Levels of up to 2 048 workgroups get HALF a CU's LDS the same way (the 8-wavefront launch
    // k_m2_factor_top_mid, two workgroups per CU).  Kept only if every such front ends up at a top level (those levels are then forced
    // onto the launch that has the LDS, upload_symbolic: V.top_wide 2 / 3); otherwise the first pass stands.  Knob TOP_BIG=0: first
    // pass only.  Measured on M10k (profiles/r03_ab_runs.log ab16-ab20): 1.594 -> 1.412 ms per Newton iteration at 128 instances,
    // 2.591 -> 2.354 at 256, 4.62 -> 4.52 at 512, 0.516 -> 0.480 for the single circuit; 1 024 instances within the noise (-0.4 %).
    void regroup_wide_top(pe_hip_engine* h, int batch, int geometry_rows, int n, int const* rp, int const* ci, double const* vals, pe::Symbolic& S,
                          pe::SymbolicOptions& so)
    {
        so.big_unknowns = nullptr;
        bool const split = geometry_rows >= 3000 && knob(h, "SPLIT", -1) != 0 && S.n_parts > 1;
        if(!split || knob(h, "TOP_BIG", 1) == 0 || knob(h, "WIDE_TOP", 1) == 0 || knob(h, "TOP_CHAIN_LDS", 1) == 0) return;
        int const gb = geometry_batch(h, batch), levels = static_cast<int>(S.top_ptr.size()) - 1;
        std::vector<char> big(static_cast<size_t>(n), 0);
        int marked = 0;
        bool const half_knob = knob(h, "TOP_BIG", 1) != 2;  // developer knob: 2 = whole-CU levels only
        // (half-CU levels: up to 2 048 workgroups -- four rounds of the 512 that fit.  Measured, profiles/r03_ab_runs.log: 1 280 against 640
        //  (ab20) 1.412 / 1.456 ms per iteration at 128 instances, 2.354 / 2.504 at 256, 4.52 / 4.56 at 512; 2 048 against 1 280 (ab23) takes in
        //  the two-front levels of 1 024 instances: 2.88 / 3.00 ms per iteration outside the dominant launch pair, nothing changes below)
        long long const half_wgs = std::max(0, knob(h, "TOP_HALF_WGS", 2048));
        for(int l = 0; l < levels; ++l)
        {
            long long const wgs = static_cast<long long>(S.top_ptr[l + 1] - S.top_ptr[l]) * gb;
            // whole CU: the rule of V.top_wide; half a CU: the 8-wavefront launch, two workgroups per CU
            int const cls = (!so.shared_cu || wgs <= std::max(0, knob(h, "TOP_WIDE_WGS", 320))) ? 1 : ((half_knob && wgs <= half_wgs) ? 2 : 0);
            if(cls == 0) continue;
            for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k)
            {
                int const s = S.top_list[k];
                for(int c = S.f_col0[s]; c < S.f_col0[s] + S.f_p[s]; ++c) big[static_cast<size_t>(S.col_src[c])] = static_cast<char>(cls);
                ++marked;
            }
        }
        if(marked < 2) return;  // (nothing to merge)
        pe::SymbolicOptions so2 = so;
        so2.big_unknowns = &big;
        // (64 = one pivot per lane of the triangular solves is the hard limit; measured on M10k, profiles/r03_ab_runs.log ab18: 64 against 48
        //  is 1.465 / 1.484 ms per iteration at 128 instances, 0.480 / 0.519 for the single circuit, equal at 256)
        so2.max_pivots_top = std::clamp(knob(h, "TOP_MAX_PIVOTS", 64), 1, 64);
        long long const whole_cu = h->lds_limit / 8 - 160 - 8;
        long long const behind = std::max<long long>(so.panel_reserve, S.max_m + 8 + 512);  // (right-hand-side column + staged child maps)
        so2.panel_doubles_top = whole_cu - 2 - behind;
        so2.panel_doubles_mid = so.shared_cu ? (h->lds_limit / 8 - 160) / 2 - 8 - 2 - behind : 0;
        pe::Symbolic S2;
        if(!pe::analyze(n, rp, ci, vals, so2, S2)) return;
        if(static_cast<int>(S2.top_ptr.size()) - 1 > 64 || S2.max_m + 8 > so.panel_reserve || S2.n_parts != S.n_parts) return;
        for(int s = 0; s < S2.nfronts; ++s)
            if(lds_class(S2, so2, s) != 0 && S2.f_kind[s] != 2) return;  // a front of a part would need a CU's LDS: keep the first pass
        S = std::move(S2);
        so.max_pivots_top = so2.max_pivots_top;
        so.panel_doubles_top = so2.panel_doubles_top;
        so.panel_doubles_mid = so2.panel_doubles_mid;
    }

    // Symbolic analysis + the LDS-fit escalation every caller needs (resident circuit AND the solve_csr_real seam): a front's
    // right-hand-side column (m doubles) must fit the reserve behind its panels, and the top of the tree the launch table.
    // (1) a larger reserve; (2) the whole LDS of a CU for one workgroup; else give up loudly.  `geometry_rows`: row count the
    // launch geometry is chosen by (0: the resident single-workgroup kernel, as the solver seam runs).
    int analyze_fitting(pe_hip_engine* h, int batch, int geometry_rows, int n, int const* rp, int const* ci, double const* vals, pe::Symbolic& S,
                        pe::SymbolicOptions& so)
    {
        so = symbolic_options(h, batch, geometry_rows);
        for(int attempt = 0;; ++attempt)
        {
            if(!pe::analyze(n, rp, ci, vals, so, S))
                return fail(h, S.structurally_singular ? PE_HIP_ERR_SINGULAR : PE_HIP_ERR_INTERNAL, "symbolic analysis: " + S.error);
            bool const too_deep = static_cast<int>(S.top_ptr.size()) - 1 > 64;
            bool const fits = S.max_m + 8 <= so.panel_reserve;
            if(fits && !too_deep)
            {
                regroup_wide_top(h, batch, geometry_rows, n, rp, ci, vals, S, so);
                return PE_HIP_OK;
            }
            if(attempt == 2) return fail(h, PE_HIP_ERR_INTERNAL, "symbolic analysis: a front of order " + std::to_string(S.max_m) + " does not fit the LDS of a CU");
            so = symbolic_options(h, batch, geometry_rows, std::max(384, S.max_m + 72), attempt == 1 ? 1 : 0);
            if(too_deep) so.n_parts = 1;
        }
    }

    int ensure_symbolic(pe_hip_engine* h, bool tr, double dt)
    {
        int const cls = tr ? 1 : 0;
        // the static pivot order was matched on representative values at ONE dt (capacitor / inductor companions scale with 1/dt):
        // a time step more than a decade away from it gets a fresh analysis, like a change of class
        bool const dt_moved = tr && h->sym_dt > 0.0 && dt > 0.0 && (dt > 10.0 * h->sym_dt || dt < 0.1 * h->sym_dt) && h->sym_values_override.empty();
        if(h->sym_class == cls && !dt_moved) return PE_HIP_OK;
        if(tr) h->sym_dt = dt;
        auto const t0 = clk::now();
        std::vector<double> av;
        if(!h->sym_values_override.empty()) av = h->sym_values_override;
        else
        {
            pe::estimate_values(h->hc, tr, dt, h->opt.g_min, r_open_of(h), av);
            // test knob: a pivot matching that cannot see magnitudes (every structural entry weighs 1) -- the deliberately bad
            // static order the residual safety net is tested against; a re-match on an instance's own values is not affected
            if(char const* k = std::getenv("PHY_ENGINE_HIP_TEST_BLIND_MATCH"); k && *k == '1') std::fill(av.begin(), av.end(), 1.0);
        }
        pe::SymbolicOptions so{};
        {
            int const rc = analyze_fitting(h, h->hc.batch, h->hc.rows, h->hc.rows, h->hc.rp.data(), h->hc.ci.data(), av.data(), h->sym, so);
            if(rc != PE_HIP_OK)
            {
                h->sym_class = -1;
                return rc;
            }
        }
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
        {
            auto const& S = h->sym;
            int nk[4]{};
            for(int s = 0; s < S.nfronts; ++s) ++nk[S.f_kind[s]];
            std::fprintf(stderr, "[pe_hip] schedule: %d fronts (%d wave, %d cooperative, %d top, %d mid), %d parts, %d top levels\n", S.nfronts, nk[0], nk[1],
                         nk[2], nk[3], S.n_parts, static_cast<int>(S.top_ptr.size()) - 1);
            if(S.quad)
                std::fprintf(stderr, "[pe_hip]   lane-group kernel: LDS stack %d doubles per instance holds %lld of %lld update-matrix doubles of the wave fronts\n",
                             S.q_lds_doubles, S.q_lds_kept, S.q_lds_total);
            if(S.quad)
            {
                std::fprintf(stderr, "[pe_hip]   wave-front lists (fronts):");
                for(size_t L = 0; 2 * L + 1 < S.q_lists.size(); ++L) std::fprintf(stderr, " %d", S.q_lists[2 * L + 1]);
                std::fprintf(stderr, "\n[pe_hip]   MID lists (fronts):");
                for(size_t L = 0; 2 * L + 1 < S.q2_lists.size(); ++L) std::fprintf(stderr, " %d", S.q2_lists[2 * L + 1]);
                std::fprintf(stderr, "\n");
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 3 && dump[1] == '4')
                        std::fprintf(stderr, "[pe_hip]   mid front %d: %dx%d children %d parent %d(kind %d)\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s],
                                     S.f_parent[s], S.f_parent[s] >= 0 ? S.f_kind[S.f_parent[s]] : -1);
            }
            for(int kind = 0; kind < 2; ++kind)
            {
                long long cnt[3]{}, su2[3]{}, spanel[3]{};
                long long const cap = kind == 0 ? std::max<long long>(S.wave_panel_doubles, 1) : so.panel_doubles;
                for(int s = 0; s < S.nfronts; ++s)
                {
                    if(S.f_kind[s] != kind) continue;
                    long long const p = S.f_p[s], u = S.f_u[s], m = p + u;
                    int const nch = S.f_child_ptr[s + 1] - S.f_child_ptr[s];
                    int const mode = m * m <= cap ? 0 : (nch == 1 && S.f_u[S.f_child[S.f_child_ptr[s]]] == m ? 2 : 1);
                    ++cnt[mode];
                    su2[mode] += u * u;
                    spanel[mode] += m * p + p * u;
                }
                std::fprintf(stderr, "[pe_hip]   %s fronts (cap ~%lld doubles): whole %lld (S %lld, panels %lld) | panel+pull %lld (S %lld, panels %lld) | chain link %lld (S %lld, panels %lld)\n",
                             kind == 0 ? "wave" : "cooperative", cap, cnt[0], su2[0], spanel[0], cnt[1], su2[1], spanel[1], cnt[2], su2[2], spanel[2]);
            }
            if(dump[1] == '3')
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 0) std::fprintf(stderr, "[pe_hip]   wave front %d: %dx%d children %d\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s]);
            if(dump[1] == '2')
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 1) std::fprintf(stderr, "[pe_hip]   coop front %d: %dx%d children %d\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s]);
            for(std::size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
            {
                std::fprintf(stderr, "[pe_hip]   top level %zu:", l);
                for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) std::fprintf(stderr, " %dx%d", S.f_p[S.top_list[k]] + S.f_u[S.top_list[k]], S.f_p[S.top_list[k]]);
                std::fprintf(stderr, "\n");
            }
        }
        pe::m2_graphs_clear(h->graphs);  // (captured sequences hold the old view: its tables are about to be freed)
        h->sym_pool.release();
        int const rc = upload_symbolic(h, h->sym_pool, h->sym, so, h->V, h->hc.batch);
        if(rc != PE_HIP_OK) return rc;
        h->active_dev.clear();  // (the quad list behind the mask depends on this analysis' strides)
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
        {
            auto const& S = h->sym;
            std::fprintf(stderr, "[pe_hip]   top fronts, LDS layout (0 whole, 1 panels, 2 chain link, 3 chain link continued in LDS; * = 16-wavefront level, %d doubles):", h->V.lds_top_doubles);
            for(std::size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
                for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) std::fprintf(stderr, " %d%s", S.f_mode[S.top_list[k]], h->V.top_wide[l] ? "*" : "");
            std::fprintf(stderr, "\n");
        }
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
        {
            // panel-layout Schur tiles by the number of children they pull (front_factor: the first two are requested a tile ahead, the rest one by one)
            auto const& S = h->sym;
            long long hist[8]{}, tiles = 0;
            for(int s = 0; s < S.nfronts; ++s)
            {
                if(S.f_kind[s] == 0 || S.f_mode[s] != 1) continue;
                int const u = S.f_u[s], nt = (u + 15) / 16, ch0 = S.f_child_ptr[s], ch1 = S.f_child_ptr[s + 1];
                for(int tj = 0; tj < nt; ++tj)
                    for(int ti = 0; ti < nt; ++ti)
                    {
                        int n = 0;
                        for(int c = ch0; c < ch1; ++c)
                        {
                            unsigned const mk = S.f_bmask[static_cast<size_t>(c)];
                            n += static_cast<int>(((mk >> std::min(ti, 31)) & (mk >> std::min(tj, 31))) & 1u);
                        }
                        ++hist[std::min(n, 7)];
                        ++tiles;
                    }
            }
            std::fprintf(stderr, "[pe_hip]   panel-layout Schur tiles by children pulled (0..7+) of %lld:", tiles);
            for(long long v: hist) std::fprintf(stderr, " %lld", v);
            std::fprintf(stderr, "\n");
        }
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
            std::fprintf(stderr, "[pe_hip]   LDS plan (doubles): factor %d, solves %d, backward %d (wave slot %d = t %d + stage %d + 64 + stack %d), wave front slot %d\n", h->V.lds_doubles,
                         h->V.lds_solve_doubles, h->V.lds_solve_b_doubles, h->V.lds_sslot, h->V.wave_m, h->V.lds_wave_stage, h->V.lds_sslot - h->V.lds_bstack_off, h->V.lds_slot);
        {
            // The matrix values live in FRONT-ASSEMBLY order on the device: slot e of `aval` is the e-th assembled entry
            // (asm_slot is a permutation of the CSR slots), so a front reads its own entries of A as one contiguous run with no
            // index indirection.  The contribution lists of the stamp are permuted to match.
            auto const& S = h->sym;
            auto const& hc = h->hc;
            size_t const nnz = hc.ci.size();
            std::vector<int> ptr2(nnz + 1, 0), src2;
            src2.reserve(hc.a_src.size());
            for(size_t e = 0; e < nnz; ++e)
            {
                int const slot = S.asm_slot[e];
                src2.insert(src2.end(), hc.a_src.begin() + hc.a_ptr[slot], hc.a_src.begin() + hc.a_ptr[slot + 1]);
                ptr2[e + 1] = static_cast<int>(src2.size());
            }
            if(src2.empty()) src2.push_back(0);
            HIPCHK(h, h->sym_pool.upload(h->V.a_ptr, ptr2));
            HIPCHK(h, h->sym_pool.upload(h->V.a_src, src2));
            // x-dependent slots / rows (pe_front.hpp stamp_dynamic_chunk): Newton iterations after the first stamp only these
            h->V.dyn_a = h->V.dyn_b = nullptr;
            h->V.n_dyn_a = h->V.n_dyn_b = 0;
            if(hc.nonlinear)
            {
                std::vector<char> const dyn = pe::dynamic_dv_mask(hc);
                std::vector<int> da, db;
                for(size_t e = 0; e < nnz; ++e)
                    for(int k = ptr2[e]; k < ptr2[e + 1]; ++k)
                        if(dyn[static_cast<size_t>(src2[k] >> 1)])
                        {
                            da.push_back(static_cast<int>(e));
                            break;
                        }
                for(int r = 0; r < hc.rows; ++r)
                    for(int k = hc.b_ptr[r]; k < hc.b_ptr[r + 1]; ++k)
                        if(dyn[static_cast<size_t>(hc.b_src[k] >> 1)])
                        {
                            db.push_back(r);
                            break;
                        }
                h->V.n_dyn_a = static_cast<int>(da.size());
                h->V.n_dyn_b = static_cast<int>(db.size());
                if(da.empty()) da.push_back(0);
                if(db.empty()) db.push_back(0);
                HIPCHK(h, h->sym_pool.upload(h->V.dyn_a, da));
                HIPCHK(h, h->sym_pool.upload(h->V.dyn_b, db));
                std::vector<unsigned char> rd(static_cast<size_t>(std::max(1, hc.rows)), 0);
                for(int k = 0; k < h->V.n_dyn_b; ++k) rd[static_cast<size_t>(db[static_cast<size_t>(k)])] = 1;
                HIPCHK(h, h->sym_pool.upload(h->V.row_dyn, rd));
            }
            h->V.asm_slot = nullptr;  // identity (pe_front.hpp front_factor); the solve_csr_real seam keeps CSR order + the map
            std::vector<int> slot_e(nnz, 0);  // CSR slot -> position in aval (residual check walks A row by row in original order)
            for(size_t e = 0; e < nnz; ++e) slot_e[S.asm_slot[e]] = static_cast<int>(e);
            HIPCHK(h, h->sym_pool.upload(h->V.slot_e, slot_e));
        }
        h->sym_class = cls;
        h->fact_valid = false;
        h->a_static.clear();
        h->analyze_ms = ms_since(t0);
        return PE_HIP_OK;
    }

}  // namespace pe_eng
