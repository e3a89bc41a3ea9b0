// pe_symbolic.cpp -- see pe_symbolic.hpp.
//
// Index conventions.  A is the n x n MNA matrix in CSR.  The factorisation works on
//     A''[k][l] = A[ row_src[k] ][ col_src[l] ]
// i.e. permuted equation k is original equation row_src[k] and permuted unknown l is original unknown
// col_src[l].  row_src = rmatch o q and col_src = q, where rmatch is the row matching (equation chosen as
// pivot row for each unknown) and q the fill-reducing symmetric ordering of pattern(A' + A'^T),
// A'[j][:] = A[rmatch[j]][:].
#include "pe_symbolic.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace pe
{
    namespace
    {
        using ivec = std::vector<int>;

        // ------------------------------------------------------------------------------------
        // 1. row matching: unknown j <- equation rmatch[j], maximising (greedily) the scaled pivot size
        // ------------------------------------------------------------------------------------
        bool match_rows(int n, int const* rp, int const* ci, double const* vals, double diag_rel, ivec& rmatch, int& swaps)
        {
            rmatch.assign(n, -1);
            ivec colof(n, -1);  // equation i -> unknown it pivots
            std::vector<double> rowmax(n, 0.0);
            for(int i = 0; i < n; ++i)
                for(int e = rp[i]; e < rp[i + 1]; ++e) rowmax[i] = std::max(rowmax[i], vals ? std::fabs(vals[e]) : 1.0);
            // pass 1: keep the diagonal wherever it is usable
            for(int i = 0; i < n; ++i)
            {
                for(int e = rp[i]; e < rp[i + 1]; ++e)
                {
                    if(ci[e] != i) continue;
                    double const w = vals ? std::fabs(vals[e]) : 1.0;
                    if(w > 0.0 && w >= diag_rel * rowmax[i])
                    {
                        rmatch[i] = i;
                        colof[i] = i;
                    }
                }
            }
            // pass 2: augmenting paths for the remaining equations, heaviest (row-scaled) candidates first
            std::vector<ivec> cand(n);
            auto build_cand = [&](int i)
            {
                if(!cand[i].empty()) return;
                std::vector<std::pair<double, int>> t;
                for(int e = rp[i]; e < rp[i + 1]; ++e)
                {
                    double const w = vals ? std::fabs(vals[e]) : 1.0;
                    if(w > 0.0) t.push_back({-w, ci[e]});
                }
                std::sort(t.begin(), t.end());
                for(auto const& [w, c]: t) cand[i].push_back(c);
            };
            ivec visited(n, -1), it(n, 0), stack_row, stack_col;
            for(int r0 = 0; r0 < n; ++r0)
            {
                if(colof[r0] >= 0) continue;
                // iterative DFS over alternating paths: equation -> candidate unknown -> equation currently holding it
                stack_row.assign(1, r0);
                stack_col.clear();
                build_cand(r0);
                it[r0] = 0;
                bool found = false;
                while(!stack_row.empty())
                {
                    int const i = stack_row.back();
                    if(it[i] >= static_cast<int>(cand[i].size()))
                    {
                        stack_row.pop_back();
                        if(!stack_col.empty()) stack_col.pop_back();
                        continue;
                    }
                    int const c = cand[i][it[i]++];
                    if(visited[c] == r0) continue;
                    visited[c] = r0;
                    stack_col.push_back(c);
                    if(rmatch[c] < 0)
                    {
                        found = true;
                        break;
                    }
                    int const nxt = rmatch[c];
                    build_cand(nxt);
                    it[nxt] = 0;
                    stack_row.push_back(nxt);
                }
                if(!found) return false;  // structurally singular
                // flip along the path: stack_row[t] takes stack_col[t]
                for(size_t t = 0; t < stack_row.size(); ++t)
                {
                    rmatch[stack_col[t]] = stack_row[t];
                    colof[stack_row[t]] = stack_col[t];
                }
            }
            swaps = 0;
            for(int j = 0; j < n; ++j) swaps += rmatch[j] != j;
            return true;
        }

        // ------------------------------------------------------------------------------------
        // 2. adjacency of pattern(A' + A'^T), no diagonal
        // ------------------------------------------------------------------------------------
        void sym_graph(int n, int const* rp, int const* ci, ivec const& rmatch, ivec& gp, ivec& gi)
        {
            std::vector<ivec> adj(n);
            for(int j = 0; j < n; ++j)
            {
                int const r = rmatch[j];
                for(int e = rp[r]; e < rp[r + 1]; ++e)
                {
                    int const c = ci[e];
                    if(c == j) continue;
                    adj[j].push_back(c);
                    adj[c].push_back(j);
                }
            }
            gp.assign(n + 1, 0);
            for(int v = 0; v < n; ++v)
            {
                std::sort(adj[v].begin(), adj[v].end());
                adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end());
                gp[v + 1] = gp[v] + static_cast<int>(adj[v].size());
            }
            gi.resize(gp[n]);
            for(int v = 0; v < n; ++v) std::copy(adj[v].begin(), adj[v].end(), gi.begin() + gp[v]);
        }

        // ------------------------------------------------------------------------------------
        // 3. nested dissection by BFS level sets, minimum degree on the leaves
        // ------------------------------------------------------------------------------------
        struct Dissector
        {
            int n;
            ivec const& gp;
            ivec const& gi;
            int leaf;
            ivec tag, level, order;
            int next_tag{1};

            Dissector(int n_, ivec const& gp_, ivec const& gi_, int leaf_) : n(n_), gp(gp_), gi(gi_), leaf(leaf_), tag(n_, 0), level(n_, 0) { order.reserve(n_); }

            // exact minimum degree on the induced subgraph of V (|V| small)
            void md_leaf(ivec const& V)
            {
                int const k = static_cast<int>(V.size());
                if(k == 0) return;
                int const t = next_tag++;
                ivec local(k);
                for(int a = 0; a < k; ++a)
                {
                    tag[V[a]] = t;
                    level[V[a]] = a;  // reuse as local id
                }
                std::vector<std::vector<char>> adj(k, std::vector<char>(k, 0));
                for(int a = 0; a < k; ++a)
                    for(int e = gp[V[a]]; e < gp[V[a] + 1]; ++e)
                        if(tag[gi[e]] == t) adj[a][level[gi[e]]] = 1;
                std::vector<char> gone(k, 0);
                for(int step = 0; step < k; ++step)
                {
                    int best = -1, bestd = 1 << 30;
                    for(int a = 0; a < k; ++a)
                    {
                        if(gone[a]) continue;
                        int d = 0;
                        for(int b = 0; b < k; ++b) d += (!gone[b] && adj[a][b]);
                        if(d < bestd)
                        {
                            bestd = d;
                            best = a;
                        }
                    }
                    gone[best] = 1;
                    order.push_back(V[best]);
                    for(int a = 0; a < k; ++a)
                    {
                        if(gone[a] || !adj[best][a]) continue;
                        for(int b = 0; b < k; ++b)
                            if(!gone[b] && b != a && adj[best][b]) adj[a][b] = 1;
                    }
                }
            }

            // BFS inside the tagged set from `start`; fills `out` (visit order) and level[]; returns eccentricity
            int bfs(int start, int t, int mark, ivec& out, ivec& seen)
            {
                out.clear();
                out.push_back(start);
                seen[start] = mark;
                level[start] = 0;
                size_t head = 0;
                int ecc = 0;
                while(head < out.size())
                {
                    int const v = out[head++];
                    for(int e = gp[v]; e < gp[v + 1]; ++e)
                    {
                        int const w = gi[e];
                        if(tag[w] != t || seen[w] == mark) continue;
                        seen[w] = mark;
                        level[w] = level[v] + 1;
                        ecc = level[w];
                        out.push_back(w);
                    }
                }
                return ecc;
            }

            ivec seen;
            int next_mark{1};

            void run(ivec V)
            {
                if(seen.empty()) seen.assign(n, 0);
                if(static_cast<int>(V.size()) <= leaf)
                {
                    md_leaf(V);
                    return;
                }
                int const t = next_tag++;
                for(int v: V) tag[v] = t;
                ivec reach;
                int ecc = bfs(V[0], t, next_mark++, reach, seen);
                if(reach.size() < V.size())
                {
                    // disconnected: dissect the components independently
                    int const mk = next_mark - 1;
                    ivec rest;
                    for(int v: V)
                        if(seen[v] != mk) rest.push_back(v);
                    ivec comp = reach;
                    run(std::move(comp));
                    run(std::move(rest));
                    return;
                }
                // pseudo-peripheral start: restart from a minimum-degree vertex of the last level while the depth grows
                for(int rounds = 0; rounds < 6; ++rounds)
                {
                    int far = reach.back(), fard = 1 << 30;
                    for(size_t a = reach.size(); a-- > 0 && level[reach[a]] == ecc;)
                    {
                        int const d = gp[reach[a] + 1] - gp[reach[a]];
                        if(d < fard)
                        {
                            fard = d;
                            far = reach[a];
                        }
                    }
                    ivec r2;
                    int const e2 = bfs(far, t, next_mark++, r2, seen);
                    if(e2 > ecc)
                    {
                        ecc = e2;
                        reach.swap(r2);
                    }
                    else
                    {
                        // keep the deeper (current `reach`) structure: recompute its levels (bfs overwrote level[])
                        int const s0 = reach[0];
                        ecc = bfs(s0, t, next_mark++, reach, seen);
                        break;
                    }
                }
                if(ecc < 2)
                {
                    md_leaf(V);
                    return;
                }
                // level sizes -> most balanced interior level
                ivec cnt(ecc + 1, 0);
                for(int v: reach) ++cnt[level[v]];
                long long below = cnt[0];
                long long const total = static_cast<long long>(reach.size());
                int best_m = 1;
                double best_cost = 1e300;
                for(int m = 1; m < ecc; ++m)
                {
                    long long const above = total - below - cnt[m];
                    double const cost = static_cast<double>(std::max(below, above)) + 0.5 * cnt[m];
                    if(cost < best_cost)
                    {
                        best_cost = cost;
                        best_m = m;
                    }
                    below += cnt[m];
                }
                ivec A, B, S;
                for(int v: reach)
                {
                    if(level[v] < best_m) A.push_back(v);
                    else if(level[v] > best_m) B.push_back(v);
                    else
                    {
                        bool touches = false;
                        for(int e = gp[v]; e < gp[v + 1] && !touches; ++e)
                            touches = tag[gi[e]] == t && level[gi[e]] == best_m + 1;
                        (touches ? S : A).push_back(v);
                    }
                }
                run(std::move(A));
                run(std::move(B));
                for(int v: S) order.push_back(v);
            }
        };

        // ------------------------------------------------------------------------------------
        // 4. elimination tree / postorder / column structures
        // ------------------------------------------------------------------------------------
        void etree_of(int n, std::vector<ivec> const& lower /* lower[i] = {k < i : B[i][k] != 0} */, ivec& parent)
        {
            parent.assign(n, -1);
            ivec anc(n, -1);
            for(int i = 0; i < n; ++i)
            {
                for(int k0: lower[i])
                {
                    int k = k0;
                    while(k != -1 && k < i)
                    {
                        int const nxt = anc[k];
                        anc[k] = i;
                        if(nxt == -1)
                        {
                            parent[k] = i;
                            break;
                        }
                        k = nxt;
                    }
                }
            }
        }

        void col_structs(int n, std::vector<ivec> const& upper /* upper[j] = {i > j : B[i][j] != 0} sorted */, ivec const& parent, std::vector<ivec>& st)
        {
            st.assign(n, {});
            std::vector<ivec> kids(n);
            for(int j = 0; j < n; ++j)
                if(parent[j] >= 0) kids[parent[j]].push_back(j);
            ivec mark(n, -1);
            for(int j = 0; j < n; ++j)
            {
                ivec& s = st[j];
                for(int i: upper[j])
                    if(mark[i] != j)
                    {
                        mark[i] = j;
                        s.push_back(i);
                    }
                for(int c: kids[j])
                    for(int i: st[c])
                        if(i != j && mark[i] != j)
                        {
                            mark[i] = j;
                            s.push_back(i);
                        }
                std::sort(s.begin(), s.end());
            }
        }
    }  // namespace

    void build_quad_plan(Symbolic& S, int lds_doubles);

    bool analyze(int n, int const* rp, int const* ci, double const* vals, SymbolicOptions const& opt, Symbolic& S)
    {
        S = Symbolic{};
        S.n = n;
        S.nnzA = rp[n];
        if(n == 0) return true;

        ivec rmatch;
        if(!match_rows(n, rp, ci, vals, opt.match_diag_rel, rmatch, S.n_row_swaps))
        {
            S.structurally_singular = true;
            S.error = "structurally singular matrix (no zero-free diagonal exists)";
            return false;
        }

        ivec gp, gi;
        sym_graph(n, rp, ci, rmatch, gp, gi);

        ivec q;
        {
            Dissector D(n, gp, gi, std::max(2, opt.nd_leaf));
            ivec all(n);
            std::iota(all.begin(), all.end(), 0);
            D.run(std::move(all));
            q = std::move(D.order);
        }
        if(static_cast<int>(q.size()) != n)
        {
            S.error = "internal: ordering lost vertices";
            return false;
        }

        // ---------------- elimination tree + column structures of an ordering (optionally re-postordered)
        std::vector<ivec> st;
        ivec parent;
        auto compute_tree = [&](ivec& ord, bool repostorder)
        {
            for(int pass = 0; pass < 2; ++pass)
            {
                ivec qinv(n);
                for(int k = 0; k < n; ++k) qinv[ord[k]] = k;
                std::vector<ivec> lower(n), upper(n);
                for(int k = 0; k < n; ++k)
                {
                    int const v = ord[k];
                    for(int e = gp[v]; e < gp[v + 1]; ++e)
                    {
                        int const l = qinv[gi[e]];
                        if(l < k) lower[k].push_back(l);
                        else upper[k].push_back(l);
                    }
                    std::sort(upper[k].begin(), upper[k].end());
                }
                etree_of(n, lower, parent);
                col_structs(n, upper, parent, st);
                if(pass == 1 || !repostorder) break;
                // postorder, children sorted by structure size ascending (largest last => mergeable with the parent)
                std::vector<ivec> kids(n);
                ivec roots;
                for(int j = 0; j < n; ++j) (parent[j] >= 0 ? kids[parent[j]] : roots).push_back(j);
                auto by_size = [&](int a, int b) { return st[a].size() != st[b].size() ? st[a].size() < st[b].size() : a < b; };
                for(auto& kv: kids) std::sort(kv.begin(), kv.end(), by_size);
                ivec post;
                post.reserve(n);
                ivec stack, itx(n, 0);
                for(int r: roots)
                {
                    stack.push_back(r);
                    while(!stack.empty())
                    {
                        int const v = stack.back();
                        if(itx[v] < static_cast<int>(kids[v].size())) stack.push_back(kids[v][itx[v]++]);
                        else
                        {
                            post.push_back(v);
                            stack.pop_back();
                        }
                    }
                }
                ivec q2(n);
                for(int k = 0; k < n; ++k) q2[k] = ord[post[k]];
                ord.swap(q2);
            }
        };
        compute_tree(q, true);
        for(int j = 0; j < n; ++j) S.nnz_LU += 2LL * static_cast<long long>(st[j].size()) + 1;

        // ---------------- supernodes: fundamental runs, then relaxed merging of last children
        struct SN
        {
            int c0, c1;         // columns [c0, c1)
            long long zeros;    // explicit zeros in the L panel (same count in U)
        };
        std::vector<SN> sn;
        auto u_of = [&](SN const& s) { return static_cast<int>(st[s.c1 - 1].size()); };
        // (second pass: columns whose fronts own a CU's LDS -- SymbolicOptions::big_unknowns; q maps a column to its unknown)
        bool const has_big = opt.big_unknowns && static_cast<int>(opt.big_unknowns->size()) == n && opt.panel_doubles_top > 0;
        auto big_class = [&](int c0, int c1)  // 0: ordinary, 1: a CU's whole LDS, 2: half of it
        {
            if(!has_big) return 0;
            int cls = 1;
            for(int c = c0; c < c1; ++c)
            {
                int const k = (*opt.big_unknowns)[q[c]];
                if(k == 0) return 0;
                cls = std::max(cls, k);
            }
            return cls;
        };
        auto big_run = [&](int c0, int c1) { return big_class(c0, c1) != 0; };
        auto pivots_max = [&](int c0, int c1) { return big_run(c0, c1) ? std::max(opt.max_pivots, opt.max_pivots_top) : opt.max_pivots; };
        auto try_merge = [&](SN& P)
        {
            while(!sn.empty())
            {
                SN const& C = sn.back();
                if(C.c1 != P.c0) break;
                int const pc = parent[C.c1 - 1];
                if(pc < P.c0 || pc >= P.c1) break;  // not a child of P
                int const pC = C.c1 - C.c0, uC = u_of(C);
                int const pP = P.c1 - P.c0, mP = pP + u_of(P);
                int const np = pC + pP;
                if(np > pivots_max(C.c0, P.c1)) break;
                long long const z = C.zeros + P.zeros + static_cast<long long>(pC) * (mP - uC);
                double const panel = static_cast<double>(np) * (np + u_of(P)) - 0.5 * np * (np - 1);
                if(np > opt.relax_small && static_cast<double>(z) > opt.relax_zero_frac * panel) break;
                P.c0 = C.c0;
                P.zeros = z;
                sn.pop_back();
            }
        };
        {
            int j = 0;
            while(j < n)
            {
                SN P{j, j + 1, 0};
                while(P.c1 < n && parent[P.c1 - 1] == P.c1 && st[P.c1 - 1].size() == st[P.c1].size() + 1 && (big_run(P.c0, P.c1 + 1) || (P.c1 - P.c0) < opt.max_pivots)) ++P.c1;  // (second pass: a fundamental run of the top stays ONE group, cut into equal links below)
                j = P.c1;
                try_merge(P);
                sn.push_back(P);
            }
        }

        // ---------------- generalised absorption: a parent swallows ANY of its children while the merged front stays
        // small enough for one wavefront (m <= absorb_m).  The absorbed columns are treated as dense (explicit zeros),
        // which removes the tiny fronts whose cost on a GPU is pure latency.  Needs a re-permutation so that the
        // merged columns are contiguous; fill is unchanged (same elimination tree order within every subtree).
        std::vector<std::pair<int, int>> groups;  // forced partition [c0, c1) in the NEW order
        {
            int const ns = static_cast<int>(sn.size());
            ivec col2sn(n);
            for(int i = 0; i < ns; ++i)
                for(int c = sn[i].c0; c < sn[i].c1; ++c) col2sn[c] = i;
            ivec par(ns, -1), gp_(ns), gu(ns);
            std::vector<ivec> kids(ns), members(ns);
            for(int i = 0; i < ns; ++i)
            {
                auto const& r = st[sn[i].c1 - 1];
                if(!r.empty()) par[i] = col2sn[r[0]];
                gp_[i] = sn[i].c1 - sn[i].c0;
                gu[i] = static_cast<int>(r.size());
                members[i] = {i};
            }
            for(int i = 0; i < ns; ++i)
                if(par[i] >= 0) kids[par[i]].push_back(i);
            std::vector<char> absorbed(ns, 0);
            for(int P = 0; P < ns; ++P)
            {
                bool again = true;
                while(again)
                {
                    again = false;
                    ivec cand = kids[P];
                    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return gp_[a] != gp_[b] ? gp_[a] < gp_[b] : a < b; });
                    for(int C: cand)
                    {
                        if(gp_[P] + gp_[C] + gu[P] > opt.absorb_m || gp_[P] + gp_[C] > opt.wave_p) continue;
                        absorbed[C] = 1;
                        gp_[P] += gp_[C];
                        members[P].insert(members[P].end(), members[C].begin(), members[C].end());
                        auto& kp = kids[P];
                        kp.erase(std::find(kp.begin(), kp.end(), C));
                        for(int g: kids[C])
                        {
                            kp.push_back(g);
                            par[g] = P;
                        }
                        kids[C].clear();
                        again = true;  // grandchildren may now fit as well
                        break;
                    }
                }
            }
            // new column order: postorder DFS of the merged tree; a node emits its member supernodes in ascending order
            ivec q2;
            q2.reserve(n);
            ivec stack, itx(ns, 0);
            for(int r = 0; r < ns; ++r)
            {
                if(par[r] >= 0 || absorbed[r]) continue;
                stack.push_back(r);
                while(!stack.empty())
                {
                    int const v = stack.back();
                    if(itx[v] == 0) std::sort(kids[v].begin(), kids[v].end());
                    if(itx[v] < static_cast<int>(kids[v].size())) stack.push_back(kids[v][itx[v]++]);
                    else
                    {
                        std::sort(members[v].begin(), members[v].end());
                        int const c0 = static_cast<int>(q2.size());
                        for(int mbr: members[v])
                            for(int c = sn[mbr].c0; c < sn[mbr].c1; ++c) q2.push_back(q[c]);
                        groups.push_back({c0, static_cast<int>(q2.size())});
                        stack.pop_back();
                    }
                }
            }
            if(static_cast<int>(q2.size()) != n)
            {
                S.error = "internal: absorption lost columns";
                return false;
            }
            q.swap(q2);
        }
        compute_tree(q, false);

        // ---------------- fronts = forced groups, split to respect the wave / LDS-panel limits
        struct FR
        {
            int c0, c1, g1;  // pivots [c0, c1); the group ends at g1 (rows = [c1, g1) + structure of column g1-1)
        };
        std::vector<FR> fr;
        for(auto const& [c0, c1]: groups)
        {
            int const ug = static_cast<int>(st[c1 - 1].size());
            int a = c0;
            while(a < c1)
            {
                int const rest = c1 - a;
                int const mfull = rest + ug;
                int b;
                if(mfull <= opt.wave_m && !big_run(c0, c1)) b = a + std::min(rest, opt.wave_p);  // (a front of the top is never a wave front)
                else if(opt.quad_mid && mfull <= 64)
                    b = a + std::min(rest, 16);  // a candidate MID front of the lane-group kernels: <= 16 pivots (one row set)
                else
                {
                    int const cls = big_class(c0, c1);  // (second pass: a front of the top that owns a CU's LDS, or half of it)
                    bool const big = cls != 0;
                    long long const budget = std::max(opt.panel_doubles, cls == 1 ? opt.panel_doubles_top : (cls == 2 ? opt.panel_doubles_mid : 0));
                    int pbest = 0;
                    for(int pp = std::min(rest, big ? std::max(opt.max_pivots, opt.max_pivots_top) : opt.max_pivots); pp >= 1; --pp)
                    {
                        long long const uch = (rest - pp) + ug, mch = pp + uch;
                        if(static_cast<long long>(pe_ld(static_cast<int>(mch))) * pp + static_cast<long long>(pe_ld(pp)) * uch <= budget)
                        {
                            pbest = pp;
                            break;
                        }
                    }
                    if(pbest == 0)
                    {
                        S.error = "front too large for the LDS panels";
                        return false;
                    }
                    // (links of equal length instead of full ones + a remnant of two or three pivots: the panel need grows with the
                    //  pivots of a link, so the shorter link fits where the longest did)
                    if(big) pbest = (rest + (rest + pbest - 1) / pbest - 1) / ((rest + pbest - 1) / pbest);
                    b = a + pbest;
                }
                fr.push_back({a, b, c1});
                a = b;
            }
        }

        int const nf = static_cast<int>(fr.size());
        S.nfronts = nf;
        S.f_col0.resize(nf);
        S.f_p.resize(nf);
        S.f_u.resize(nf);
        S.f_parent.assign(nf, -1);
        S.f_rows_ptr.assign(nf + 1, 0);
        ivec col2front(n);
        for(int s = 0; s < nf; ++s)
        {
            S.f_col0[s] = fr[s].c0;
            S.f_p[s] = fr[s].c1 - fr[s].c0;
            S.f_u[s] = (fr[s].g1 - fr[s].c1) + static_cast<int>(st[fr[s].g1 - 1].size());
            S.f_rows_ptr[s + 1] = S.f_rows_ptr[s] + S.f_u[s];
            for(int c = fr[s].c0; c < fr[s].c1; ++c) col2front[c] = s;
            S.nnz_LU_stored += 2LL * S.f_p[s] * S.f_u[s] + static_cast<long long>(S.f_p[s]) * S.f_p[s];
        }
        S.f_rows.resize(S.f_rows_ptr[nf]);
        for(int s = 0; s < nf; ++s)
        {
            int* dst = S.f_rows.data() + S.f_rows_ptr[s];
            for(int c = fr[s].c1; c < fr[s].g1; ++c) *dst++ = c;
            auto const& r = st[fr[s].g1 - 1];
            std::copy(r.begin(), r.end(), dst);
            if(S.f_u[s] > 0) S.f_parent[s] = col2front[S.f_rows[S.f_rows_ptr[s]]];
            int const m = S.f_p[s] + S.f_u[s];
            S.max_m = std::max(S.max_m, m);
            S.max_u = std::max(S.max_u, S.f_u[s]);
            for(int k = 0; k < S.f_p[s]; ++k)
            {
                double const r1 = m - k - 1;
                S.flops += 2.0 * r1 * r1 + r1;
            }
        }
        if(S.max_m >= 65536)
        {
            S.error = "front order exceeds 65535";
            return false;
        }

        auto local_of = [&](int s, int t) -> int
        {
            int const c0 = S.f_col0[s], p = S.f_p[s];
            if(t < c0 + p) return t - c0;
            auto b = S.f_rows.begin() + S.f_rows_ptr[s], e = S.f_rows.begin() + S.f_rows_ptr[s + 1];
            auto it = std::lower_bound(b, e, t);
            if(it == e || *it != t) return -1;
            return p + static_cast<int>(it - b);
        };

        // children, relative indices, depth
        S.f_child_ptr.assign(nf + 1, 0);
        for(int s = 0; s < nf; ++s)
            if(S.f_parent[s] >= 0) ++S.f_child_ptr[S.f_parent[s] + 1];
        for(int s = 0; s < nf; ++s) S.f_child_ptr[s + 1] += S.f_child_ptr[s];
        S.f_child.resize(S.f_child_ptr[nf]);
        {
            ivec fill(S.f_child_ptr.begin(), S.f_child_ptr.end() - 1);
            for(int s = 0; s < nf; ++s)
                if(S.f_parent[s] >= 0) S.f_child[fill[S.f_parent[s]]++] = s;
        }
        S.f_rel_ptr = S.f_rows_ptr;
        S.f_rel.assign(S.f_rows.size(), -1);
        ivec depth(nf, 1);
        {
            for(int s = nf - 1; s >= 0; --s)
            {
                int const P = S.f_parent[s];
                if(P >= 0)
                {
                    depth[s] = depth[P] + 1;
                    if(P <= s)
                    {
                        S.error = "internal: front tree is not in postorder";
                        return false;
                    }
                    for(int a = S.f_rows_ptr[s]; a < S.f_rows_ptr[s + 1]; ++a)
                    {
                        int const l = local_of(P, S.f_rows[a]);
                        if(l < 0)
                        {
                            S.error = "internal: child update row missing from parent front";
                            return false;
                        }
                        S.f_rel[a] = l;
                    }
                }
                S.tree_depth = std::max(S.tree_depth, depth[s]);
            }
        }

        // permutations
        S.col_src = q;
        S.row_src.resize(n);
        for(int k = 0; k < n; ++k) S.row_src[k] = rmatch[q[k]];

        // assembly lists: A slot e=(i_old, j_old) -> permuted (k, l) -> owner front = front of min(k, l)
        {
            ivec qinv(n), eq2k(n);
            for(int k = 0; k < n; ++k) qinv[q[k]] = k;
            for(int k = 0; k < n; ++k) eq2k[S.row_src[k]] = k;
            S.f_asm_ptr.assign(nf + 1, 0);
            int const nnz = rp[n];
            ivec owner(nnz), pos(nnz);
            for(int i = 0; i < n; ++i)
            {
                int const k = eq2k[i];
                for(int e = rp[i]; e < rp[i + 1]; ++e)
                {
                    int const l = qinv[ci[e]];
                    int const s = col2front[std::min(k, l)];
                    int const r = local_of(s, k), c = local_of(s, l);
                    if(r < 0 || c < 0)
                    {
                        S.error = "internal: matrix entry outside its front";
                        return false;
                    }
                    owner[e] = s;
                    pos[e] = (r << 16) | c;
                    ++S.f_asm_ptr[s + 1];
                }
            }
            for(int s = 0; s < nf; ++s) S.f_asm_ptr[s + 1] += S.f_asm_ptr[s];
            S.asm_slot.resize(nnz);
            S.asm_pos.resize(nnz);
            ivec fill(S.f_asm_ptr.begin(), S.f_asm_ptr.end() - 1);
            for(int e = 0; e < nnz; ++e)
            {
                int const d = fill[owner[e]]++;
                S.asm_slot[d] = e;
                S.asm_pos[d] = pos[e];
            }
        }

        // ---------------- schedule
        // Two nested proportional cuts of the assembly tree (from the roots downwards):
        //   level 1 (only when n_parts > 1, the multi-workgroup mode for one or few instances): subtrees costing at most
        //           total / (n_parts * part_cut) become PART subtrees, spread over n_parts workgroups; what stays above is
        //           the TOP, processed one tree level per kernel launch with one workgroup per front;
        //   level 2 (inside every part): subtrees that fit one wavefront's LDS slot and cost at most
        //           part_total / (W * cut_factor) become WAVE subtrees, spread over the W wavefronts of the workgroup;
        //           the rest of the part is COOPERATIVE (whole workgroup).
        // Near a root there is no tree parallelism left, so the upper fronts are always handled by the wider executor.
        int const W = std::max(1, opt.n_waves);
        int const K = std::max(1, opt.n_parts);
        S.n_parts = K;
        S.f_kind.assign(nf, 1);
        ivec part_of(nf, K > 1 ? -1 : 0);
        std::vector<double> cost(nf, 0.0), sub(nf, 0.0);
        std::vector<char> fits(nf, 1), fits_mid(nf, opt.quad_mid ? 1 : 0), fits_quad(nf, opt.quad ? 1 : 0);
        double total = 0.0;
        for(int s = 0; s < nf; ++s)
        {
            double const m = S.f_p[s] + S.f_u[s];
            cost[s] = 400.0 + m * m * (2.0 + S.f_p[s]);
            total += cost[s];
            fits[s] = fits[s] && m <= opt.wave_m && S.f_p[s] <= opt.wave_p;
            // lane-group kernel (pe_quad.hpp): two row sets of 16, pivots in the first, one-byte entry indices
            // lane-group kernel (pe_quad.hpp): two row sets of 16, pivots in the first, one-byte entry indices -- a wave front is a QUAD front
            // when its whole subtree satisfies this; the other wave fronts stay with the per-instance wave phase of factor_part
            fits_quad[s] = fits_quad[s] && m <= 32 && S.f_p[s] <= 16 && S.f_asm_ptr[s + 1] - S.f_asm_ptr[s] <= 255 && S.f_child_ptr[s + 1] - S.f_child_ptr[s] <= 16;
            if(fits[s] && opt.wave_slot > 0)
            {
                // (the slot also holds the right-hand-side column: m doubles behind the image / the panels)
                long long const mm = S.f_p[s] + S.f_u[s], whole = (pe_ld(static_cast<int>(mm)) + 1LL) * mm,
                                panel = static_cast<long long>(pe_ld(static_cast<int>(mm))) * S.f_p[s] + static_cast<long long>(pe_ld(S.f_p[s])) * S.f_u[s] + mm;
                if(whole > opt.wave_slot && panel > opt.wave_slot) fits[s] = 0;
            }
            // MID class of the lane-group kernels: order <= 64 (four row sets), otherwise the wave fronts' limits, whole subtree
            fits_mid[s] = fits_mid[s] && m <= 64 && S.f_p[s] <= 16 && S.f_asm_ptr[s + 1] - S.f_asm_ptr[s] <= 255 && S.f_child_ptr[s + 1] - S.f_child_ptr[s] <= 16;
            sub[s] += cost[s];
            int const P = S.f_parent[s];
            if(P >= 0)
            {
                sub[P] += sub[s];
                if(!fits[s]) fits[P] = 0;
                if(!fits_mid[s]) fits_mid[P] = 0;
                if(!fits_quad[s]) fits_quad[P] = 0;
            }
        }
        auto mark_subtree = [&](int root, auto&& fn)
        {
            ivec st2{root};
            while(!st2.empty())
            {
                int const t = st2.back();
                st2.pop_back();
                fn(t);
                for(int a = S.f_child_ptr[t]; a < S.f_child_ptr[t + 1]; ++a) st2.push_back(S.f_child[a]);
            }
        };
        std::vector<double> part_total(K, 0.0);
        if(K > 1)
        {
            double const limit1 = total / (static_cast<double>(K) * opt.part_cut);
            ivec stack, sroots;
            for(int s = 0; s < nf; ++s)
                if(S.f_parent[s] < 0) stack.push_back(s);
            while(!stack.empty())
            {
                int const s = stack.back();
                stack.pop_back();
                if(sub[s] <= limit1) sroots.push_back(s);
                else
                    for(int a = S.f_child_ptr[s]; a < S.f_child_ptr[s + 1]; ++a) stack.push_back(S.f_child[a]);
            }
            std::sort(sroots.begin(), sroots.end(), [&](int a, int b) { return sub[a] != sub[b] ? sub[a] > sub[b] : a < b; });
            for(int r: sroots)
            {
                int const pidx = static_cast<int>(std::min_element(part_total.begin(), part_total.end()) - part_total.begin());
                part_total[pidx] += sub[r];
                mark_subtree(r, [&](int t) { part_of[t] = pidx; });
            }
            // parts in descending cost: the launch dispatches workgroups part by part, the cheapest last (shortest tail)
            ivec order(K), rank(K);
            for(int q = 0; q < K; ++q) order[q] = q;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return part_total[a] > part_total[b]; });
            std::vector<double> sorted_total(K);
            for(int q = 0; q < K; ++q)
            {
                rank[order[q]] = q;
                sorted_total[q] = part_total[order[q]];
            }
            part_total = sorted_total;
            for(int t = 0; t < nf; ++t)
                if(part_of[t] >= 0) part_of[t] = rank[part_of[t]];
        }
        else
            part_total[0] = total;
        // level 2
        for(int s = nf - 1; s >= 0; --s)
        {
            if(part_of[s] < 0)
            {
                S.f_kind[s] = 2;
                continue;
            }
            int const P = S.f_parent[s];
            bool const parent_is_wave = P >= 0 && part_of[P] == part_of[s] && S.f_kind[P] == 0;
            if(parent_is_wave) S.f_kind[s] = 0;  // inside a wave subtree
            else
            {
                double const limit2 = part_total[part_of[s]] / (opt.cut_factor * W);
                S.f_kind[s] = (fits[s] && sub[s] <= limit2) ? 0 : (fits_mid[s] ? 3 : 1);
            }
        }
        S.f_quad.assign(nf, 0);
        for(int s = 0; s < nf; ++s) S.f_quad[s] = (opt.quad && S.f_kind[s] == 0 && fits_quad[s]) ? 1 : 0;
        // executors: wave w of part q -> q * (W + 1) + w ; cooperative phase of part q -> q * (W + 1) + W ; top fronts: none
        ivec exec_of(nf, -1), wroot_of(nf, -1);
        S.f_wstack.assign(nf, 0);
        S.f_wpar.assign(nf, -1);
        S.wave_stack = 1;
        {
            // wave subtree roots per part, LPT over the W wavefronts
            std::vector<ivec> wroots(K);
            for(int s = nf - 1; s >= 0; --s)
            {
                if(S.f_kind[s] != 0) continue;
                int const P = S.f_parent[s];
                bool const inner = P >= 0 && part_of[P] == part_of[s] && S.f_kind[P] == 0;
                wroot_of[s] = inner ? wroot_of[P] : s;
                S.f_wpar[s] = inner ? S.f_wstack[P] : -1;
                S.f_wstack[s] = inner ? S.f_wstack[P] + S.f_p[P] + S.f_u[P] : 0;
                S.wave_stack = std::max(S.wave_stack, S.f_wstack[s] + S.f_p[s] + S.f_u[s]);
                if(!inner) wroots[part_of[s]].push_back(s);
            }
            // With the lane-group kernel the quad fronts of a list cost the part's workgroup nothing (they were factored by the launch
            // before, one wavefront per (quad, list), dynamically scheduled: the balance BETWEEN lists hardly matters there), so the W
            // wavefronts of a part are balanced on what they really walk -- the non-quad fronts (measured before: 140..230 us per
            // wavefront of a part with the lists balanced on all fronts) -- and the pure-quad subtrees even out the quad lists after that.
            std::vector<double> sub_nq(nf, 0.0);
            for(int s = 0; s < nf; ++s)
            {
                if(S.f_kind[s] != 0) continue;
                if(!S.f_quad[s]) sub_nq[s] += cost[s];
                int const P = S.f_parent[s];
                if(P >= 0 && S.f_kind[P] == 0 && part_of[P] == part_of[s]) sub_nq[P] += sub_nq[s];
            }
            for(int q = 0; q < K; ++q)
            {
                auto& r = wroots[q];
                std::sort(r.begin(), r.end(),
                          [&](int a, int b)
                          {
                              if(sub_nq[a] != sub_nq[b]) return sub_nq[a] > sub_nq[b];
                              return sub[a] != sub[b] ? sub[a] > sub[b] : a < b;
                          });
                std::vector<double> load(W, 0.0), load_q(W, 0.0);
                for(int root: r)
                {
                    int w = 0;
                    for(int k = 1; k < W; ++k)
                    {
                        bool const better = sub_nq[root] > 0.0 ? (load[k] < load[w] || (load[k] == load[w] && load_q[k] < load_q[w])) : load_q[k] < load_q[w];
                        if(better) w = k;
                    }
                    load[w] += sub_nq[root];
                    load_q[w] += sub[root] - sub_nq[root];
                    exec_of[root] = q * (W + 1) + w;
                }
            }
            for(int s = 0; s < nf; ++s)
            {
                if(S.f_kind[s] == 0) exec_of[s] = exec_of[wroot_of[s]];
                else if(S.f_kind[s] == 1 || S.f_kind[s] == 3)
                    exec_of[s] = part_of[s] * (W + 1) + W;
            }
        }
        // lists (ascending front index = postorder inside every executor)
        {
            std::vector<ivec> lists(static_cast<size_t>(K) * (W + 1));
            for(int s = 0; s < nf; ++s)
                if(exec_of[s] >= 0) lists[exec_of[s]].push_back(s);
            S.wave_ptr.assign(static_cast<size_t>(K) * (W + 1) + 1, 0);  // per part: W wave lists (entry W of a part is empty, keeps the stride)
            S.wave_list.clear();
            S.coop_ptr.assign(K + 1, 0);
            S.coop_list.clear();
            for(int q = 0; q < K; ++q)
            {
                for(int w = 0; w <= W; ++w)
                {
                    size_t const e = static_cast<size_t>(q) * (W + 1) + w;
                    if(w < W) S.wave_list.insert(S.wave_list.end(), lists[e].begin(), lists[e].end());
                    S.wave_ptr[e + 1] = static_cast<int>(S.wave_list.size());
                }
                auto const& cl = lists[static_cast<size_t>(q) * (W + 1) + W];
                S.coop_list.insert(S.coop_list.end(), cl.begin(), cl.end());
                S.coop_ptr[q + 1] = static_cast<int>(S.coop_list.size());
            }
            // top levels: height above the parts
            ivec h(nf, 0);
            int maxh = 0;
            for(int s = 0; s < nf; ++s)
            {
                if(S.f_kind[s] != 2) continue;
                int hh = 1;
                for(int a = S.f_child_ptr[s]; a < S.f_child_ptr[s + 1]; ++a)
                    if(S.f_kind[S.f_child[a]] == 2) hh = std::max(hh, h[S.f_child[a]] + 1);
                h[s] = hh;
                maxh = std::max(maxh, hh);
            }
            S.top_ptr.assign(maxh + 1, 0);
            S.top_list.clear();
            for(int lev = 1; lev <= maxh; ++lev)
            {
                for(int s = 0; s < nf; ++s)
                    if(S.f_kind[s] == 2 && h[s] == lev) S.top_list.push_back(s);
                S.top_ptr[lev] = static_cast<int>(S.top_list.size());
            }

            // storage offsets: factor panels
            S.f_lptr.resize(nf);
            S.f_uptr.resize(nf);
            S.f_sptr.assign(nf, 0);
            long long fo = 0;
            for(int s = 0; s < nf; ++s)
            {
                long long const p = S.f_p[s], u = S.f_u[s], m = p + u;
                S.f_lptr[s] = fo;
                fo += m * p;
                S.f_uptr[s] = fo;
                fo += p * u;
            }
            S.factor_doubles = fo;
            // update-matrix arena; a slot holds the u x u update matrix followed by the u-vector of the forward substitution
            // (the right-hand side rides through the factorisation as one more column, pe_front.hpp).  A front whose parent runs on a different executor (or is a top front) keeps its update
            // matrix in a persistent slot; inside one executor the matrices live on TWO LIFO stacks chosen by the parity of
            // the tree depth: a front writes its Schur tiles while later tiles still gather from its children's, so a parent
            // must never overlap its children -- children always sit on the other stack.
            long long base = 0;
            auto persistent = [&](int s)
            {
                int const P = S.f_parent[s];
                // (a MID front is factored by its own launch, before and in another order than the cooperative list it belongs to:
                //  its update matrix cannot live on that executor's LIFO stacks)
                //  its update matrix cannot live on that executor's LIFO stacks; likewise a QUAD front under a wave front of the per-instance path)
                return exec_of[s] < 0 || P < 0 || exec_of[P] != exec_of[s] || S.f_kind[s] == 3 || (S.f_quad[s] && !S.f_quad[P]);
            };
            for(int s = 0; s < nf; ++s)
                if(persistent(s))
                {
                    S.f_sptr[s] = base;
                    base += static_cast<long long>(S.f_u[s]) * (S.f_u[s] + 1);
                }
            for(auto const& lst: lists)
            {
                long long sp[2] = {0, 0}, peak[2] = {0, 0};
                std::vector<long long> rel_off(lst.size(), 0);
                for(size_t qi = 0; qi < lst.size(); ++qi)
                {
                    int const s = lst[qi];
                    for(int a = S.f_child_ptr[s]; a < S.f_child_ptr[s + 1]; ++a)
                    {
                        int const c = S.f_child[a];
                        if(!persistent(c)) sp[depth[c] & 1] -= static_cast<long long>(S.f_u[c]) * (S.f_u[c] + 1);
                    }
                    if(persistent(s)) continue;
                    int const q = depth[s] & 1;
                    rel_off[qi] = sp[q];
                    sp[q] += static_cast<long long>(S.f_u[s]) * (S.f_u[s] + 1);
                    peak[q] = std::max(peak[q], sp[q]);
                }
                if(sp[0] != 0 || sp[1] != 0)
                {
                    S.error = "internal: executor stacks not empty at the end";
                    return false;
                }
                for(size_t qi = 0; qi < lst.size(); ++qi)
                {
                    int const s = lst[qi];
                    if(!persistent(s)) S.f_sptr[s] = base + ((depth[s] & 1) ? peak[0] : 0) + rel_off[qi];
                }
                base += peak[0] + peak[1];
            }
            S.work_doubles = 0;
            if(opt.quad)
            {
                S.q_zero_off = base;  // never written: lanes of the lane-group kernel without a contribution read their zeros here
                base += Symbolic::Q_ZERO;
            }
            S.arena_doubles = base;
        }
        // inverse relative maps of every parent front
        S.f_inv_off.assign(S.f_child.size(), -1);
        S.f_cnp.assign(S.f_child.size(), 0);
        S.f_bmask.assign(S.f_child.size(), 0u);
        S.f_inv.clear();
        S.wave_panel_doubles = 1;
        for(int s = 0; s < nf; ++s)
        {
            int const m = S.f_p[s] + S.f_u[s];
            if(S.f_kind[s] == 0) S.wave_panel_doubles = std::max<long long>(S.wave_panel_doubles, static_cast<long long>(pe_ld(m)) * S.f_p[s] + static_cast<long long>(pe_ld(S.f_p[s])) * S.f_u[s]);
            for(int a = S.f_child_ptr[s]; a < S.f_child_ptr[s + 1]; ++a)
            {
                int const c = S.f_child[a];
                S.f_inv_off[a] = static_cast<long long>(S.f_inv.size());
                S.f_inv.resize(S.f_inv.size() + m, -1);
                int* inv = S.f_inv.data() + S.f_inv_off[a];
                for(int i = 0; i < S.f_u[c]; ++i)
                {
                    int const l = S.f_rel[S.f_rows_ptr[c] + i];
                    inv[l] = i;
                    if(l >= S.f_p[s]) S.f_bmask[a] |= 1u << std::min((l - S.f_p[s]) >> 4, 31);
                }
                int np = 0;
                while(np < S.f_u[c] && S.f_rel[S.f_rows_ptr[c] + np] < S.f_p[s]) ++np;
                S.f_cnp[a] = np;
            }
        }
        if(opt.quad) build_quad_plan(S, opt.quad_lds_doubles);
        return true;
    }

    // index program of the lane-group kernels (pe_symbolic.hpp "quad plan") for the given lists of fronts (each in postorder).  Every
    // front has <= 16 pivots, <= 255 own entries and <= 16 children (the schedule's `fits` / `fits_mid`); order <= 32 for the wave
    // fronts, <= 64 for the MID fronts; and so has every child (whole subtrees).
    static void build_quad_program(Symbolic& S, std::vector<std::vector<int>> const& lists, int lds_doubles, std::vector<int>& prog, std::vector<int>& lptr,
                                   std::vector<unsigned char>& lane)
    {
        prog.clear();
        lane.clear();
        lptr.assign(2 * lists.size(), 0);
        std::vector<int> lds_off(S.nfronts, -1);
        for(size_t L = 0; L < lists.size(); ++L)
        {
            auto const& lst = lists[L];
            lptr[2 * L] = static_cast<int>(prog.size());
            lptr[2 * L + 1] = static_cast<int>(lst.size());
            size_t prev_hdr = 0;
            int top = 0;  // LDS stack pointer of this list (doubles per instance)
            for(size_t i = 0; i < lst.size(); ++i)
            {
                int const s = lst[i];
                int const p = S.f_p[s], u = S.f_u[s], m = p + u, rs = (m + 15) / 16, M = 16 * rs, rec = M + 16;
                int const ch0 = S.f_child_ptr[s], nch = S.f_child_ptr[s + 1] - ch0;
                size_t const h = prog.size();
                prog.resize(h + Symbolic::Q_HDR + static_cast<size_t>(nch) * Symbolic::Q_CHILD, 0);
                int* q = prog.data() + h;
                size_t const l0 = lane.size();
                lane.resize(l0 + static_cast<size_t>(M) * rec, 0);
                q[0] = m;
                q[1] = p;
                q[2] = S.f_col0[s];
                q[3] = S.f_asm_ptr[s];
                q[4] = nch;
                q[5] = rs;
                q[6] = static_cast<int>(S.f_lptr[s] & 0xffffffffll);
                q[7] = static_cast<int>(S.f_lptr[s] >> 32);
                q[8] = static_cast<int>(S.f_sptr[s] & 0xffffffffll);
                q[9] = static_cast<int>(S.f_sptr[s] >> 32);
                q[10] = static_cast<int>(l0);
                {
                    // this front consumes its children (the top of the stack, postorder), then parks its own update matrix if its
                    // parent belongs to the same list and the stack has room
                    for(int a = nch - 1; a >= 0; --a)
                    {
                        int const c = S.f_child[ch0 + a];
                        if(lds_off[c] >= 0) top = lds_off[c];
                    }
                    int const need = u * (u + 1), P = S.f_parent[s];
                    bool const same_list = P >= 0 && std::find(lst.begin(), lst.end(), P) != lst.end();
                    if(lds_doubles > 0) S.q_lds_total += need;
                    if(same_list && need > 0 && top + need <= lds_doubles)
                    {
                        lds_off[s] = top;
                        top += need;
                        S.q_lds_kept += need;
                    }
                    q[13] = lds_off[s];
                }
                if(i > 0)
                {
                    prog[prev_hdr + 11] = rs;
                    prog[prev_hdr + 12] = static_cast<int>(l0);
                }
                prev_hdr = h;
                for(int e = S.f_asm_ptr[s]; e < S.f_asm_ptr[s + 1]; ++e)
                {
                    int const r = S.asm_pos[e] >> 16, c = S.asm_pos[e] & 0xffff;
                    lane[l0 + static_cast<size_t>(r) * rec + c] = static_cast<unsigned char>(1 + (e - S.f_asm_ptr[s]));
                }
                for(int a = 0; a < nch; ++a)
                {
                    int const c = S.f_child[ch0 + a];
                    int* cb = q + Symbolic::Q_HDR + a * Symbolic::Q_CHILD;
                    cb[0] = static_cast<int>(S.f_sptr[c]);
                    cb[1] = S.f_u[c];
                    cb[18] = lds_off[c];
                    unsigned char* cm = reinterpret_cast<unsigned char*>(cb + 2);
                    for(int k = 0; k < S.f_u[c]; ++k)
                    {
                        int const l = S.f_rel[S.f_rows_ptr[c] + k];
                        cm[l] = static_cast<unsigned char>(1 + k);
                        lane[l0 + static_cast<size_t>(l) * rec + M + a] = static_cast<unsigned char>(1 + k);
                    }
                }
            }
        }
        lane.resize(lane.size() + 64, 0);  // (slack: the prefetch of a row record reads whole 16-byte pieces)
    }

    void build_quad_plan(Symbolic& S, int lds_doubles)
    {
        S.quad = 1;
        S.q_lds_doubles = std::max(0, lds_doubles);
        S.q_lds_kept = S.q_lds_total = 0;
        int const K = std::max(1, S.n_parts), W = static_cast<int>((S.wave_ptr.size() - 1) / K) - 1;
        std::vector<std::vector<int>> lists(static_cast<size_t>(K) * W);
        for(int part = 0; part < K; ++part)
            for(int w = 0; w < W; ++w)
                for(int i = S.wave_ptr[part * (W + 1) + w]; i < S.wave_ptr[part * (W + 1) + w + 1]; ++i)
                    if(S.f_quad[S.wave_list[i]]) lists[static_cast<size_t>(part) * W + w].push_back(S.wave_list[i]);
        build_quad_program(S, lists, S.q_lds_doubles, S.q_prog, S.q_lists, S.q_lane);
        S.q_bprog.clear();
        // owner front of every pivot column: a front of the backward walk reads the unknowns of its update rows, i.e. of the fronts that
        // own those columns.  It needs the wavefront's fence only if one of them was written by this wavefront since its last fence
        // (the front right before it in the walk, typically its parent); unknowns from earlier launches or from behind an earlier fence
        // are in memory already -- q[6] tells (half of the fronts of M10k: a first child right behind its parent yes, its siblings no).
        std::vector<int> col_owner(static_cast<size_t>(S.n), -1);
        for(int s = 0; s < S.nfronts; ++s)
            for(int c = 0; c < S.f_p[s]; ++c) col_owner[static_cast<size_t>(S.f_col0[s] + c)] = s;
        std::vector<char> pending(static_cast<size_t>(S.nfronts), 0);
        for(auto const& lst: lists)  // (q_lists[2 L + 1] fronts of list L, stored back to back in list order: block index = fronts before it)
        {
            std::vector<int> since_fence;
            for(auto it = lst.rbegin(); it != lst.rend(); ++it)
            {
                int const s = *it;
                bool need = it == lst.rbegin();
                for(int j = 0; j < S.f_u[s] && !need; ++j)
                {
                    int const o = col_owner[static_cast<size_t>(S.f_rows[S.f_rows_ptr[s] + j])];
                    need = o >= 0 && pending[static_cast<size_t>(o)];
                }
                if(need)
                {
                    for(int t: since_fence) pending[static_cast<size_t>(t)] = 0;
                    since_fence.clear();
                }
                pending[static_cast<size_t>(s)] = 1;
                since_fence.push_back(s);
                size_t const h = S.q_bprog.size();
                S.q_bprog.resize(h + Symbolic::Q_BACK, 0);
                int* q = S.q_bprog.data() + h;
                q[0] = S.f_p[s] + S.f_u[s];
                q[1] = S.f_p[s];
                q[2] = S.f_col0[s];
                q[3] = S.f_u[s];
                q[4] = static_cast<int>(S.f_lptr[s] & 0xffffffffll);
                q[5] = static_cast<int>(S.f_lptr[s] >> 32);
                q[6] = need ? 1 : 0;
                for(int j = 0; j < S.f_u[s]; ++j) q[8 + j] = S.f_rows[S.f_rows_ptr[s] + j];
            }
            for(int t: since_fence) pending[static_cast<size_t>(t)] = 0;
        }
        // MID fronts: the subtrees of kind-3 fronts inside a part's cooperative list, dealt out to W lists per part (longest first)
        S.n_mid = 0;
        std::vector<std::vector<int>> mlists(static_cast<size_t>(K) * W);
        for(int part = 0; part < K; ++part)
        {
            std::vector<int> roots, root_of(S.nfronts, -1);
            std::vector<double> cost(S.nfronts, 0.0);
            for(int i = S.coop_ptr[part + 1] - 1; i >= S.coop_ptr[part]; --i)  // parents before children
            {
                int const s = S.coop_list[i];
                if(S.f_kind[s] != 3) continue;
                ++S.n_mid;
                int const P = S.f_parent[s];
                root_of[s] = (P >= 0 && S.f_kind[P] == 3) ? root_of[P] : s;
                if(root_of[s] == s) roots.push_back(s);
                double const m = S.f_p[s] + S.f_u[s];
                cost[root_of[s]] += 400.0 + m * m * (2.0 + S.f_p[s]);
            }
            std::sort(roots.begin(), roots.end(), [&](int a, int b) { return cost[a] != cost[b] ? cost[a] > cost[b] : a < b; });
            std::vector<double> load(W, 0.0);
            std::vector<int> list_of(S.nfronts, -1);
            for(int r: roots)
            {
                int const w = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
                load[w] += cost[r];
                list_of[r] = w;
            }
            for(int i = S.coop_ptr[part]; i < S.coop_ptr[part + 1]; ++i)  // (ascending = postorder)
            {
                int const s = S.coop_list[i];
                if(S.f_kind[s] == 3) mlists[static_cast<size_t>(part) * W + list_of[root_of[s]]].push_back(s);
            }
        }
        build_quad_program(S, mlists, 0, S.q2_prog, S.q2_lists, S.q2_lane);
    }

    bool build_assembly_lists(Symbolic& S, long long cap_wave, long long cap_team, int const* top_wide, long long cap_top, long long cap_mid)
    {
        int const nf = S.nfronts;
        S.f_mode.assign(nf, 0);
        S.f_keep.assign(nf, 0);
        // top fronts of a level that runs one wide workgroup per front own a CU's LDS: larger cap, and a chain link that follows its
        // child in the same run of single-front levels (same workgroup, one after the other) finds its front already in LDS
        std::vector<int> wide_level(nf, -1);  // level of a top front at a wide level, else -1
        if(top_wide && cap_top > 0)
            for(size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
                // (single-front levels only: measured at 128 instances, the levels with two fronts run 3..5 us SLOWER in the whole-front
                //  layout of the larger cap -- 27 -> 30, 31 -> 36 us -- while the run of single-front levels gains 12 us, 178 -> 166)
                //  (top_wide 2: a level with fronts formed against the larger cap -- every front of it is laid out against that cap)
                if(top_wide[l] == 2 || (top_wide[l] == 1 && S.top_ptr[l + 1] - S.top_ptr[l] == 1))
                    for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) wide_level[S.top_list[k]] = static_cast<int>(l);
        std::vector<char> mid_level(nf, 0);  // a top front at a level of the 8-wavefront launch with half a CU's LDS (top_wide 3)
        if(top_wide && cap_mid > 0)
            for(size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
                if(top_wide[l] == 3)
                    for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) mid_level[S.top_list[k]] = 1;
        S.gl_ptr.assign(nf + 1, 0);
        S.gl_rptr.assign(nf + 1, 0);
        S.gl_sptr.assign(nf + 1, 0);
        S.gl_dst.clear();
        S.gl_cnt.clear();
        S.gl_src.clear();
        if(S.arena_doubles >= (1ll << 31))
        {
            S.error = "update-matrix arena exceeds 2^31 doubles per instance";
            return false;
        }
        std::vector<int> first, next, srcv, cells, cnt;  // per LDS cell: chained list of sources (append order = children's order)
        std::vector<int> last;
        for(int s = 0; s < nf; ++s)
        {
            long long const p = S.f_p[s], u = S.f_u[s], m = p + u;
            long long const cap = S.f_kind[s] == 0 ? cap_wave : (wide_level[s] >= 0 ? std::max(cap_team, cap_top) : (mid_level[s] ? std::max(cap_team, cap_mid) : cap_team));
            int const ch0 = S.f_child_ptr[s], ch1 = S.f_child_ptr[s + 1];
            long long const ldl = pe_ld(static_cast<int>(m)), ldp = pe_ld(static_cast<int>(p));  // odd LDS leading dimensions (pe_device.hpp)
            bool const full = ldl * m + m <= cap;
            bool const link = ch1 - ch0 == 1 && S.f_u[S.f_child[ch0]] == m;
            bool const chain = !full && link;
            S.f_mode[s] = full ? 0 : (chain ? 2 : 1);
            if(link && wide_level[s] >= 1)
            {
                // continued in LDS: the child is the only front of the level below, ran in the whole-front layout (its Schur block sits
                // in its image) in the same workgroup right before this front
                int const c = S.f_child[ch0], l = wide_level[s];
                bool const same_run = wide_level[c] == l - 1 && S.top_ptr[l + 1] - S.top_ptr[l] == 1 && S.top_ptr[l] - S.top_ptr[l - 1] == 1;
                if(same_run && (S.f_mode[c] == 0 || S.f_mode[c] == 3))
                {
                    S.f_mode[s] = 3;
                    S.f_keep[c] = 1;
                }
            }
            if(S.f_mode[s] != 3 && !chain && ch1 > ch0)
            {
                long long const nlds = full ? ldl * m : ldl * p + ldp * u;
                if(nlds + m > 65535)
                {
                    S.error = "front image exceeds the 16-bit cell index of the assembly lists";
                    return false;
                }
                first.assign(static_cast<size_t>(nlds + m), -1);
                last.assign(static_cast<size_t>(nlds + m), -1);
                next.clear();
                srcv.clear();
                auto add = [&](long long d, long long src)
                {
                    int const e = static_cast<int>(srcv.size());
                    srcv.push_back(static_cast<int>(src));
                    next.push_back(-1);
                    if(first[d] < 0) first[d] = e;
                    else
                        next[last[d]] = e;
                    last[d] = e;
                };
                for(int a = ch0; a < ch1; ++a)
                {
                    int const c = S.f_child[a];
                    long long const uc = S.f_u[c], sp = S.f_sptr[c];
                    int const* rel = S.f_rel.data() + S.f_rows_ptr[c];
                    if(full)
                    {
                        for(long long j = 0; j < uc; ++j)
                            for(long long i = 0; i < uc; ++i) add(rel[i] + rel[j] * ldl, sp + i + j * uc);
                    }
                    else
                    {
                        long long np = 0;  // the child's leading update rows that are pivots of this front (f_rel ascends)
                        while(np < uc && rel[np] < p) ++np;
                        for(long long j = 0; j < np; ++j)
                            for(long long i = 0; i < uc; ++i) add(rel[i] + rel[j] * ldl, sp + i + j * uc);          // L panel (columns < p)
                        for(long long j = np; j < uc; ++j)
                            for(long long i = 0; i < np; ++i) add(ldl * p + rel[i] + (rel[j] - p) * ldp, sp + i + j * uc);  // U panel (rows < p)
                    }
                    for(long long i = 0; i < uc; ++i) add(nlds + rel[i], sp + uc * uc + i);  // update vector -> right-hand-side column
                }
                cells.clear();
                cnt.assign(first.size(), 0);
                for(size_t d = 0; d < first.size(); ++d)
                    if(first[d] >= 0)
                    {
                        cells.push_back(static_cast<int>(d));
                        for(int e = first[d]; e >= 0; e = next[e]) ++cnt[d];
                    }
                std::sort(cells.begin(), cells.end(),
                          [&](int a, int b)
                          {
                              if(cnt[a] != cnt[b]) return cnt[a] > cnt[b];
                              return srcv[first[a]] < srcv[first[b]];
                          });
                int const rounds = cells.empty() ? 0 : cnt[cells[0]];
                for(int d: cells) S.gl_dst.push_back(static_cast<unsigned short>(d));
                std::vector<int> cur(cells.size());
                for(size_t k = 0; k < cells.size(); ++k) cur[k] = first[cells[k]];
                for(int r = 0; r < rounds; ++r)
                {
                    int n_r = 0;
                    for(size_t k = 0; k < cells.size() && cnt[cells[k]] > r; ++k)
                    {
                        S.gl_src.push_back(srcv[cur[k]]);
                        cur[k] = next[cur[k]];
                        ++n_r;
                    }
                    S.gl_cnt.push_back(n_r);
                }
            }
            S.gl_ptr[s + 1] = static_cast<int>(S.gl_dst.size());
            S.gl_rptr[s + 1] = static_cast<int>(S.gl_cnt.size());
            S.gl_sptr[s + 1] = static_cast<long long>(S.gl_src.size());
        }
        return true;
    }
}  // namespace pe
