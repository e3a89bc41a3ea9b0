// pe_top_plan.hpp -- host side: which launch runs which top levels of the assembly tree with how much LDS.  ONE definition for the
// HIP launcher (pe_kernels.hip m2_sequence) and for the host emulation of the kernels (tests/emu): the emulation gives every level
// of a launch exactly the LDS of that launch, so a level that ends up on a launch with too little of it trips the capacity assert
// of front_factor on the CPU instead of overrunning the LDS on the device.
#pragma once
#include "pe_device.hpp"

namespace pe
{
    struct TopLaunch
    {
        int level, nlev;        // levels level .. level + nlev - 1 (nlev > 1: a run of single-front levels, one workgroup per instance)
        int kind;               // 0: the plan's workgroup (k_m2_factor_top), 1: 16 wavefronts + a CU's LDS (k_m2_factor_top_wide),
                                // 2: 8 wavefronts, the plan's LDS, 3: 8 wavefronts + half a CU's LDS (k_m2_factor_top_mid)
        long long lds_doubles;  // dynamic LDS of the launch = what every front of these levels may use
    };

    // launches that differ in their LDS (V.top_wide: 0 -> the plan's share, 1 / 2 -> a CU's, 3 -> half a CU's)
    inline int top_launch_class(DevView const& V, int l) { return V.top_wide[l] == 3 ? 3 : (V.top_wide[l] != 0 ? 1 : 0); }

    // runs of single-front levels OF THE SAME CLASS share a launch (a launch per level is most of such a level's time)
    inline int top_run(DevView const& V, int l)
    {
        int n = 1;
        if(V.top_cnt[l] == 1)
            while(l + n < V.n_top_levels && V.top_cnt[l + n] == 1 && (V.top_run_any_class || top_launch_class(V, l + n) == top_launch_class(V, l))) ++n;
        return n;
    }

    // `mid_ok`: the geometry has an 8-wavefront variant (four 4-wavefront workgroups per CU); `mid_limit`: levels of at most this many
    // workgroups (and more than the wide limit) use it with the plan's LDS
    template <class F>
    inline void for_each_top_launch(DevView const& V, int batch, bool mid_ok, int mid_limit, F&& f)
    {
        for(int l = 0; l < V.n_top_levels;)
        {
            int const n = top_run(V, l), cls = top_launch_class(V, l);
            TopLaunch t{l, n, 0, V.lds_doubles};
            if(cls == 1) t = {l, n, 1, V.lds_top_doubles};
            else if(cls == 3)
                t = {l, n, 3, V.lds_mid_doubles};
            else if(mid_ok && V.top_cnt[l] * batch <= mid_limit)
                t.kind = 2;
            f(t);
            l += n;
        }
    }
}  // namespace pe
