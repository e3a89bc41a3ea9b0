// tests/emu/quad_emu.hpp -- TEST INFRASTRUCTURE ONLY: the host execution model of pe_quad.hpp.  The 64 lanes of a wavefront are the
// 64 elements of a vector type; control flow is wavefront-uniform in that code, so running it once with vector values IS running
// the 64 lanes in lockstep.  Every load / store goes through plain pointers: AddressSanitizer sees each lane's address.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>

namespace pe
{
    template <class T>
    struct Vec64
    {
        T v[64];
        Vec64() = default;
        Vec64(T s)
        {
            for(int l = 0; l < 64; ++l) v[l] = s;
        }
    };
#define PE_V64_BIN(op)                                                      \
    template <class T>                                                      \
    Vec64<T> operator op(Vec64<T> const& a, Vec64<T> const& b)              \
    {                                                                       \
        Vec64<T> r;                                                         \
        for(int l = 0; l < 64; ++l) r.v[l] = static_cast<T>(a.v[l] op b.v[l]); \
        return r;                                                           \
    }                                                                       \
    template <class T, class S>                                             \
    Vec64<T> operator op(Vec64<T> const& a, S b)                            \
    {                                                                       \
        Vec64<T> r;                                                         \
        for(int l = 0; l < 64; ++l) r.v[l] = static_cast<T>(a.v[l] op static_cast<T>(b)); \
        return r;                                                           \
    }
    PE_V64_BIN(/)
    PE_V64_BIN(+)
    PE_V64_BIN(-)
    PE_V64_BIN(*)
    PE_V64_BIN(&)
    PE_V64_BIN(|)
#undef PE_V64_BIN
    template <class T>
    Vec64<T> operator<<(Vec64<T> const& a, int s)
    {
        Vec64<T> r;
        for(int l = 0; l < 64; ++l) r.v[l] = static_cast<T>(a.v[l] << s);
        return r;
    }
    template <class T>
    Vec64<T> operator>>(Vec64<T> const& a, int s)
    {
        Vec64<T> r;
        for(int l = 0; l < 64; ++l) r.v[l] = static_cast<T>(a.v[l] >> s);
        return r;
    }
    template <class T>
    Vec64<T> operator-(Vec64<T> const& a)
    {
        Vec64<T> r;
        for(int l = 0; l < 64; ++l) r.v[l] = -a.v[l];
        return r;
    }
#define PE_V64_CMP(op)                                                 \
    template <class T, class S>                                        \
    Vec64<bool> operator op(Vec64<T> const& a, S b)                    \
    {                                                                  \
        Vec64<bool> r;                                                 \
        for(int l = 0; l < 64; ++l) r.v[l] = a.v[l] op static_cast<T>(b); \
        return r;                                                      \
    }
    PE_V64_CMP(<)
    PE_V64_CMP(>)
    PE_V64_CMP(<=)
    PE_V64_CMP(>=)
    PE_V64_CMP(==)
    PE_V64_CMP(!=)
#undef PE_V64_CMP

    struct QuadEmu
    {
        using vd = Vec64<double>;
        using vi = Vec64<int>;
        using vu = Vec64<unsigned>;
        using vm = Vec64<bool>;
        static vm& cur()
        {
            static thread_local vm m{false};
            return m;
        }
        static vi lane()
        {
            vi r;
            for(int l = 0; l < 64; ++l) r.v[l] = l;
            return r;
        }
        static vu to_u(vi const& a)
        {
            vu r;
            for(int l = 0; l < 64; ++l) r.v[l] = static_cast<unsigned>(a.v[l]);
            return r;
        }
        static vd bcast(vd const& a, int k)
        {
            vd r;
            for(int l = 0; l < 64; ++l) r.v[l] = a.v[(l & ~15) + k];
            return r;
        }
        static vd ld(char const* base, vu const& off)
        {
            vd r;
            for(int l = 0; l < 64; ++l) std::memcpy(&r.v[l], base + off.v[l], 8);
            return r;
        }
        static void ld_u32x4(unsigned char const* base, vu const& off, vu* out)
        {
            for(int l = 0; l < 64; ++l)
                for(int k = 0; k < 4; ++k) std::memcpy(&out[k].v[l], base + off.v[l] + 4 * k, 4);
        }
        static vi ld_i32(int const* base, vu const& off)
        {
            vi r;
            for(int l = 0; l < 64; ++l) std::memcpy(&r.v[l], reinterpret_cast<char const*>(base) + off.v[l], 4);
            return r;
        }
        template <class F>
        static void when(vm const& mask, F&& body)
        {
            cur() = mask;
            body();
            cur() = vm{false};
        }
        static void st(char* base, vu const& off, vd const& v)
        {
            vm const& m = cur();
            for(int l = 0; l < 64; ++l)
                if(m.v[l]) std::memcpy(base + off.v[l], &v.v[l], 8);
        }
        // the wavefront's LDS (bounds-checked: an out-of-range address aborts the test run)
        static std::vector<double>& lds()
        {
            static thread_local std::vector<double> mem;
            return mem;
        }
        static vd lds_ld(vu const& addr)
        {
            vd r;
            for(int l = 0; l < 64; ++l)
            {
                if(addr.v[l] % 8 != 0 || addr.v[l] / 8 >= lds().size()) std::abort();
                r.v[l] = lds()[addr.v[l] / 8];
            }
            return r;
        }
        static void lds_st(vu const& addr, vd const& v)
        {
            vm const& m = cur();
            for(int l = 0; l < 64; ++l)
                if(m.v[l])
                {
                    if(addr.v[l] % 8 != 0 || addr.v[l] / 8 >= lds().size()) std::abort();
                    lds()[addr.v[l] / 8] = v.v[l];
                }
        }
        static void lds_fence() {}
        static void fence() {}
        static void st_if(bool all, vm const& mask, char* base, vu const& off, vd const& v)
        {
            vm const& m = cur();
            for(int l = 0; l < 64; ++l)
                if(m.v[l] && (all || mask.v[l])) std::memcpy(base + off.v[l], &v.v[l], 8);
        }
        template <class T>
        static Vec64<T> sel(vm const& m, Vec64<T> const& a, Vec64<T> const& b)
        {
            Vec64<T> r;
            for(int l = 0; l < 64; ++l) r.v[l] = m.v[l] ? a.v[l] : b.v[l];
            return r;
        }
        static vd rcp(vd const& d)
        {
            vd r;
            for(int l = 0; l < 64; ++l) r.v[l] = emu_rcp(d.v[l]);
            return r;
        }
        static vd fma(vd const& a, vd const& b, vd const& c)
        {
            vd r;
            for(int l = 0; l < 64; ++l) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]);
            return r;
        }
        static vm bad(vd const& p)
        {
            vm r;
            for(int l = 0; l < 64; ++l) r.v[l] = p.v[l] == 0.0 || !(std::fabs(p.v[l]) <= 1.7976931348623157e308);
            return r;
        }
        static vm none() { return vm{false}; }
        static long long clock() { return 0; }
        static long long clock(vd const&) { return 0; }
        static void prof(long long* dst, long long const* v, int n)
        {
            for(int k = 0; k < n; ++k) dst[k] += v[k];
        }
        static void flag(int* f, vi const& idx, int bits, vm const& mask)
        {
            for(int l = 0; l < 64; ++l)
                if(mask.v[l]) f[idx.v[l]] |= bits;
        }
    };
}  // namespace pe
