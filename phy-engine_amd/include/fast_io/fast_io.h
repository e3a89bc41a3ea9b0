#pragma once
// Vocabulary shim, NOT the fast_io library.  Phy-Engine's plug-in API and its tests use a thin slice of fast_io: the string-view
// type in the model concepts (fast_io_dsal/string_view.h) and, in test programs, formatted printing (`::fast_io::io::print /
// println / perr / perrln`, the manipulators `mnp::fixed` / `mnp::boolalpha`), a scope timer and an entropy engine.  The
// MI355X host layer does not vendor fast_io; this header provides that slice over the standard library so that user models and
// the reference's own test programs compile unchanged against `phy-engine_amd/include` (tests/test_reference_sources.py).
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <random>
#include <string>
#include <string_view>
#include <type_traits>

#include "fast_io_dsal/string_view.h"
#include "fast_io_dsal/vector.h"

namespace fast_io
{
    namespace mnp
    {
        template <class T>
        struct fixed_t
        {
            T value;
        };
        template <class T>
        constexpr fixed_t<T> fixed(T const& v) noexcept
        {
            return {v};
        }
        struct boolalpha_t
        {
            bool value;
        };
        constexpr boolalpha_t boolalpha(bool v) noexcept { return {v}; }
    }  // namespace mnp

    namespace details
    {
        inline void emit(::std::FILE* f, char const* s) { ::std::fputs(s, f); }
        inline void emit(::std::FILE* f, ::std::string_view s) { ::std::fwrite(s.data(), 1, s.size(), f); }
        inline void emit(::std::FILE* f, ::std::string const& s) { ::std::fwrite(s.data(), 1, s.size(), f); }
        inline void emit(::std::FILE* f, ::std::u8string_view s) { ::std::fwrite(s.data(), 1, s.size(), f); }
        inline void emit(::std::FILE* f, ::std::u8string const& s) { ::std::fwrite(s.data(), 1, s.size(), f); }
        inline void emit(::std::FILE* f, char8_t const* s) { emit(f, ::std::u8string_view{s}); }
        inline void emit(::std::FILE* f, bool v) { ::std::fputs(v ? "1" : "0", f); }
        inline void emit(::std::FILE* f, char c) { ::std::fputc(c, f); }
        inline void emit(::std::FILE* f, mnp::boolalpha_t v) { ::std::fputs(v.value ? "true" : "false", f); }
        template <class T>
            requires (::std::is_floating_point_v<T>)
        inline void emit(::std::FILE* f, T v)
        {
            ::std::fprintf(f, "%.17g", static_cast<double>(v));
        }
        template <class T>
            requires (::std::is_integral_v<T> && !::std::is_same_v<T, bool> && !::std::is_same_v<T, char>)
        inline void emit(::std::FILE* f, T v)
        {
            if constexpr(::std::is_signed_v<T>) ::std::fprintf(f, "%lld", static_cast<long long>(v));
            else
                ::std::fprintf(f, "%llu", static_cast<unsigned long long>(v));
        }
        template <class T>
            requires (::std::is_enum_v<T>)
        inline void emit(::std::FILE* f, T v)
        {
            emit(f, static_cast<::std::underlying_type_t<T>>(v));
        }
        template <class T>
        inline void emit(::std::FILE* f, ::std::complex<T> const& v)
        {
            ::std::fprintf(f, "(%.17g,%.17g)", static_cast<double>(v.real()), static_cast<double>(v.imag()));
        }
        template <class T>
        inline void emit(::std::FILE* f, mnp::fixed_t<T> const& v)
        {
            if constexpr(::std::is_arithmetic_v<T>) ::std::fprintf(f, "%f", static_cast<double>(v.value));
            else
                ::std::fprintf(f, "(%f,%f)", static_cast<double>(v.value.real()), static_cast<double>(v.value.imag()));
        }
        template <class... Args>
        inline void emit_all(::std::FILE* f, Args const&... args)
        {
            (emit(f, args), ...);
        }
    }  // namespace details

    namespace io
    {
        template <class... Args>
        inline void print(Args const&... args)
        {
            details::emit_all(stdout, args...);
        }
        template <class... Args>
        inline void println(Args const&... args)
        {
            details::emit_all(stdout, args..., '\n');
        }
        template <class... Args>
        inline void perr(Args const&... args)
        {
            details::emit_all(stderr, args...);
        }
        template <class... Args>
        inline void perrln(Args const&... args)
        {
            details::emit_all(stderr, args..., '\n');
        }
    }  // namespace io
    using io::perr;
    using io::perrln;
    using io::print;
    using io::println;

    // scope timer: prints "<label>: <seconds>s" to stderr when it goes out of scope
    struct timer
    {
        ::std::u8string label;
        ::std::chrono::steady_clock::time_point t0{::std::chrono::steady_clock::now()};
        explicit timer(::std::u8string_view l) : label{l} {}
        timer(timer const&) = delete;
        timer& operator=(timer const&) = delete;
        ~timer()
        {
            double const s{::std::chrono::duration<double>(::std::chrono::steady_clock::now() - t0).count()};
            details::emit_all(stderr, label, ": ", s, "s\n");
        }
    };

    // uniform random bit generator backed by the operating system's entropy source
    struct ibuf_white_hole_engine
    {
        using result_type = ::std::uint_least64_t;
        ::std::random_device dev{};
        static constexpr result_type min() noexcept { return 0; }
        static constexpr result_type max() noexcept { return ~result_type{}; }
        result_type operator()() { return (static_cast<result_type>(dev()) << 32) | static_cast<result_type>(dev()); }
    };
}  // namespace fast_io
