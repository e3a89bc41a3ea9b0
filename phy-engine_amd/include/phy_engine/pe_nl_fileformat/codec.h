// pe_nl_fileformat/codec.h -- byte encodings of the PE-NL values (reference: pe_nl_fileformat/codec.h:20-163): little-endian fixed
// width integers and IEEE doubles, ULEB128 varints, length-prefixed byte strings.
#pragma once
#include <bit>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <type_traits>

#include "status.h"

namespace phy_engine::pe_nl_fileformat::details
{
    static_assert(std::endian::native == std::endian::little, "the container stores little-endian values; this build targets x86-64 hosts");

    inline void append_bytes(std::string& out, void const* p, std::size_t n) { out.append(static_cast<char const*>(p), n); }
    template <typename T>
    inline void append_trivial(std::string& out, T const& v)
    {
        static_assert(std::is_trivially_copyable_v<T>);
        append_bytes(out, &v, sizeof(T));
    }
    inline void append_u8(std::string& out, std::uint8_t v) { out.push_back(static_cast<char>(v)); }
    inline void append_uleb128(std::string& out, std::uint64_t v)
    {
        for(; v >= 0x80u; v >>= 7u) append_u8(out, static_cast<std::uint8_t>(v | 0x80u));
        append_u8(out, static_cast<std::uint8_t>(v));
    }
    inline void append_f64(std::string& out, double x) { append_trivial(out, x); }
    inline void append_string(std::string& out, std::string_view s)
    {
        append_uleb128(out, s.size());
        out.append(s);
    }

    inline status read_uleb128(std::string_view in, std::size_t& off, std::uint64_t& v)
    {
        v = 0;
        for(unsigned shift = 0;; shift += 7)
        {
            if(shift >= 64) return {errc::corrupt, "varint overflow"};
            if(off >= in.size()) return {errc::corrupt, "unexpected EOF while reading varint"};
            auto const b = static_cast<std::uint8_t>(in[off++]);
            v |= static_cast<std::uint64_t>(b & 0x7fu) << shift;
            if(!(b & 0x80u)) return {};
        }
    }
    template <typename T>
    inline status read_trivial(std::string_view in, std::size_t& off, T& v)
    {
        static_assert(std::is_trivially_copyable_v<T>);
        if(off > in.size() || in.size() - off < sizeof(T)) return {errc::corrupt, "unexpected EOF while reading fixed-size value"};
        std::memcpy(&v, in.data() + off, sizeof(T));
        off += sizeof(T);
        return {};
    }
    inline status read_f64(std::string_view in, std::size_t& off, double& x) { return read_trivial(in, off, x); }
    inline status read_u8(std::string_view in, std::size_t& off, std::uint8_t& v, char const* what)
    {
        if(off >= in.size()) return {errc::corrupt, std::string("unexpected EOF reading ") + what};
        v = static_cast<std::uint8_t>(in[off++]);
        return {};
    }
    inline status read_string(std::string_view in, std::size_t& off, std::string& s)
    {
        std::uint64_t n{};
        if(auto st = read_uleb128(in, off, n); !st) return st;
        if(n > in.size() - off) return {errc::corrupt, "invalid string length"};
        s.assign(in.data() + off, static_cast<std::size_t>(n));
        off += static_cast<std::size_t>(n);
        return {};
    }

    // the text types of the plug-in API are char8_t based; the container stores their bytes
    template <class U8>
    inline std::string u8sv_to_bytes(U8 const& v)
    {
        return std::string(reinterpret_cast<char const*>(v.data()), v.size());
    }
    inline std::u8string bytes_to_u8string(std::string_view v) { return std::u8string(reinterpret_cast<char8_t const*>(v.data()), v.size()); }

    // FNV-1a, 64 bit (archive checksum, stable ids)
    inline constexpr std::uint64_t fnv1a_basis{14695981039346656037ull};
    inline constexpr std::uint64_t fnv1a_prime{1099511628211ull};
    inline std::uint64_t fnv1a_update(std::uint64_t h, void const* data, std::size_t n) noexcept
    {
        auto const* p = static_cast<unsigned char const*>(data);
        for(std::size_t i = 0; i < n; ++i) h = (h ^ p[i]) * fnv1a_prime;
        return h;
    }
}  // namespace phy_engine::pe_nl_fileformat::details
