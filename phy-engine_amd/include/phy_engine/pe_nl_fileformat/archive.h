// pe_nl_fileformat/archive.h -- the single-file layout of a PE-NL container: the database directory packed into one file
// (reference: pe_nl_fileformat/archive.h:103-264, same bytes):
//
//   "PENLDBA1" | u32 version = 1 | u64 file count | per file { u64 path length, path (relative, '/'), u64 size, bytes } | u64 FNV-1a
//
// The checksum runs over everything after the header exactly as it is stored (lengths as 8 little-endian bytes).
#pragma once
#include <cstdint>
#include <filesystem>
#include <fstream>
#include <string>
#include <string_view>
#include <vector>

#include "codec.h"
#include "status.h"

namespace phy_engine::pe_nl_fileformat
{
    namespace details
    {
        inline status ensure_safe_relative_path(std::string_view rel)
        {
            std::filesystem::path const p{std::string(rel)};
            if(p.empty()) return {errc::corrupt, "empty path in archive"};
            if(p.is_absolute()) return {errc::corrupt, "absolute paths not allowed in archive"};
            for(auto const& part: p.lexically_normal())
                if(part == "..") return {errc::corrupt, "parent path (..) not allowed in archive"};
            return {};
        }
    }  // namespace details

    inline status pack_directory_to_file(std::filesystem::path const& dir, std::filesystem::path const& out_file)
    {
        std::error_code ec;
        if(!std::filesystem::is_directory(dir, ec)) return {errc::invalid_argument, "pack: input is not a directory"};
        std::vector<std::filesystem::path> files;
        for(auto const& ent: std::filesystem::recursive_directory_iterator(dir, ec))
            if(ent.is_regular_file()) files.push_back(ent.path());
        std::sort(files.begin(), files.end());  // (any order is valid; a fixed one makes the archive reproducible)
        std::string body;
        for(auto const& abs: files)
        {
            std::string const rel = std::filesystem::relative(abs, dir).generic_string();
            if(auto st = details::ensure_safe_relative_path(rel); !st) return st;
            std::ifstream in(abs, std::ios::binary);
            if(!in) return {errc::io_error, "failed to open input file while packing"};
            std::string const bytes{std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>()};
            details::append_trivial(body, static_cast<std::uint64_t>(rel.size()));
            body.append(rel);
            details::append_trivial(body, static_cast<std::uint64_t>(bytes.size()));
            body.append(bytes);
        }
        std::string head{"PENLDBA1"};
        details::append_trivial(head, static_cast<std::uint32_t>(1));
        details::append_trivial(head, static_cast<std::uint64_t>(files.size()));
        std::uint64_t const hash = details::fnv1a_update(details::fnv1a_basis, body.data(), body.size());
        std::ofstream out(out_file, std::ios::binary | std::ios::trunc);
        if(!out) return {errc::io_error, "failed to open output file"};
        out.write(head.data(), static_cast<std::streamsize>(head.size()));
        out.write(body.data(), static_cast<std::streamsize>(body.size()));
        out.write(reinterpret_cast<char const*>(&hash), sizeof hash);
        out.flush();
        if(!out) return {errc::io_error, "failed finalizing archive file"};
        return {};
    }

    inline status unpack_file_to_directory(std::filesystem::path const& in_file, std::filesystem::path const& dir)
    {
        std::ifstream in(in_file, std::ios::binary);
        if(!in) return {errc::io_error, "failed to open archive file"};
        std::string const all{std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>()};
        std::string_view const v{all};
        if(v.size() < 8) return {errc::io_error, "failed reading magic"};
        if(v.substr(0, 8) != "PENLDBA1") return {errc::unsupported, "not a pe_nl single-file archive"};
        std::size_t off = 8;
        std::uint32_t ver{};
        std::uint64_t count{};
        if(!details::read_trivial(v, off, ver)) return {errc::io_error, "failed reading u32"};
        if(ver != 1) return {errc::unsupported, "unsupported archive version"};
        if(!details::read_trivial(v, off, count)) return {errc::io_error, "failed reading u64"};
        if(count > 10'000'000ull) return {errc::corrupt, "archive file_count too large"};
        std::size_t const body0 = off;
        std::vector<std::pair<std::string, std::string_view>> entries;
        for(std::uint64_t i = 0; i < count; ++i)
        {
            std::uint64_t n{};
            if(!details::read_trivial(v, off, n)) return {errc::io_error, "failed reading u64"};
            if(n > 4096) return {errc::corrupt, "archive path too long"};
            if(n > v.size() - off) return {errc::io_error, "failed reading path bytes"};
            std::string rel{v.substr(off, static_cast<std::size_t>(n))};
            off += static_cast<std::size_t>(n);
            if(auto st = details::ensure_safe_relative_path(rel); !st) return st;
            if(!details::read_trivial(v, off, n)) return {errc::io_error, "failed reading u64"};
            if(n > v.size() - off) return {errc::io_error, "failed reading file bytes while unpacking"};
            entries.emplace_back(std::move(rel), v.substr(off, static_cast<std::size_t>(n)));
            off += static_cast<std::size_t>(n);
        }
        std::uint64_t stored{};
        std::size_t const body1 = off;
        if(!details::read_trivial(v, off, stored)) return {errc::io_error, "failed reading u64"};
        // (verified BEFORE anything is written: a damaged archive leaves no half-unpacked directory behind)
        if(details::fnv1a_update(details::fnv1a_basis, v.data() + body0, body1 - body0) != stored) return {errc::corrupt, "archive checksum mismatch"};
        std::error_code ec;
        std::filesystem::create_directories(dir, ec);
        for(auto const& [rel, bytes]: entries)
        {
            auto const out_path = dir / std::filesystem::path{rel};
            std::filesystem::create_directories(out_path.parent_path(), ec);
            std::ofstream out(out_path, std::ios::binary | std::ios::trunc);
            if(!out) return {errc::io_error, "failed to create output file while unpacking"};
            out.write(bytes.data(), static_cast<std::streamsize>(bytes.size()));
            if(!out) return {errc::io_error, "failed writing file bytes while unpacking"};
        }
        return {};
    }
}  // namespace phy_engine::pe_nl_fileformat
