// pe_nl_fileformat/kv_store.h -- the key/value directory under a PE-NL container, in LevelDB's ON-DISK FORMAT.
//
// The reference stores a circuit as a LevelDB database (pe_nl_fileformat.h:584-803 writes ONE WriteBatch into a fresh database,
// :805-1313 reads it back with point lookups) and packs that directory into a single file (archive.h).  To exchange files with
// it this build needs LevelDB's formats, not its engine: no compaction, no concurrency, no iterators over live data.  This header
// is a fresh, minimal implementation of exactly that much, written from the published format description (LevelDB
// doc/log_format.md, doc/table_format.md, db/version_edit.cc tag numbers, db/write_batch.cc record layout) -- no LevelDB code:
//
//   write_fresh()  what `DB::Open(create_if_missing) + Write(batch)` leaves behind: CURRENT -> MANIFEST-000002 (comparator record +
//                  {log 3, next file 4, last sequence 0}), the batch as one record of 000003.log, empty LOCK.  LevelDB opens it.
//   read_all()     CURRENT -> MANIFEST -> live tables (*.ldb / *.sst, uncompressed blocks) + the write-ahead logs not yet
//                  flushed; newest sequence number per key wins, deletions honoured.  Covers a directory LevelDB has re-opened
//                  (its recovery turns the log into a level-0 table) as well as a fresh one.
//
// Checksums (CRC-32C, masked) are verified on everything read.  Snappy / zstd compressed table blocks are reported as
// `unsupported` (the reference's vendored LevelDB is built without either: its CMake finds no such library in its tree).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "status.h"

namespace phy_engine::pe_nl_fileformat::kv
{
    // ---- CRC-32C (Castagnoli, reflected polynomial 0x82f63b78), and the rotation + offset LevelDB stores it with
    inline std::uint32_t crc32c(void const* data, std::size_t n, std::uint32_t crc = 0) noexcept
    {
        static auto const table = []
        {
            std::array<std::uint32_t, 256> t{};
            for(std::uint32_t i = 0; i < 256; ++i)
            {
                std::uint32_t c = i;
                for(int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
                t[i] = c;
            }
            return t;
        }();
        auto const* p = static_cast<unsigned char const*>(data);
        crc = ~crc;
        for(std::size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return ~crc;
    }
    inline std::uint32_t crc_mask(std::uint32_t crc) noexcept { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

    inline void put_fixed32(std::string& out, std::uint32_t v) { out.append(reinterpret_cast<char const*>(&v), 4); }
    inline void put_fixed64(std::string& out, std::uint64_t v) { out.append(reinterpret_cast<char const*>(&v), 8); }
    inline void put_varint(std::string& out, std::uint64_t v)
    {
        for(; v >= 0x80u; v >>= 7u) out.push_back(static_cast<char>(v | 0x80u));
        out.push_back(static_cast<char>(v));
    }
    inline bool get_varint(std::string_view in, std::size_t& off, std::uint64_t& v)
    {
        v = 0;
        for(unsigned shift = 0; shift < 64 && off < in.size(); shift += 7)
        {
            auto const b = static_cast<unsigned char>(in[off++]);
            v |= static_cast<std::uint64_t>(b & 0x7fu) << shift;
            if(!(b & 0x80u)) return true;
        }
        return false;
    }
    inline bool get_fixed32(std::string_view in, std::size_t off, std::uint32_t& v)
    {
        if(off > in.size() || in.size() - off < 4) return false;  // (overflow-free: `off` may come from a 64-bit length in the file)
        std::memcpy(&v, in.data() + off, 4);
        return true;
    }
    inline bool get_slice(std::string_view in, std::size_t& off, std::string_view& s)  // varint32 length + bytes
    {
        std::uint64_t n{};
        if(!get_varint(in, off, n) || n > in.size() - off) return false;
        s = in.substr(off, static_cast<std::size_t>(n));
        off += static_cast<std::size_t>(n);
        return true;
    }

    // ---- record log (write-ahead log and MANIFEST): 32 KiB blocks of [masked crc32c(type + payload) : 4][length : 2][type : 1][payload];
    // type 1 = whole record, 2 / 3 / 4 = first / middle / last fragment; a block tail shorter than a header is zero filled
    inline constexpr std::size_t log_block = 32768, log_header = 7;
    inline void log_append(std::string& file, std::string_view payload)
    {
        std::size_t pos = 0;
        for(bool first = true;; first = false)
        {
            std::size_t const left = log_block - file.size() % log_block;
            if(left < log_header) file.append(left, '\0');  // (a block tail too short for a header is zero filled)
            std::size_t const room = log_block - file.size() % log_block - log_header;
            std::size_t const n = std::min(payload.size() - pos, room);
            bool const last = pos + n == payload.size();
            char const type = static_cast<char>(first ? (last ? 1 : 2) : (last ? 4 : 3));
            std::uint32_t crc = crc32c(&type, 1);
            crc = crc32c(payload.data() + pos, n, crc);
            put_fixed32(file, crc_mask(crc));
            file.push_back(static_cast<char>(n & 0xffu));
            file.push_back(static_cast<char>(n >> 8));
            file.push_back(type);
            file.append(payload.substr(pos, n));
            pos += n;
            if(last) break;
        }
    }
    inline status log_records(std::string_view file, std::vector<std::string>& out, char const* what)
    {
        std::string cur;
        bool in_record = false;
        for(std::size_t base = 0; base < file.size(); base += log_block)
        {
            std::string_view const blk = file.substr(base, log_block);
            std::size_t off = 0;
            while(blk.size() - off >= log_header)
            {
                std::uint32_t stored{};
                get_fixed32(blk, off, stored);
                std::size_t const n = static_cast<unsigned char>(blk[off + 4]) | (static_cast<std::size_t>(static_cast<unsigned char>(blk[off + 5])) << 8);
                int const type = static_cast<unsigned char>(blk[off + 6]);
                if(type == 0 && n == 0 && stored == 0) break;  // zero fill (pre-allocated tail): nothing more in this block
                if(off + log_header + n > blk.size())
                {
                    // a fragment that runs past a SHORT last block is a torn tail (a writer that died mid-append): end of log, as LevelDB's
                    // reader reports it (kEof, dropped silently); past a full 32 KiB block it is corruption
                    if(base + log_block >= file.size() && blk.size() < log_block) return {};
                    return {errc::corrupt, std::string(what) + ": record fragment runs past its block"};
                }
                char const tb = static_cast<char>(type);
                std::uint32_t crc = crc32c(&tb, 1);
                crc = crc32c(blk.data() + off + log_header, n, crc);
                if(crc_mask(crc) != stored) return {errc::corrupt, std::string(what) + ": record checksum mismatch"};
                std::string_view const frag = blk.substr(off + log_header, n);
                off += log_header + n;
                if(type == 1 || type == 2)
                {
                    if(in_record) return {errc::corrupt, std::string(what) + ": fragment sequence broken"};
                    cur.assign(frag);
                    in_record = type == 2;
                    if(type == 1) out.push_back(std::move(cur)), cur.clear();
                }
                else if(type == 3 || type == 4)
                {
                    if(!in_record) return {errc::corrupt, std::string(what) + ": fragment sequence broken"};
                    cur.append(frag);
                    if(type == 4)
                    {
                        out.push_back(std::move(cur));
                        cur.clear();
                        in_record = false;
                    }
                }
                else
                    return {errc::corrupt, std::string(what) + ": unknown record type"};
            }
        }
        // (a trailing partial record = a writer that died mid-append: LevelDB drops it silently, so does this reader)
        return {};
    }

    // ---- one version of a user key
    struct versioned
    {
        std::uint64_t seq{};
        bool deleted{};
        std::string value{};
    };
    using table_t = std::map<std::string, versioned, std::less<>>;
    inline void offer(table_t& t, std::string_view key, std::uint64_t seq, bool deleted, std::string_view value)
    {
        auto it = t.find(key);
        if(it == t.end()) t.emplace(std::string(key), versioned{seq, deleted, std::string(value)});
        else if(seq >= it->second.seq)
            it->second = versioned{seq, deleted, std::string(value)};
    }

    // write batch: [sequence : 8][count : 4] then per entry [1 = put | 0 = delete][key slice][value slice (put only)]
    inline std::string encode_batch(std::uint64_t seq, std::vector<std::pair<std::string, std::string>> const& entries)
    {
        std::string b;
        put_fixed64(b, seq);
        put_fixed32(b, static_cast<std::uint32_t>(entries.size()));
        for(auto const& [k, v]: entries)
        {
            b.push_back(1);
            put_varint(b, k.size());
            b.append(k);
            put_varint(b, v.size());
            b.append(v);
        }
        return b;
    }
    inline status apply_batch(std::string_view rec, table_t& t)
    {
        if(rec.size() < 12) return {errc::corrupt, "write batch shorter than its header"};
        std::uint64_t seq{};
        std::uint32_t count{};
        std::memcpy(&seq, rec.data(), 8);
        std::memcpy(&count, rec.data() + 8, 4);
        std::size_t off = 12;
        for(std::uint32_t i = 0; i < count; ++i)
        {
            if(off >= rec.size()) return {errc::corrupt, "write batch: fewer entries than announced"};
            int const tag = static_cast<unsigned char>(rec[off++]);
            std::string_view k, v;
            if(!get_slice(rec, off, k)) return {errc::corrupt, "write batch: bad key"};
            if(tag == 1)
            {
                if(!get_slice(rec, off, v)) return {errc::corrupt, "write batch: bad value"};
            }
            else if(tag != 0)
                return {errc::corrupt, "write batch: unknown entry tag"};
            offer(t, k, seq + i, tag == 0, v);
        }
        return {};
    }

    inline status read_file(std::filesystem::path const& p, std::string& out)
    {
        std::ifstream in(p, std::ios::binary);
        if(!in) return {errc::io_error, "cannot open " + p.string()};
        out.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        return {};
    }
    inline status write_file(std::filesystem::path const& p, std::string_view data)
    {
        std::ofstream out(p, std::ios::binary | std::ios::trunc);
        if(!out) return {errc::io_error, "cannot create " + p.string()};
        out.write(data.data(), static_cast<std::streamsize>(data.size()));
        out.flush();
        if(!out) return {errc::io_error, "cannot write " + p.string()};
        return {};
    }

    // ---- sorted table: data blocks + index block + footer (48 bytes: two block handles, padding, magic)
    inline constexpr std::uint64_t table_magic = 0xdb4775248b80fb57ull;
    inline status table_block(std::string_view file, std::uint64_t offset, std::uint64_t size, std::string_view& contents, char const* what)
    {
        // (overflow-free: offset and size are 64-bit varints from a footer / index entry no checksum protects -- size = 2^64 - 5 .. 2^64 - 1
        //  wrapped `size + 5` and passed the check; ASan: heap-buffer-overflow read below)
        if(offset > file.size() || size > file.size() - offset || file.size() - offset - size < 5) return {errc::corrupt, std::string(what) + ": block handle out of range"};
        std::string_view const raw = file.substr(static_cast<std::size_t>(offset), static_cast<std::size_t>(size) + 5);
        std::uint32_t stored{};
        get_fixed32(raw, static_cast<std::size_t>(size) + 1, stored);
        if(crc_mask(crc32c(raw.data(), static_cast<std::size_t>(size) + 1)) != stored) return {errc::corrupt, std::string(what) + ": block checksum mismatch"};
        if(raw[static_cast<std::size_t>(size)] != 0) return {errc::unsupported, std::string(what) + ": compressed table block (snappy / zstd) -- not produced by the reference's LevelDB build"};
        contents = raw.substr(0, static_cast<std::size_t>(size));
        return {};
    }
    // entries of a block: [shared : varint][non-shared : varint][value length : varint][key suffix][value], then the restart array
    template <class F>
    inline status block_entries(std::string_view blk, char const* what, F&& f)
    {
        std::uint32_t n_restarts{};
        if(blk.size() < 4 || !get_fixed32(blk, blk.size() - 4, n_restarts) || (static_cast<std::uint64_t>(n_restarts) + 1) * 4 > blk.size())
            return {errc::corrupt, std::string(what) + ": bad restart array"};
        std::size_t const end = blk.size() - (static_cast<std::size_t>(n_restarts) + 1) * 4;
        std::string key;
        for(std::size_t off = 0; off < end;)
        {
            std::uint64_t shared{}, fresh{}, vlen{};
            if(!get_varint(blk, off, shared) || !get_varint(blk, off, fresh) || !get_varint(blk, off, vlen) || shared > key.size() || fresh > end - off || vlen > end - off - fresh)
                return {errc::corrupt, std::string(what) + ": bad block entry"};
            key.resize(static_cast<std::size_t>(shared));
            key.append(blk.substr(off, static_cast<std::size_t>(fresh)));
            off += static_cast<std::size_t>(fresh);
            if(auto st = f(std::string_view{key}, blk.substr(off, static_cast<std::size_t>(vlen))); !st) return st;
            off += static_cast<std::size_t>(vlen);
        }
        return {};
    }
    inline status table_entries(std::string_view file, table_t& t, char const* what)
    {
        if(file.size() < 48) return {errc::corrupt, std::string(what) + ": shorter than a table footer"};
        std::string_view const footer = file.substr(file.size() - 48);
        std::uint64_t magic{};
        std::memcpy(&magic, footer.data() + 40, 8);
        if(magic != table_magic) return {errc::corrupt, std::string(what) + ": not a sorted table (bad magic)"};
        std::size_t off = 0;
        std::uint64_t mo{}, ms{}, io{}, is{};
        if(!get_varint(footer, off, mo) || !get_varint(footer, off, ms) || !get_varint(footer, off, io) || !get_varint(footer, off, is))
            return {errc::corrupt, std::string(what) + ": bad footer"};
        std::string_view index;
        if(auto st = table_block(file, io, is, index, what); !st) return st;
        return block_entries(index, what,
                             [&](std::string_view, std::string_view handle) -> status
                             {
                                 std::size_t ho = 0;
                                 std::uint64_t bo{}, bs{};
                                 if(!get_varint(handle, ho, bo) || !get_varint(handle, ho, bs)) return {errc::corrupt, std::string(what) + ": bad block handle in the index"};
                                 std::string_view data;
                                 if(auto st = table_block(file, bo, bs, data, what); !st) return st;
                                 return block_entries(data, what,
                                                      [&](std::string_view ikey, std::string_view value) -> status
                                                      {
                                                          // internal key = user key + [sequence << 8 | type] (8 bytes, little endian)
                                                          if(ikey.size() < 8) return {errc::corrupt, std::string(what) + ": internal key shorter than its trailer"};
                                                          std::uint64_t tail{};
                                                          std::memcpy(&tail, ikey.data() + ikey.size() - 8, 8);
                                                          offer(t, ikey.substr(0, ikey.size() - 8), tail >> 8, (tail & 0xffu) == 0, value);
                                                          return {};
                                                      });
                             });
    }

    // ---- MANIFEST: a log of version edits, fields tagged 1 comparator, 2 log number, 3 next file, 4 last sequence, 5 compact pointer,
    // 6 deleted file, 7 new file, 9 previous log number
    struct manifest_state
    {
        std::string comparator{};
        std::uint64_t log_number{}, prev_log_number{}, last_sequence{};
        std::map<std::uint64_t, int> files{};  // live table file -> level
    };
    inline status apply_edit(std::string_view rec, manifest_state& m)
    {
        for(std::size_t off = 0; off < rec.size();)
        {
            std::uint64_t tag{}, a{}, b{}, c{};
            std::string_view s1, s2;
            if(!get_varint(rec, off, tag)) return {errc::corrupt, "MANIFEST: bad tag"};
            bool ok = true;
            switch(tag)
            {
                case 1: ok = get_slice(rec, off, s1); m.comparator.assign(s1); break;
                case 2: ok = get_varint(rec, off, m.log_number); break;
                case 9: ok = get_varint(rec, off, m.prev_log_number); break;
                case 3: ok = get_varint(rec, off, a); break;
                case 4: ok = get_varint(rec, off, m.last_sequence); break;
                case 5: ok = get_varint(rec, off, a) && get_slice(rec, off, s1); break;
                case 6:
                    ok = get_varint(rec, off, a) && get_varint(rec, off, b);
                    if(ok) m.files.erase(b);
                    break;
                case 7:
                    ok = get_varint(rec, off, a) && get_varint(rec, off, b) && get_varint(rec, off, c) && get_slice(rec, off, s1) && get_slice(rec, off, s2);
                    if(ok) m.files[b] = static_cast<int>(a);
                    break;
                default: return {errc::corrupt, "MANIFEST: unknown field tag " + std::to_string(tag)};
            }
            if(!ok) return {errc::corrupt, "MANIFEST: truncated field"};
        }
        return {};
    }

    inline std::string file_name(std::uint64_t number, char const* suffix)
    {
        char buf[32];
        std::snprintf(buf, sizeof buf, "%06llu.%s", static_cast<unsigned long long>(number), suffix);
        return buf;
    }

    // every live key of the database directory `dir`
    inline status read_all(std::filesystem::path const& dir, std::map<std::string, std::string, std::less<>>& out)
    {
        out.clear();
        std::string current;
        if(auto st = read_file(dir / "CURRENT", current); !st) return {errc::db_error, "not a PE-NL database directory (no CURRENT): " + dir.string()};
        while(!current.empty() && (current.back() == '\n' || current.back() == '\r')) current.pop_back();
        if(current.empty() || current.find('/') != std::string::npos) return {errc::corrupt, "CURRENT does not name a MANIFEST"};
        std::string mf;
        if(auto st = read_file(dir / current, mf); !st) return st;
        std::vector<std::string> edits;
        if(auto st = log_records(mf, edits, "MANIFEST"); !st) return st;
        manifest_state m;
        for(auto const& e: edits)
            if(auto st = apply_edit(e, m); !st) return st;
        if(!m.comparator.empty() && m.comparator != "leveldb.BytewiseComparator") return {errc::unsupported, "database uses comparator " + m.comparator};
        table_t t;
        for(auto const& [number, level]: m.files)
        {
            std::string data;
            auto p = dir / file_name(number, "ldb");
            if(!std::filesystem::exists(p)) p = dir / file_name(number, "sst");
            if(auto st = read_file(p, data); !st) return st;
            if(auto st = table_entries(data, t, p.filename().string().c_str()); !st) return st;
        }
        // write-ahead logs the MANIFEST has not retired yet, oldest first
        std::vector<std::pair<std::uint64_t, std::filesystem::path>> logs;
        std::error_code ec;
        for(auto const& ent: std::filesystem::directory_iterator(dir, ec))
        {
            auto const name = ent.path().filename().string();
            if(name.size() < 5 || name.substr(name.size() - 4) != ".log") continue;
            std::uint64_t number = 0;
            bool digits = true;
            for(char ch: name.substr(0, name.size() - 4))
            {
                if(ch < '0' || ch > '9')
                {
                    digits = false;
                    break;
                }
                number = number * 10 + static_cast<std::uint64_t>(ch - '0');
            }
            if(digits && (number >= m.log_number || number == m.prev_log_number)) logs.emplace_back(number, ent.path());
        }
        std::sort(logs.begin(), logs.end());
        for(auto const& [number, p]: logs)
        {
            std::string data;
            if(auto st = read_file(p, data); !st) return st;
            std::vector<std::string> recs;
            if(auto st = log_records(data, recs, p.filename().string().c_str()); !st) return st;
            for(auto const& r: recs)
                if(auto st = apply_batch(r, t); !st) return st;
        }
        for(auto& [k, v]: t)
            if(!v.deleted) out.emplace(k, std::move(v.value));
        return {};
    }

    // a fresh database directory holding exactly `entries` (one write batch, as the reference's save_to_leveldb leaves it)
    inline status write_fresh(std::filesystem::path const& dir, std::vector<std::pair<std::string, std::string>> const& entries)
    {
        std::error_code ec;
        std::filesystem::create_directories(dir, ec);
        if(ec) return {errc::io_error, "cannot create " + dir.string()};
        // (overwrite: whatever database files sat there are removed first, like DestroyDB)
        for(auto const& ent: std::filesystem::directory_iterator(dir, ec))
        {
            auto const name = ent.path().filename().string();
            bool const ours = name == "CURRENT" || name == "LOCK" || name == "LOG" || name == "LOG.old" || name.rfind("MANIFEST-", 0) == 0 ||
                              (name.size() > 4 && (name.substr(name.size() - 4) == ".log" || name.substr(name.size() - 4) == ".ldb" || name.substr(name.size() - 4) == ".sst"));
            if(ours) std::filesystem::remove(ent.path(), ec);
        }
        std::string log;
        log_append(log, encode_batch(1, entries));
        std::string manifest, edit;
        edit.push_back(1);
        put_varint(edit, 26);
        edit.append("leveldb.BytewiseComparator");
        log_append(manifest, edit);
        edit.clear();
        edit.push_back(2), put_varint(edit, 3);  // log number
        edit.push_back(9), put_varint(edit, 0);  // previous log number
        edit.push_back(3), put_varint(edit, 4);  // next file number
        edit.push_back(4), put_varint(edit, 0);  // last sequence (the batch in the log carries sequence 1)
        log_append(manifest, edit);
        if(auto st = write_file(dir / "000003.log", log); !st) return st;
        if(auto st = write_file(dir / "MANIFEST-000002", manifest); !st) return st;
        if(auto st = write_file(dir / "LOCK", ""); !st) return st;
        return write_file(dir / "CURRENT", "MANIFEST-000002\n");
    }
}  // namespace phy_engine::pe_nl_fileformat::kv
