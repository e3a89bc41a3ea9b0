// oracle/ref_kvdump.cpp -- TEST INFRASTRUCTURE: opens a database directory with the REAL LevelDB (the reference's vendored sources,
// compiled by oracle/Makefile) and prints every live key with the size and FNV-1a hash of its value.  tests/test_penl.py uses it to
// prove that the directories phy_engine/pe_nl_fileformat/kv_store.h writes are LevelDB databases (fragmented log records included).
#include <cstdint>
#include <cstdio>
#include <memory>

#include <leveldb/db.h>

int main(int argc, char** argv)
{
    if(argc < 2) return 1;
    leveldb::Options o;
    o.create_if_missing = false;
    o.paranoid_checks = true;
    leveldb::DB* raw{};
    auto const st = leveldb::DB::Open(o, argv[1], &raw);
    if(!st.ok())
    {
        std::fprintf(stderr, "open: %s\n", st.ToString().c_str());
        return 2;
    }
    std::unique_ptr<leveldb::DB> db{raw};
    leveldb::ReadOptions ro;
    ro.verify_checksums = true;
    std::unique_ptr<leveldb::Iterator> it{db->NewIterator(ro)};
    for(it->SeekToFirst(); it->Valid(); it->Next())
    {
        std::uint64_t h = 14695981039346656037ull;
        auto const v = it->value();
        for(std::size_t i = 0; i < v.size(); ++i) h = (h ^ static_cast<unsigned char>(v[i])) * 1099511628211ull;
        std::printf("%s %zu %016llx\n", it->key().ToString().c_str(), v.size(), static_cast<unsigned long long>(h));
    }
    return it->status().ok() ? 0 : 3;
}
