// pe_nl_fileformat/status.h -- result type of the PE-NL container API (same names and codes as the reference's
// include/phy_engine/pe_nl_fileformat/status.h:7-31, so callers compile unchanged).
#pragma once
#include <string>
#include <utility>

namespace phy_engine::pe_nl_fileformat
{
    enum class errc
    {
        ok = 0,
        invalid_argument,
        io_error,
        db_error,
        corrupt,
        unsupported,
        not_found
    };

    struct status
    {
        errc code{errc::ok};
        std::string message{};

        status() noexcept = default;
        status(errc c, std::string m) : code{c}, message{std::move(m)} {}
        [[nodiscard]] bool ok() const noexcept { return code == errc::ok; }
        [[nodiscard]] explicit operator bool() const noexcept { return ok(); }
        static status success() { return {}; }
    };
}  // namespace phy_engine::pe_nl_fileformat
