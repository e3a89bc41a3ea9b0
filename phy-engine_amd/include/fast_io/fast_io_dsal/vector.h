#pragma once
// Vocabulary shim (see ../fast_io.h): ::fast_io::vector as the plug-in API exposes it -- std::vector plus the *_unchecked
// accessors user code calls on circult's index tables (e.g. `c.size_t_to_node_p.index_unchecked(i)`).
#include <cstddef>
#include <utility>
#include <vector>
namespace fast_io
{
    template <class T>
    struct vector : ::std::vector<T>
    {
        using base = ::std::vector<T>;
        using base::base;
        vector() = default;
        T& index_unchecked(::std::size_t i) noexcept { return base::data()[i]; }
        T const& index_unchecked(::std::size_t i) const noexcept { return base::data()[i]; }
        T& front_unchecked() noexcept { return *base::data(); }
        T const& front_unchecked() const noexcept { return *base::data(); }
        T& back_unchecked() noexcept { return base::data()[base::size() - 1]; }
        T const& back_unchecked() const noexcept { return base::data()[base::size() - 1]; }
        void push_back_unchecked(T const& v) { base::push_back(v); }
        void push_back_unchecked(T&& v) { base::push_back(::std::move(v)); }
        template <class... A>
        T& emplace_back_unchecked(A&&... a)
        {
            return base::emplace_back(::std::forward<A>(a)...);
        }
    };
}  // namespace fast_io
