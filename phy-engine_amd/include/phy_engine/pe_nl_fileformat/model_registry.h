// pe_nl_fileformat/model_registry.h -- model name -> how to create one and how to carry its state (reference:
// pe_nl_fileformat/model_registry.h:24-58 for the entry / registry types, :60-105 and :463-577 for the state envelopes).
//
// State blobs.  The reference dumps the model OBJECT ([sizeof][alignof][bytes of the struct], model_registry.h:70-105): an ABI image of
// ITS struct, pin pointers included.  Such an image means nothing to another build, and this build keeps the analog simulation
// state (companion histories, junction state) on the device, not in the model objects (difference 4).  So:
//   * written here:  [sizeof][alignof]["PEHIPST1"][bytes] for trivially copyable models (the digital blocks' flip-flop / counter
//     state lives there), [0][0]["PEHIPST1"] otherwise.  The marker makes the length differ from the reference's envelope, so the
//     reference reports its own "length mismatch" instead of copying foreign bytes over its struct;
//   * read here:  a blob with the marker and this build's size restores the object -- pins excepted: connectivity is the
//     container's business (m/<id>/pins), the pointers in a dump are the writer's;  any other well-formed blob (a reference file)
//     is accepted and ignored: parameters travel as attributes, analog state restarts.
#pragma once
#include <cstring>
#include <string>
#include <string_view>
#include <type_traits>
#include <vector>

#include <phy_engine/phy_engine.h>

#include "codec.h"

namespace phy_engine::pe_nl_fileformat
{
    struct model_codec_entry
    {
        ::fast_io::u8string_view model_name{};
        ::phy_engine::netlist::add_model_retstr (*add_model)(::phy_engine::netlist::netlist& nl) noexcept {};
        status (*save_state)(::phy_engine::model::model_base const& m, std::string& out) {};
        status (*load_state)(::phy_engine::model::model_base& m, std::string_view in) {};
        status (*load_checkpoint_state)(::phy_engine::model::model_base& m, std::string_view in) {};  // (null: load_state)
    };

    class model_registry
    {
    public:
        void add(model_codec_entry e) { entries_.push_back(e); }
        [[nodiscard]] model_codec_entry const* find(::fast_io::u8string_view model_name) const noexcept
        {
            for(auto const& e: entries_)
                if(e.model_name.size() == model_name.size() && std::memcmp(e.model_name.data(), model_name.data(), model_name.size()) == 0) return &e;
            return nullptr;
        }

    private:
        std::vector<model_codec_entry> entries_{};
    };

    namespace details
    {
        inline constexpr std::string_view state_marker{"PEHIPST1"};

        template <typename Mod>
        inline ::phy_engine::model::details::model_derv_impl<Mod>* get_impl(::phy_engine::model::model_base const& mb) noexcept
        {
            return mb.ptr ? dynamic_cast<::phy_engine::model::details::model_derv_impl<Mod>*>(mb.ptr) : nullptr;
        }

        template <typename Mod>
        inline status save_object(::phy_engine::model::model_base const& mb, std::string& out)
        {
            auto const* impl = get_impl<Mod>(mb);
            if(impl == nullptr) return {errc::corrupt, "model type mismatch (save state)"};
            out.clear();
            if constexpr(std::is_trivially_copyable_v<Mod>)
            {
                append_uleb128(out, sizeof(Mod));
                append_uleb128(out, alignof(Mod));
                out.append(state_marker);
                append_bytes(out, &impl->m, sizeof(Mod));
            }
            else
            {
                append_uleb128(out, 0);
                append_uleb128(out, 0);
                out.append(state_marker);
            }
            return {};
        }

        template <typename Mod>
        inline status load_object(::phy_engine::model::model_base& mb, std::string_view in)
        {
            auto* impl = get_impl<Mod>(mb);
            if(impl == nullptr) return {errc::corrupt, "model type mismatch (load state)"};
            std::size_t off{};
            std::uint64_t sz{}, al{};
            if(auto st = read_uleb128(in, off, sz); !st) return st;
            if(auto st = read_uleb128(in, off, al); !st) return st;
            bool const own = in.size() - off >= state_marker.size() && in.substr(off, state_marker.size()) == state_marker;
            if(!own) return {};  // another build's object image: parameters came with the attributes
            off += state_marker.size();
            if constexpr(std::is_trivially_copyable_v<Mod>)
            {
                if(sz == 0 && off == in.size()) return {};
                if(sz != sizeof(Mod) || al != alignof(Mod)) return {errc::unsupported, "model state was written by a build with another object layout"};
                if(in.size() - off != sizeof(Mod)) return {errc::corrupt, "model state length mismatch"};
                // the pins keep what the container connected them to (and their names, which point into this program)
                auto pv = mb.ptr->generate_pin_view();
                std::vector<::phy_engine::model::pin> keep(pv.pins, pv.pins + pv.size);
                std::memcpy(static_cast<void*>(&impl->m), in.data() + off, sizeof(Mod));
                pv = mb.ptr->generate_pin_view();
                for(std::size_t i = 0; i < pv.size && i < keep.size(); ++i) pv.pins[i] = keep[i];
            }
            else if(off != in.size())
                return {errc::corrupt, "trailing bytes in model state"};
            return {};
        }

        template <typename Mod>
        inline model_codec_entry make_entry()
        {
            model_codec_entry e{};
            e.model_name = Mod::model_name;
            e.add_model = [](::phy_engine::netlist::netlist& nl) noexcept { return ::phy_engine::netlist::add_model(nl, Mod{}); };
            e.save_state = &save_object<Mod>;
            e.load_state = &load_object<Mod>;
            return e;
        }
    }  // namespace details
}  // namespace phy_engine::pe_nl_fileformat
