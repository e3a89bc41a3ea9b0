// Vocabulary shim: Phy-Engine's plug-in API spells its string-view type ::fast_io::u8string_view
// (model/model_refs/concept.h:196-217 requires `model_name` / `identification_name` of exactly that type).  The
// MI355X host layer does not vendor fast_io; this alias keeps model definitions source compatible.
#pragma once
#include <string_view>
namespace fast_io
{
    using u8string_view = ::std::u8string_view;
}
