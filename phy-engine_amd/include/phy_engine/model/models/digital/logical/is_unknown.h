// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/logical/is_unknown.h)
#pragma once
#include <phy_engine/digital_builtin.h>
