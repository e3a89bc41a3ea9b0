// path-compatible forwarding header (reference: include/phy_engine/model/models/non-linear/full_bridge_rectifier.h)
#pragma once
#include <phy_engine/models_builtin.h>
