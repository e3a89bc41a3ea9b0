// path-compatible forwarding header (reference: include/phy_engine/model/models/non-linear/nmosfet.h)
#pragma once
#include <phy_engine/models_builtin.h>
