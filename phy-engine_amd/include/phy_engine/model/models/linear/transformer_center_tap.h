// path-compatible forwarding header (reference: include/phy_engine/model/models/linear/transformer_center_tap.h)
#pragma once
#include <phy_engine/models_builtin.h>
