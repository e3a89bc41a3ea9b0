// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/logical/non_implication.h)
#pragma once
#include <phy_engine/digital_builtin.h>
