// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/combinational/d_latch.h)
#pragma once
#include <phy_engine/digital_builtin.h>
