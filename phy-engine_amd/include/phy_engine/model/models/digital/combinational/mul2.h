// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/combinational/mul2.h)
#pragma once
#include <phy_engine/digital_builtin.h>
