// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/combinational/random_generator4.h)
#pragma once
#include <phy_engine/digital_builtin.h>
