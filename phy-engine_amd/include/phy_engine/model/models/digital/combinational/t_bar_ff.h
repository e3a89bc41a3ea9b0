// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/combinational/t_bar_ff.h)
#pragma once
#include <phy_engine/digital_builtin.h>
