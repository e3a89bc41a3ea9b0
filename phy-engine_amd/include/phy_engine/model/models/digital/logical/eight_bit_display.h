// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/logical/eight_bit_display.h)
#pragma once
#include <phy_engine/digital_builtin.h>
