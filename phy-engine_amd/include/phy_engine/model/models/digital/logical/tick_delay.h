// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/logical/tick_delay.h)
#pragma once
#include <phy_engine/digital_builtin.h>
