// path-compatible forwarding header (reference: include/phy_engine/model/models/controller/comparator.h)
#pragma once
#include <phy_engine/digital_builtin.h>
