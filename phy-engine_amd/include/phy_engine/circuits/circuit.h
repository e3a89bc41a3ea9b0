// path-compatible forwarding header (reference: include/phy_engine/circuits/circuit.h)
#pragma once
#include <phy_engine/phy_engine_core.h>
