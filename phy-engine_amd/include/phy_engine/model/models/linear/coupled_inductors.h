// path-compatible forwarding header (reference: include/phy_engine/model/models/linear/coupled_inductors.h)
#pragma once
#include <phy_engine/models_builtin.h>
