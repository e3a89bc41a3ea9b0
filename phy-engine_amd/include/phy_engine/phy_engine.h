// umbrella header (reference: include/phy_engine/phy_engine.h)
#pragma once
#include <phy_engine/phy_engine_core.h>
#include <phy_engine/models_builtin.h>
#include <phy_engine/digital_builtin.h>
