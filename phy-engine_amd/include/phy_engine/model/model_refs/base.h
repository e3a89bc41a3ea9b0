// path-compatible forwarding header (reference: include/phy_engine/model/model_refs/base.h)
#pragma once
#include <phy_engine/phy_engine_core.h>
