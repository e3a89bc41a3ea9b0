// path-compatible forwarding header (reference: include/phy_engine/model/model_refs/concept.h)
#pragma once
#include <phy_engine/phy_engine_core.h>
