// path-compatible forwarding header (reference: include/phy_engine/model/models/controller/relay.h)
#pragma once
#include <phy_engine/models_builtin.h>
