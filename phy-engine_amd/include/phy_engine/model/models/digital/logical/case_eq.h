// path-compatible forwarding header (reference: include/phy_engine/model/models/digital/logical/case_eq.h)
#pragma once
#include <phy_engine/digital_builtin.h>
