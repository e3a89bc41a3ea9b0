#pragma once
#include "fast_io_dsal/string_view.h"
