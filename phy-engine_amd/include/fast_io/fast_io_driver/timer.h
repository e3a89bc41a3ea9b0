#pragma once
// include-path compatibility: ::fast_io::timer lives in the vocabulary shim
#include "../fast_io.h"
