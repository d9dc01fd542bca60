// Forwarding header: put include/compat in front of the reference's include path and its callers
// (#include "ternary_image_codec_v6_min.hpp": old/src/main.cpp, old/src/main_bare.cpp, old/include/io_image.hpp ...)
// get the MI355X implementation of the Word27 hot path instead of the header-only CPU one.
#pragma once
#include "../ternary_codec_v6.hpp"
