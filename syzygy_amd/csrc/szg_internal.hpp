// szg_internal.hpp — what the translation units of libszg_hip.so share besides the public headers.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace szg
{
// sets the calling thread's szg_last_error() text (defined in szg_api.cpp)
void set_last_error(const char* message);
} // namespace szg
