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
// host_jpeg.cpp: baseline / extended-sequential JPEG -> tightly packed RGBA8 (alpha 255); `why` says what failed
bool decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& why);
} // namespace szg
