// fuzz_assets.cpp — memory-safety harness for include/szg/assets.h, built with AddressSanitizer + UBSan on the CPU
// (tests/test_assets_sanitized.py). Reads seed files, applies random byte mutations (and truncations) and feeds every
// variant to the glTF / GLB / PNG loaders: whatever the input, the loaders may only fail with a status code.
//   usage: fuzz_assets <rounds> <seed file>...      (".png" seeds go to the image decoder, ".gltf" to the JSON loader)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "szg/assets.h"

namespace szg
{
void set_last_error(const char*) {} // the library's own definition lives in szg_api.cpp (HIP); not needed here
} // namespace szg

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t next()
{
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return g_state;
}

static unsigned long run(const std::vector<uint8_t>& data, int kind)
{
    unsigned long ok = 0;
    if (kind == 2)
    {
        uint32_t w = 0, h = 0;
        uint8_t* rgba = nullptr;
        if (szg_decode_image_rgba(data.data(), data.size(), &w, &h, &rgba) == SZG_OK)
        {
            volatile uint8_t sink = rgba[(size_t)w * h * 4 - 1];
            (void)sink;
            ok++;
        }
        szg_free_rgba(rgba);
        return ok;
    }
    szg_gltf* asset = nullptr;
    if (szg_gltf_load_memory(data.data(), data.size(), kind, nullptr, SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES, &asset) == SZG_OK)
    {
        ok++;
        for (uint32_t i = 0; i < szg_gltf_mesh_count(asset); i++)
        {
            szg_asset_mesh m;
            szg_gltf_mesh(asset, i, &m);
            volatile float sink = m.vertex_count != 0 ? m.vertices[m.vertex_count - 1].color[3] : 0.0f;
            (void)sink;
        }
        for (uint32_t i = 0; i < szg_gltf_material_count(asset); i++)
        {
            szg_asset_material m;
            szg_gltf_material(asset, i, &m);
        }
        szg_gltf_destroy(asset);
    }
    return ok;
}

int main(int argc, char** argv)
{
    if (argc < 3)
    {
        return 2;
    }
    int const rounds = std::atoi(argv[1]);
    unsigned long total = 0, loaded = 0;
    for (int f = 2; f < argc; f++)
    {
        std::string const path = argv[f];
        std::ifstream file(path, std::ios::binary);
        std::vector<uint8_t> const seed((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        if (seed.empty())
        {
            std::fprintf(stderr, "cannot read %s\n", path.c_str());
            return 2;
        }
        auto ends = [&](const char* s) { return path.size() >= std::strlen(s) && path.compare(path.size() - std::strlen(s), std::string::npos, s) == 0; };
        int const kind = ends(".png") ? 2 : ends(".gltf") ? 0 : 1;
        loaded += run(seed, kind);
        total++;
        for (int r = 0; r < rounds; r++)
        {
            std::vector<uint8_t> data = seed;
            int const edits = 1 + (int)(next() % 6);
            for (int e = 0; e < edits; e++)
            {
                size_t const at = next() % data.size();
                switch (next() % 4)
                {
                case 0: data[at] ^= (uint8_t)(1u << (next() % 8)); break;
                case 1: data[at] = (uint8_t)next(); break;
                case 2: data[at] = (next() & 1) ? 0xFF : 0x00; break;
                default:
                    if (data.size() > 16 && (next() % 8) == 0)
                    {
                        data.resize(1 + next() % data.size());
                    }
                    else
                    {
                        data[at] += 1;
                    }
                    break;
                }
            }
            loaded += run(data, kind);
            total++;
        }
    }
    std::printf("%lu inputs, %lu loaded\n", total, loaded);
    return 0;
}
