// host_assets.cpp — szg/assets.h: glTF 2.0 / GLB -> meshes, surfaces and material maps for the raster passes
// (assets/assets.cpp:406-1092, :1192-1266). CPU only, no third-party code: a small JSON reader, base64, inflate and a
// PNG decoder stand where the reference calls fastgltf (a FetchContent download, not in the checkout) and stb_image
// (vendored under thirdparty/stb; not used or restated here — PNG is decoded from its specification, JPEG is refused).
// Out of the hot path's scope and frozen: it only feeds the rasteriser's tests.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <limits>
#include <memory>
#include <new>
#include <optional>
#include <string>
#include <sys/stat.h>
#include <utility>
#include <vector>

#include "szg/assets.h"
#include "szg_internal.hpp"

namespace
{
using Bytes = std::vector<uint8_t>;

// ---------------------------------------------------------------------------------------------------------------
// JSON (RFC 8259), just enough of a DOM for glTF
// ---------------------------------------------------------------------------------------------------------------
struct Json
{
    enum Type
    {
        Null,
        Bool,
        Number,
        String,
        Array,
        Object
    } type = Null;
    bool boolean = false;
    double number = 0.0;
    std::string string;
    std::vector<Json> array;
    std::vector<std::pair<std::string, Json>> object;

    const Json* get(const char* key) const
    {
        if (type != Object)
        {
            return nullptr;
        }
        for (auto const& kv : object)
        {
            if (kv.first == key)
            {
                return &kv.second;
            }
        }
        return nullptr;
    }
    // a member that is a non-negative integer (glTF indices, counts, offsets)
    std::optional<size_t> index(const char* key) const
    {
        const Json* v = get(key);
        if (v == nullptr || v->type != Number || !(v->number >= 0.0) || v->number > 9.0e15 || v->number != std::floor(v->number))
        {
            return std::nullopt;
        }
        return static_cast<size_t>(v->number);
    }
    std::string text(const char* key) const
    {
        const Json* v = get(key);
        return (v != nullptr && v->type == String) ? v->string : std::string{};
    }
    const std::vector<Json>& list(const char* key) const
    {
        static const std::vector<Json> empty;
        const Json* v = get(key);
        return (v != nullptr && v->type == Array) ? v->array : empty;
    }
};

class JsonParser
{
  public:
    JsonParser(const char* begin, const char* end) : p_(begin), end_(end) {}
    bool parse(Json& out, std::string& error)
    {
        if (end_ - p_ >= 3 && static_cast<uint8_t>(p_[0]) == 0xEF && static_cast<uint8_t>(p_[1]) == 0xBB &&
            static_cast<uint8_t>(p_[2]) == 0xBF)
        {
            p_ += 3; // UTF-8 byte order mark
        }
        if (!value(out, 0))
        {
            error = error_.empty() ? "malformed JSON" : error_;
            return false;
        }
        space();
        if (p_ != end_)
        {
            error = "trailing characters after the JSON document";
            return false;
        }
        return true;
    }

  private:
    const char* p_;
    const char* end_;
    std::string error_;

    void space()
    {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r'))
        {
            p_++;
        }
    }
    bool fail(const char* what)
    {
        if (error_.empty())
        {
            error_ = what;
        }
        return false;
    }
    bool literal(const char* word)
    {
        size_t const n = std::strlen(word);
        if (static_cast<size_t>(end_ - p_) < n || std::memcmp(p_, word, n) != 0)
        {
            return fail("unknown literal");
        }
        p_ += n;
        return true;
    }
    static void utf8(std::string& s, uint32_t cp)
    {
        if (cp < 0x80)
        {
            s += static_cast<char>(cp);
        }
        else if (cp < 0x800)
        {
            s += static_cast<char>(0xC0 | (cp >> 6));
            s += static_cast<char>(0x80 | (cp & 0x3F));
        }
        else if (cp < 0x10000)
        {
            s += static_cast<char>(0xE0 | (cp >> 12));
            s += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            s += static_cast<char>(0x80 | (cp & 0x3F));
        }
        else
        {
            s += static_cast<char>(0xF0 | (cp >> 18));
            s += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
            s += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            s += static_cast<char>(0x80 | (cp & 0x3F));
        }
    }
    bool hex4(uint32_t& out)
    {
        if (end_ - p_ < 4)
        {
            return fail("truncated \\u escape");
        }
        out = 0;
        for (int k = 0; k < 4; k++)
        {
            char const c = *p_++;
            uint32_t d;
            if (c >= '0' && c <= '9')
            {
                d = static_cast<uint32_t>(c - '0');
            }
            else if (c >= 'a' && c <= 'f')
            {
                d = static_cast<uint32_t>(c - 'a' + 10);
            }
            else if (c >= 'A' && c <= 'F')
            {
                d = static_cast<uint32_t>(c - 'A' + 10);
            }
            else
            {
                return fail("bad \\u escape");
            }
            out = out * 16 + d;
        }
        return true;
    }
    bool stringBody(std::string& out)
    {
        p_++; // opening quote
        while (p_ < end_)
        {
            char const c = *p_++;
            if (c == '"')
            {
                return true;
            }
            if (c != '\\')
            {
                out += c;
                continue;
            }
            if (p_ >= end_)
            {
                break;
            }
            char const e = *p_++;
            switch (e)
            {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u':
            {
                uint32_t cp = 0;
                if (!hex4(cp))
                {
                    return false;
                }
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u')
                {
                    p_ += 2;
                    uint32_t low = 0;
                    if (!hex4(low))
                    {
                        return false;
                    }
                    if (low >= 0xDC00 && low < 0xE000)
                    {
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (low - 0xDC00);
                    }
                }
                utf8(out, cp);
                break;
            }
            default: return fail("bad escape in string");
            }
        }
        return fail("unterminated string");
    }
    bool value(Json& out, int depth)
    {
        if (depth > 128)
        {
            return fail("JSON nested too deeply");
        }
        space();
        if (p_ >= end_)
        {
            return fail("unexpected end of JSON");
        }
        char const c = *p_;
        if (c == '{')
        {
            out.type = Json::Object;
            p_++;
            space();
            if (p_ < end_ && *p_ == '}')
            {
                p_++;
                return true;
            }
            for (;;)
            {
                space();
                if (p_ >= end_ || *p_ != '"')
                {
                    return fail("object key expected");
                }
                std::string key;
                if (!stringBody(key))
                {
                    return false;
                }
                space();
                if (p_ >= end_ || *p_ != ':')
                {
                    return fail("':' expected");
                }
                p_++;
                out.object.emplace_back(std::move(key), Json{});
                if (!value(out.object.back().second, depth + 1))
                {
                    return false;
                }
                space();
                if (p_ < end_ && *p_ == ',')
                {
                    p_++;
                    continue;
                }
                if (p_ < end_ && *p_ == '}')
                {
                    p_++;
                    return true;
                }
                return fail("',' or '}' expected");
            }
        }
        if (c == '[')
        {
            out.type = Json::Array;
            p_++;
            space();
            if (p_ < end_ && *p_ == ']')
            {
                p_++;
                return true;
            }
            for (;;)
            {
                out.array.emplace_back();
                if (!value(out.array.back(), depth + 1))
                {
                    return false;
                }
                space();
                if (p_ < end_ && *p_ == ',')
                {
                    p_++;
                    continue;
                }
                if (p_ < end_ && *p_ == ']')
                {
                    p_++;
                    return true;
                }
                return fail("',' or ']' expected");
            }
        }
        if (c == '"')
        {
            out.type = Json::String;
            return stringBody(out.string);
        }
        if (c == 't')
        {
            out.type = Json::Bool;
            out.boolean = true;
            return literal("true");
        }
        if (c == 'f')
        {
            out.type = Json::Bool;
            out.boolean = false;
            return literal("false");
        }
        if (c == 'n')
        {
            out.type = Json::Null;
            return literal("null");
        }
        if (c == '-' || (c >= '0' && c <= '9'))
        {
            const char* q = p_;
            while (q < end_ && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9')))
            {
                q++;
            }
            std::string const token(p_, q);
            char* tail = nullptr;
            out.type = Json::Number;
            out.number = std::strtod(token.c_str(), &tail);
            if (tail == token.c_str() || *tail != '\0')
            {
                return fail("malformed number");
            }
            p_ = q;
            return true;
        }
        return fail("unexpected character in JSON");
    }
};

// ---------------------------------------------------------------------------------------------------------------
// base64 (RFC 4648) and percent-decoding for URIs
// ---------------------------------------------------------------------------------------------------------------
bool base64Decode(const std::string& text, size_t begin, Bytes& out)
{
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = begin; i < text.size(); i++)
    {
        char const c = text[i];
        uint32_t v;
        if (c >= 'A' && c <= 'Z')
        {
            v = static_cast<uint32_t>(c - 'A');
        }
        else if (c >= 'a' && c <= 'z')
        {
            v = static_cast<uint32_t>(c - 'a' + 26);
        }
        else if (c >= '0' && c <= '9')
        {
            v = static_cast<uint32_t>(c - '0' + 52);
        }
        else if (c == '+' || c == '-')
        {
            v = 62;
        }
        else if (c == '/' || c == '_')
        {
            v = 63;
        }
        else if (c == '=' || c == '\n' || c == '\r' || c == ' ')
        {
            continue;
        }
        else
        {
            return false;
        }
        acc = (acc << 6) | v;
        bits += 6;
        if (bits >= 8)
        {
            bits -= 8;
            out.push_back(static_cast<uint8_t>((acc >> bits) & 0xFFu));
        }
    }
    return true;
}

std::string percentDecode(const std::string& s)
{
    auto hex = [](char c) -> int {
        if (c >= '0' && c <= '9')
        {
            return c - '0';
        }
        if (c >= 'a' && c <= 'f')
        {
            return c - 'a' + 10;
        }
        if (c >= 'A' && c <= 'F')
        {
            return c - 'A' + 10;
        }
        return -1;
    };
    std::string out;
    for (size_t i = 0; i < s.size(); i++)
    {
        if (s[i] == '%' && i + 2 < s.size() && hex(s[i + 1]) >= 0 && hex(s[i + 2]) >= 0)
        {
            out += static_cast<char>(hex(s[i + 1]) * 16 + hex(s[i + 2]));
            i += 2;
        }
        else
        {
            out += s[i];
        }
    }
    return out;
}

bool readFile(const std::string& path, Bytes& out)
{
    struct stat st
    {
    };
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode))
    {
        return false;
    }
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open())
    {
        return false;
    }
    out.resize(static_cast<size_t>(st.st_size));
    file.read(reinterpret_cast<char*>(out.data()), static_cast<std::streamsize>(out.size()));
    return static_cast<size_t>(file.gcount()) == out.size();
}

std::string joinPath(const std::string& root, const std::string& relative)
{
    if (!relative.empty() && relative[0] == '/')
    {
        return relative;
    }
    if (root.empty())
    {
        return relative;
    }
    return root.back() == '/' ? root + relative : root + "/" + relative;
}

// ---------------------------------------------------------------------------------------------------------------
// inflate (RFC 1951) inside a zlib stream (RFC 1950); the Adler-32 trailer is not verified (stb_image does not either)
// ---------------------------------------------------------------------------------------------------------------
class Inflater
{
  public:
    Inflater(const uint8_t* data, size_t size) : p_(data), end_(data + size) {}

    bool zlib(Bytes& out, size_t limit)
    {
        out.resize(limit); // written through data_ / size_ (no per-byte push_back); trimmed to what was produced at the end
        data_ = out.data();
        size_ = 0;
        bool const ok = stream(limit);
        out.resize(size_);
        return ok;
    }

  private:
    static constexpr int FAST_BITS = 10;
    struct Huffman
    {
        uint16_t fast[1 << FAST_BITS]; // (length << 9) | symbol, 0 = not a short code
        uint16_t count[16];
        uint16_t symbol[288];
    };

    const uint8_t* p_;
    const uint8_t* end_;
    uint8_t* data_ = nullptr; // output buffer of `limit` bytes and the number of bytes produced so far
    size_t size_ = 0;
    uint64_t hold_ = 0;
    int held_ = 0;    // bits in hold_, including ...
    int phantom_ = 0; // ... zero bits appended past the end of the input (look-ahead of decode())
    Huffman lit_{}, dist_{};

    bool stream(size_t limit)
    {
        if (end_ - p_ < 2)
        {
            return false;
        }
        unsigned const cmf = p_[0], flg = p_[1];
        p_ += 2;
        if ((cmf * 256u + flg) % 31u != 0 || (cmf & 15u) != 8u || (flg & 32u) != 0)
        {
            return false; // bad header, not deflate, or preset dictionary
        }
        for (;;)
        {
            unsigned const last = bits(1);
            unsigned const type = bits(2);
            bool ok;
            if (type == 0)
            {
                ok = stored(limit);
            }
            else if (type == 1)
            {
                fixedTables();
                ok = block(limit);
            }
            else if (type == 2)
            {
                ok = dynamicTables() && block(limit);
            }
            else
            {
                ok = false;
            }
            if (!ok || overrun())
            {
                return false;
            }
            if (last != 0)
            {
                return true;
            }
        }
    }

    // true once bits that are not in the input have been consumed
    bool overrun() const { return held_ < phantom_; }
    void fill(int need)
    {
        while (held_ < need)
        {
            uint64_t byte = 0;
            if (p_ < end_)
            {
                byte = *p_++;
            }
            else
            {
                phantom_ += 8;
            }
            hold_ |= byte << held_;
            held_ += 8;
        }
    }
    unsigned bits(int n)
    {
        if (n == 0)
        {
            return 0;
        }
        fill(n);
        unsigned const v = static_cast<unsigned>(hold_ & ((1ull << n) - 1ull));
        hold_ >>= n;
        held_ -= n;
        return v;
    }
    static bool build(Huffman& h, const uint8_t* lengths, int n)
    {
        std::memset(h.fast, 0, sizeof h.fast);
        std::memset(h.count, 0, sizeof h.count);
        for (int i = 0; i < n; i++)
        {
            h.count[lengths[i]]++;
        }
        h.count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++)
        {
            left = (left << 1) - h.count[len];
            if (left < 0)
            {
                return false; // over-subscribed
            }
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++)
        {
            offs[len + 1] = static_cast<uint16_t>(offs[len] + h.count[len]);
        }
        uint16_t next[16];
        {
            unsigned code = 0;
            for (int len = 1; len < 16; len++)
            {
                code = (code + h.count[len - 1]) << 1;
                next[len] = static_cast<uint16_t>(code);
            }
        }
        for (int i = 0; i < n; i++)
        {
            int const len = lengths[i];
            if (len == 0)
            {
                continue;
            }
            h.symbol[offs[len]++] = static_cast<uint16_t>(i);
            unsigned const code = next[len]++;
            if (len <= FAST_BITS)
            {
                unsigned rev = 0;
                for (int b = 0; b < len; b++)
                {
                    rev |= ((code >> b) & 1u) << (len - 1 - b);
                }
                for (unsigned k = rev; k < (1u << FAST_BITS); k += (1u << len))
                {
                    h.fast[k] = static_cast<uint16_t>((len << 9) | i);
                }
            }
        }
        return true;
    }
    int decode(const Huffman& h)
    {
        fill(15);
        unsigned const entry = h.fast[hold_ & ((1u << FAST_BITS) - 1u)];
        if (entry != 0)
        {
            int const len = static_cast<int>(entry >> 9);
            hold_ >>= len;
            held_ -= len;
            return static_cast<int>(entry & 511u);
        }
        int code = 0, first = 0, index = 0;
        uint64_t window = hold_;
        for (int len = 1; len < 16; len++)
        {
            code |= static_cast<int>(window & 1u);
            window >>= 1;
            int const count = h.count[len];
            if (code - count < first)
            {
                hold_ >>= len;
                held_ -= len;
                return h.symbol[index + (code - first)];
            }
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
    bool stored(size_t limit)
    {
        hold_ >>= (held_ & 7);
        held_ -= (held_ & 7);
        unsigned const len = bits(16);
        unsigned const nlen = bits(16);
        if (overrun() || (len ^ 0xFFFFu) != nlen || size_ + len > limit)
        {
            return false;
        }
        // whole bytes still held in the bit buffer come first
        unsigned remaining = len;
        while (remaining > 0 && held_ >= 8)
        {
            data_[size_++] = static_cast<uint8_t>(hold_ & 0xFFu);
            hold_ >>= 8;
            held_ -= 8;
            remaining--;
        }
        if (static_cast<size_t>(end_ - p_) < remaining)
        {
            return false;
        }
        std::memcpy(data_ + size_, p_, remaining);
        size_ += remaining;
        p_ += remaining;
        return true;
    }
    void fixedTables()
    {
        uint8_t lengths[288];
        for (int i = 0; i < 288; i++)
        {
            lengths[i] = static_cast<uint8_t>(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
        }
        build(lit_, lengths, 288);
        for (int i = 0; i < 30; i++)
        {
            lengths[i] = 5;
        }
        build(dist_, lengths, 30);
    }
    bool dynamicTables()
    {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int const nlen = static_cast<int>(bits(5)) + 257;
        int const ndist = static_cast<int>(bits(5)) + 1;
        int const ncode = static_cast<int>(bits(4)) + 4;
        if (nlen > 286 || ndist > 30)
        {
            return false;
        }
        uint8_t lengths[320];
        std::memset(lengths, 0, sizeof lengths);
        for (int i = 0; i < ncode; i++)
        {
            lengths[order[i]] = static_cast<uint8_t>(bits(3));
        }
        Huffman codeLengths{};
        if (!build(codeLengths, lengths, 19))
        {
            return false;
        }
        std::memset(lengths, 0, sizeof lengths);
        int index = 0;
        while (index < nlen + ndist)
        {
            int const sym = decode(codeLengths);
            if (sym < 0 || overrun())
            {
                return false;
            }
            if (sym < 16)
            {
                lengths[index++] = static_cast<uint8_t>(sym);
                continue;
            }
            int repeat;
            uint8_t value = 0;
            if (sym == 16)
            {
                if (index == 0)
                {
                    return false;
                }
                value = lengths[index - 1];
                repeat = 3 + static_cast<int>(bits(2));
            }
            else if (sym == 17)
            {
                repeat = 3 + static_cast<int>(bits(3));
            }
            else
            {
                repeat = 11 + static_cast<int>(bits(7));
            }
            if (index + repeat > nlen + ndist)
            {
                return false;
            }
            while (repeat-- > 0)
            {
                lengths[index++] = value;
            }
        }
        if (lengths[256] == 0)
        {
            return false; // no end-of-block code
        }
        return build(lit_, lengths, nlen) && build(dist_, lengths + nlen, ndist);
    }
    bool block(size_t limit)
    {
        static const uint16_t lengthBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                                31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lengthExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t distBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                              193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;)
        {
            int const sym = decode(lit_);
            if (sym < 0 || overrun())
            {
                return false;
            }
            if (sym < 256)
            {
                if (size_ >= limit)
                {
                    return false;
                }
                data_[size_++] = static_cast<uint8_t>(sym);
                continue;
            }
            if (sym == 256)
            {
                return true;
            }
            int const li = sym - 257;
            if (li >= 29)
            {
                return false;
            }
            size_t const length = lengthBase[li] + bits(lengthExtra[li]);
            int const di = decode(dist_);
            if (di < 0 || di >= 30)
            {
                return false;
            }
            size_t const distance = distBase[di] + bits(distExtra[di]);
            if (distance > size_ || size_ + length > limit)
            {
                return false;
            }
            const uint8_t* from = data_ + (size_ - distance);
            uint8_t* to = data_ + size_;
            for (size_t k = 0; k < length; k++) // forward byte copy: source and destination may overlap (run-length matches)
            {
                to[k] = from[k];
            }
            size_ += length;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// PNG -> RGBA8 the way stb_image answers a 4-channel request (stbi_load_from_memory(..., 4), assets.cpp:328-335)
// ---------------------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return (uint32_t{p[0]} << 24) | (uint32_t{p[1]} << 16) | (uint32_t{p[2]} << 8) | uint32_t{p[3]}; }
uint32_t be16(const uint8_t* p) { return (uint32_t{p[0]} << 8) | uint32_t{p[1]}; }

bool decodePng(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, Bytes& rgba, std::string& why);

// detail_stbi::loadRGBA (assets.cpp:319-364): the two encodings glTF 2.0 allows, told apart by their signatures
bool decodeImageBytes(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, Bytes& rgba, std::string& why)
{
    if (size >= 2 && data[0] == 0xFF && data[1] == 0xD8)
    {
        // asset IO is outside the hot path: PNG only; a JPEG fails like any undecodable image
        // does in the reference (assets.cpp:336-343: "stbi: Failed to convert image." and the default map)
        why = "JPEG streams are not decoded by this build (PNG only)";
        return false;
    }
    return decodePng(data, size, width, height, rgba, why);
}

bool decodePng(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, Bytes& rgba, std::string& why)
{
    static const uint8_t signature[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (size < 8 || std::memcmp(data, signature, 8) != 0)
    {
        why = "not a PNG (the image encoding this build decodes)";
        return false;
    }
    size_t at = 8;
    bool haveHeader = false, havePalette = false, done = false;
    uint32_t depth = 0, colorType = 0, interlace = 0;
    uint8_t palette[256][4];
    for (auto& entry : palette)
    {
        entry[0] = entry[1] = entry[2] = 0;
        entry[3] = 255;
    }
    size_t paletteSize = 0;
    bool hasKey = false;
    uint32_t key[3] = {0, 0, 0};
    Bytes compressed;
    while (!done)
    {
        if (size - at < 12)
        {
            why = "truncated PNG";
            return false;
        }
        uint32_t const length = be32(data + at);
        const uint8_t* const type = data + at + 4;
        if (length > size - at - 12)
        {
            why = "PNG chunk runs past the end of the file";
            return false;
        }
        const uint8_t* const body = data + at + 8;
        auto is = [&](const char* name) { return std::memcmp(type, name, 4) == 0; };
        if (!haveHeader && !is("IHDR"))
        {
            why = "PNG does not start with IHDR";
            return false;
        }
        if (is("IHDR"))
        {
            if (haveHeader || length != 13)
            {
                why = "bad IHDR";
                return false;
            }
            haveHeader = true;
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colorType = body[9];
            interlace = body[12];
            bool const depthOk = (colorType == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                                 (colorType == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                                 ((colorType == 2 || colorType == 4 || colorType == 6) && (depth == 8 || depth == 16));
            if (width == 0 || height == 0 || width > (1u << 24) || height > (1u << 24) || !depthOk || body[10] != 0 || body[11] != 0 ||
                interlace > 1)
            {
                why = "unsupported PNG header";
                return false;
            }
            if (static_cast<uint64_t>(width) * height > (1ull << 28))
            {
                why = "PNG too large";
                return false;
            }
        }
        else if (is("PLTE"))
        {
            if (length > 768 || length % 3 != 0)
            {
                why = "bad PLTE";
                return false;
            }
            paletteSize = length / 3;
            for (size_t i = 0; i < paletteSize; i++)
            {
                palette[i][0] = body[i * 3];
                palette[i][1] = body[i * 3 + 1];
                palette[i][2] = body[i * 3 + 2];
                palette[i][3] = 255;
            }
            havePalette = true;
        }
        else if (is("tRNS"))
        {
            if (!compressed.empty())
            {
                why = "tRNS after IDAT";
                return false;
            }
            if (colorType == 3)
            {
                if (!havePalette || length > paletteSize)
                {
                    why = "bad tRNS";
                    return false;
                }
                for (size_t i = 0; i < length; i++)
                {
                    palette[i][3] = body[i];
                }
            }
            else if (colorType == 0 || colorType == 2)
            {
                size_t const n = colorType == 0 ? 1 : 3;
                if (length != n * 2)
                {
                    why = "bad tRNS";
                    return false;
                }
                hasKey = true;
                for (size_t k = 0; k < n; k++)
                {
                    key[k] = be16(body + k * 2);
                }
            }
            else
            {
                why = "tRNS with an alpha channel";
                return false;
            }
        }
        else if (is("IDAT"))
        {
            if (colorType == 3 && !havePalette)
            {
                why = "paletted PNG without PLTE";
                return false;
            }
            compressed.insert(compressed.end(), body, body + length);
        }
        else if (is("IEND"))
        {
            done = true;
        }
        else if ((type[0] & 32u) == 0)
        {
            why = "unknown critical PNG chunk";
            return false;
        }
        at += 12 + static_cast<size_t>(length);
    }
    if (!haveHeader || compressed.empty())
    {
        why = "PNG without image data";
        return false;
    }

    uint32_t const channels = colorType == 0 ? 1 : colorType == 2 ? 3 : colorType == 3 ? 1 : colorType == 4 ? 2 : 4;
    uint32_t const bitsPerPixel = channels * depth;
    size_t const filterStride = std::max<size_t>(1, bitsPerPixel / 8);

    struct Pass
    {
        uint32_t x0, y0, dx, dy;
    };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    int const passCount = interlace != 0 ? 7 : 1;
    size_t expected = 0;
    for (int k = 0; k < passCount; k++)
    {
        Pass const& ps = interlace != 0 ? adam7[k] : whole;
        uint32_t const pw = (width - ps.x0 + ps.dx - 1) / ps.dx, ph = (height - ps.y0 + ps.dy - 1) / ps.dy;
        if (width > ps.x0 && height > ps.y0 && pw > 0 && ph > 0)
        {
            expected += (static_cast<size_t>((static_cast<uint64_t>(pw) * bitsPerPixel + 7) / 8) + 1) * ph;
        }
    }
    if (expected / 1032u > compressed.size() + 1u)
    {
        why = "corrupt PNG image data"; // deflate cannot expand by more than 1032 : 1: the header promises more than the data holds
        return false;
    }
    Bytes raw;
    Inflater inflater(compressed.data(), compressed.size());
    if (!inflater.zlib(raw, expected) || raw.size() < expected)
    {
        why = "corrupt PNG image data";
        return false;
    }

    rgba.assign(static_cast<size_t>(width) * height * 4, 0);
    uint32_t const lowDepthScale = depth == 1 ? 0xFFu : depth == 2 ? 0x55u : depth == 4 ? 0x11u : 1u;
    size_t cursor = 0;
    Bytes previous, current;
    for (int k = 0; k < passCount; k++)
    {
        Pass const& ps = interlace != 0 ? adam7[k] : whole;
        if (width <= ps.x0 || height <= ps.y0)
        {
            continue;
        }
        uint32_t const pw = (width - ps.x0 + ps.dx - 1) / ps.dx, ph = (height - ps.y0 + ps.dy - 1) / ps.dy;
        size_t const rowBytes = static_cast<size_t>((static_cast<uint64_t>(pw) * bitsPerPixel + 7) / 8);
        previous.assign(rowBytes, 0);
        current.assign(rowBytes, 0);
        for (uint32_t j = 0; j < ph; j++)
        {
            uint8_t const filter = raw[cursor++];
            const uint8_t* const in = raw.data() + cursor;
            cursor += rowBytes;
            if (filter > 4)
            {
                why = "bad PNG filter";
                return false;
            }
            uint8_t* const cur = current.data();
            const uint8_t* const up = previous.data();
            size_t const head = std::min(filterStride, rowBytes); // bytes without a left neighbour
            switch (filter)
            {
            case 0: std::memcpy(cur, in, rowBytes); break;
            case 1:
                std::memcpy(cur, in, head);
                for (size_t i = head; i < rowBytes; i++)
                {
                    cur[i] = static_cast<uint8_t>(in[i] + cur[i - filterStride]);
                }
                break;
            case 2:
                for (size_t i = 0; i < rowBytes; i++)
                {
                    cur[i] = static_cast<uint8_t>(in[i] + up[i]);
                }
                break;
            case 3:
                for (size_t i = 0; i < head; i++)
                {
                    cur[i] = static_cast<uint8_t>(in[i] + (up[i] >> 1));
                }
                for (size_t i = head; i < rowBytes; i++)
                {
                    cur[i] = static_cast<uint8_t>(in[i] + ((cur[i - filterStride] + up[i]) >> 1));
                }
                break;
            default: // Paeth
                for (size_t i = 0; i < head; i++)
                {
                    cur[i] = static_cast<uint8_t>(in[i] + up[i]); // a = c = 0: the predictor is b
                }
                for (size_t i = head; i < rowBytes; i++)
                {
                    int const a = cur[i - filterStride], b = up[i], c = up[i - filterStride];
                    int const pp = a + b - c;
                    int const pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                    cur[i] = static_cast<uint8_t>(in[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
                }
                break;
            }
            // samples of this row -> RGBA8
            uint8_t* const outRow = rgba.data() + (static_cast<size_t>(ps.y0 + j * ps.dy) * width) * 4;
            if (depth == 8 && ps.dx == 1 && (colorType == 6 || (colorType == 2 && !hasKey)))
            {
                // the common layouts, a row at a time (same values as the general path below)
                if (colorType == 6)
                {
                    std::memcpy(outRow, current.data(), static_cast<size_t>(pw) * 4);
                }
                else
                {
                    const uint8_t* src = current.data();
                    for (uint32_t i = 0; i < pw; i++)
                    {
                        outRow[i * 4] = src[i * 3];
                        outRow[i * 4 + 1] = src[i * 3 + 1];
                        outRow[i * 4 + 2] = src[i * 3 + 2];
                        outRow[i * 4 + 3] = 255;
                    }
                }
                previous.swap(current);
                continue;
            }
            for (uint32_t i = 0; i < pw; i++)
            {
                uint32_t sample[4] = {0, 0, 0, 0};
                for (uint32_t ch = 0; ch < channels; ch++)
                {
                    size_t const n = static_cast<size_t>(i) * channels + ch;
                    if (depth == 16)
                    {
                        sample[ch] = be16(current.data() + n * 2);
                    }
                    else if (depth == 8)
                    {
                        sample[ch] = current[n];
                    }
                    else
                    {
                        size_t const bit = n * depth;
                        sample[ch] = (current[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    }
                }
                uint8_t* const px = outRow + static_cast<size_t>(ps.x0 + i * ps.dx) * 4;
                if (colorType == 3)
                {
                    std::memcpy(px, palette[sample[0] & 255u], 4);
                    continue;
                }
                // 16-bit samples keep their high byte; low bit depths are scaled to 0..255 (stb_image)
                uint32_t v[4];
                for (uint32_t ch = 0; ch < channels; ch++)
                {
                    v[ch] = depth == 16 ? (sample[ch] >> 8) : sample[ch] * lowDepthScale;
                }
                uint8_t alpha = 255;
                if (hasKey)
                {
                    // colour key: compared at 16 bits for 16-bit images, otherwise on the low byte scaled like the samples
                    bool match = true;
                    for (uint32_t ch = 0; ch < channels; ch++)
                    {
                        match = match && (depth == 16 ? sample[ch] == key[ch] : v[ch] == ((key[ch] & 255u) * lowDepthScale));
                    }
                    alpha = match ? 0 : 255;
                }
                switch (colorType)
                {
                case 0:
                    px[0] = px[1] = px[2] = static_cast<uint8_t>(v[0]);
                    px[3] = alpha;
                    break;
                case 2:
                    px[0] = static_cast<uint8_t>(v[0]);
                    px[1] = static_cast<uint8_t>(v[1]);
                    px[2] = static_cast<uint8_t>(v[2]);
                    px[3] = alpha;
                    break;
                case 4:
                    px[0] = px[1] = px[2] = static_cast<uint8_t>(v[0]);
                    px[3] = static_cast<uint8_t>(v[1]);
                    break;
                default:
                    px[0] = static_cast<uint8_t>(v[0]);
                    px[1] = static_cast<uint8_t>(v[1]);
                    px[2] = static_cast<uint8_t>(v[2]);
                    px[3] = static_cast<uint8_t>(v[3]);
                    break;
                }
            }
            previous.swap(current);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// glTF
// ---------------------------------------------------------------------------------------------------------------
struct TextureData
{
    Bytes rgba;
    uint32_t width = 0, height = 0;
    uint32_t srgb = 0;
    std::string name;
    bool present = false;
};
struct MaterialRecord
{
    std::string name;
    TextureData color, normal, orm;
};
struct MeshRecord
{
    std::string name;
    std::vector<szg_vertex_packed> vertices;
    std::vector<uint32_t> indices;
    std::vector<szg_asset_surface> surfaces;
    szg_aabb bounds{};
    int32_t gltfIndex = 0;
};

struct BufferView
{
    size_t buffer = 0, offset = 0, length = 0, stride = 0;
    bool valid = false;
};
struct Accessor
{
    std::optional<size_t> view;
    size_t offset = 0, count = 0;
    unsigned componentType = 0;
    unsigned components = 0; // 1..4 (matrices are never read on this path)
    bool normalized = false;
    const Json* sparse = nullptr;
    bool valid = false;
};

unsigned componentSize(unsigned componentType)
{
    switch (componentType)
    {
    case 5120:
    case 5121: return 1;
    case 5122:
    case 5123: return 2;
    case 5125:
    case 5126: return 4;
    default: return 0;
    }
}

// one component as the double fastgltf's accessor tools would hand on: the stored value, or for `normalized`
// integers the glTF dequantisation (c / max, signed clamped at -1)
double readComponent(const uint8_t* p, unsigned componentType, bool normalized)
{
    switch (componentType)
    {
    case 5120:
    {
        int8_t v;
        std::memcpy(&v, p, 1);
        return normalized ? std::max(static_cast<double>(v) / 127.0, -1.0) : static_cast<double>(v);
    }
    case 5121: return normalized ? static_cast<double>(*p) / 255.0 : static_cast<double>(*p);
    case 5122:
    {
        int16_t v;
        std::memcpy(&v, p, 2);
        return normalized ? std::max(static_cast<double>(v) / 32767.0, -1.0) : static_cast<double>(v);
    }
    case 5123:
    {
        uint16_t v;
        std::memcpy(&v, p, 2);
        return normalized ? static_cast<double>(v) / 65535.0 : static_cast<double>(v);
    }
    case 5125:
    {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return static_cast<double>(v);
    }
    default:
    {
        float v;
        std::memcpy(&v, p, 4);
        return static_cast<double>(v);
    }
    }
}

} // namespace

struct szg_gltf
{
    std::vector<MeshRecord> meshes;
    std::vector<MaterialRecord> materials;
    std::string warnings;
};

namespace
{
class Loader
{
  public:
    Loader(szg_gltf& out, std::string root, uint32_t flags) : out_(out), root_(std::move(root)), flags_(flags) {}

    int load(const uint8_t* bytes, size_t size, bool binary, std::string& error)
    {
        const char* jsonBegin = reinterpret_cast<const char*>(bytes);
        size_t jsonSize = size;
        if (binary)
        {
            // GLB: 12-byte header, then chunks {u32 length, u32 type, payload}; JSON first, BIN optional
            if (size < 20 || std::memcmp(bytes, "glTF", 4) != 0)
            {
                error = "not a GLB container (bad magic)";
                return SZG_ERR_PARSE;
            }
            uint32_t version, total, chunkLength, chunkType;
            std::memcpy(&version, bytes + 4, 4);
            std::memcpy(&total, bytes + 8, 4);
            std::memcpy(&chunkLength, bytes + 12, 4);
            std::memcpy(&chunkType, bytes + 16, 4);
            if (version != 2 || total > size || chunkType != 0x4E4F534Au || chunkLength > total - 20)
            {
                error = "unsupported or truncated GLB container";
                return SZG_ERR_PARSE;
            }
            jsonBegin = reinterpret_cast<const char*>(bytes + 20);
            jsonSize = chunkLength;
            size_t at = 20 + static_cast<size_t>(chunkLength);
            at = (at + 3) & ~size_t{3};
            if (at + 8 <= total)
            {
                uint32_t binLength, binType;
                std::memcpy(&binLength, bytes + at, 4);
                std::memcpy(&binType, bytes + at + 4, 4);
                if (binType == 0x004E4942u && binLength <= total - at - 8)
                {
                    glbBin_.assign(bytes + at + 8, bytes + at + 8 + binLength);
                    haveGlbBin_ = true;
                }
            }
        }
        JsonParser parser(jsonBegin, jsonBegin + jsonSize);
        std::string parseError;
        if (!parser.parse(doc_, parseError) || doc_.type != Json::Object)
        {
            error = "glTF JSON: " + (parseError.empty() ? std::string("not an object") : parseError);
            return SZG_ERR_PARSE;
        }
        const Json* asset = doc_.get("asset");
        if (asset == nullptr || asset->type != Json::Object || asset->text("version").empty())
        {
            error = "glTF: invalid or missing asset field";
            return SZG_ERR_PARSE;
        }
        loadBuffers();
        loadViewsAndAccessors();
        loadMaterials();
        loadMeshes();
        if (jpegRefused_ > 0)
        {
            // (not a line the reference prints: it decodes JPEG through stb_image. Asset IO is outside the hot path and frozen; the
            // summary is there so that a caller notices why materials came out with the default maps.)
            warn("Summary: " + std::to_string(jpegRefused_) + " image(s) of this asset are JPEG streams, which this build does not decode (PNG only); "
                 "the materials that reference them use the default maps.");
        }
        return SZG_OK;
    }

  private:
    szg_gltf& out_;
    std::string root_;
    uint32_t flags_;
    Json doc_;
    Bytes glbBin_;
    bool haveGlbBin_ = false;
    std::vector<Bytes> buffers_;
    std::vector<bool> bufferLoaded_;
    std::vector<BufferView> views_;
    std::vector<Accessor> accessors_;
    // decoded images by glTF image index (decoded at most once)
    struct DecodedImage
    {
        bool tried = false, ok = false;
        uint32_t width = 0, height = 0;
        Bytes rgba;
    };
    std::vector<DecodedImage> images_;
    int jpegRefused_ = 0; // images refused because they are JPEG streams (decodeImageBytes), for the summary warning

    void warn(const std::string& line)
    {
        out_.warnings += line;
        out_.warnings += '\n';
    }

    // a uri is either a base64 data: URI or a path relative to the asset's directory
    bool resolveUri(const std::string& uri, Bytes& out, std::string& path)
    {
        if (uri.compare(0, 5, "data:") == 0)
        {
            size_t const comma = uri.find(',');
            if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma)
            {
                return false;
            }
            path.clear();
            return base64Decode(uri, comma + 1, out);
        }
        if (uri.find("://") != std::string::npos)
        {
            return false; // not a local path
        }
        path = joinPath(root_, percentDecode(uri));
        return readFile(path, out);
    }

    void loadBuffers()
    {
        auto const& list = doc_.list("buffers");
        buffers_.resize(list.size());
        bufferLoaded_.assign(list.size(), false);
        for (size_t i = 0; i < list.size(); i++)
        {
            std::string const uri = list[i].text("uri");
            if (uri.empty())
            {
                if (i == 0 && haveGlbBin_)
                {
                    buffers_[i] = glbBin_;
                    bufferLoaded_[i] = true;
                }
                else
                {
                    warn("glTF buffer " + std::to_string(i) + " has no uri and there is no GLB binary chunk for it.");
                }
                continue;
            }
            std::string path;
            if (resolveUri(uri, buffers_[i], path))
            {
                bufferLoaded_[i] = true;
            }
            else
            {
                buffers_[i].clear();
                warn("glTF buffer " + std::to_string(i) + " could not be loaded from its uri.");
            }
        }
    }

    void loadViewsAndAccessors()
    {
        for (Json const& j : doc_.list("bufferViews"))
        {
            BufferView v;
            auto const buffer = j.index("buffer");
            auto const length = j.index("byteLength");
            v.offset = j.index("byteOffset").value_or(0);
            v.stride = j.index("byteStride").value_or(0);
            if (buffer.has_value() && length.has_value() && buffer.value() < buffers_.size() && bufferLoaded_[buffer.value()] &&
                v.offset <= buffers_[buffer.value()].size() && length.value() <= buffers_[buffer.value()].size() - v.offset)
            {
                v.buffer = buffer.value();
                v.length = length.value();
                v.valid = true;
            }
            views_.push_back(v);
        }
        for (Json const& j : doc_.list("accessors"))
        {
            Accessor a;
            a.view = j.index("bufferView");
            a.offset = j.index("byteOffset").value_or(0);
            a.count = j.index("count").value_or(0);
            a.componentType = static_cast<unsigned>(j.index("componentType").value_or(0));
            const Json* normalized = j.get("normalized");
            a.normalized = normalized != nullptr && normalized->type == Json::Bool && normalized->boolean;
            std::string const type = j.text("type");
            a.components = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
            const Json* sparse = j.get("sparse");
            a.sparse = (sparse != nullptr && sparse->type == Json::Object) ? sparse : nullptr;
            a.valid = componentSize(a.componentType) != 0 && a.components != 0 && j.index("count").has_value() &&
                      a.count <= (size_t{1} << 31);
            accessors_.push_back(a);
        }
    }

    // the bytes of `count` elements of `elementSize` starting `offset` into a view, honouring its stride
    bool gather(size_t viewIndex, size_t offset, size_t count, size_t elementSize, std::vector<const uint8_t*>& out) const
    {
        if (viewIndex >= views_.size() || !views_[viewIndex].valid)
        {
            return false;
        }
        BufferView const& v = views_[viewIndex];
        size_t const stride = v.stride != 0 ? v.stride : elementSize;
        if (count == 0)
        {
            return true;
        }
        // offset + (count - 1) * stride + elementSize <= v.length, without overflow
        if (offset > v.length || elementSize > v.length - offset || (count - 1) > (v.length - offset - elementSize) / std::max<size_t>(stride, 1))
        {
            return false;
        }
        const uint8_t* const base = buffers_[v.buffer].data() + v.offset + offset;
        out.resize(count);
        for (size_t i = 0; i < count; i++)
        {
            out[i] = base + i * stride;
        }
        return true;
    }

    // accessor -> count * components doubles (sparse substitution applied)
    bool readAccessor(size_t index, unsigned wantComponents, std::vector<double>& out, size_t& count)
    {
        if (index >= accessors_.size() || !accessors_[index].valid || accessors_[index].components != wantComponents)
        {
            return false;
        }
        Accessor const& a = accessors_[index];
        unsigned const cs = componentSize(a.componentType);
        size_t const elementSize = static_cast<size_t>(cs) * a.components;
        count = a.count;
        std::vector<const uint8_t*> elements;
        if (a.view.has_value() ? !gather(a.view.value(), a.offset, count, elementSize, elements) : count > (size_t{1} << 24))
        {
            return false; // past its buffer view; or an accessor without storage that asks for more zeros than any mesh holds
        }
        out.assign(count * a.components, 0.0);
        if (a.view.has_value())
        {
            for (size_t i = 0; i < count; i++)
            {
                for (unsigned c = 0; c < a.components; c++)
                {
                    out[i * a.components + c] = readComponent(elements[i] + static_cast<size_t>(c) * cs, a.componentType, a.normalized);
                }
            }
        }
        else if (a.sparse == nullptr)
        {
            return true; // all zeros (glTF 2.0, 5.1.1)
        }
        if (a.sparse != nullptr)
        {
            const Json* indices = a.sparse->get("indices");
            const Json* values = a.sparse->get("values");
            size_t const n = a.sparse->index("count").value_or(0);
            if (indices == nullptr || values == nullptr || n > count)
            {
                return false;
            }
            unsigned const indexType = static_cast<unsigned>(indices->index("componentType").value_or(0));
            unsigned const is = componentSize(indexType);
            if (is == 0 || indexType == 5126 || !indices->index("bufferView").has_value() || !values->index("bufferView").has_value())
            {
                return false;
            }
            std::vector<const uint8_t*> indexElements, valueElements;
            if (!gather(indices->index("bufferView").value(), indices->index("byteOffset").value_or(0), n, is, indexElements) ||
                !gather(values->index("bufferView").value(), values->index("byteOffset").value_or(0), n, elementSize, valueElements))
            {
                return false;
            }
            for (size_t k = 0; k < n; k++)
            {
                double const where = readComponent(indexElements[k], indexType, false);
                if (!(where >= 0.0) || where >= static_cast<double>(count))
                {
                    return false;
                }
                size_t const i = static_cast<size_t>(where);
                for (unsigned c = 0; c < a.components; c++)
                {
                    out[i * a.components + c] = readComponent(valueElements[k] + static_cast<size_t>(c) * cs, a.componentType, a.normalized);
                }
            }
        }
        return true;
    }

    // ---- images and materials (assets.cpp:434-879)
    bool decodeImage(size_t imageIndex, const std::string& what)
    {
        DecodedImage& d = images_[imageIndex];
        if (d.tried)
        {
            return d.ok;
        }
        d.tried = true;
        Json const& image = doc_.list("images")[imageIndex];
        Bytes bytes;
        std::string const uri = image.text("uri");
        if (!uri.empty())
        {
            std::string path;
            bool const isData = uri.compare(0, 5, "data:") == 0;
            if (!resolveUri(uri, bytes, path))
            {
                if (isData)
                {
                    warn("Failed to load image from glTF.");
                }
                else
                {
                    warn("glTF image source URI does not result in a valid file path. URI was: " + uri + ". Full path is: " + path);
                }
                return false;
            }
        }
        else if (image.index("bufferView").has_value())
        {
            if ((flags_ & SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES) == 0)
            {
                warn("Unsupported glTF image source found.");
                return false;
            }
            size_t const view = image.index("bufferView").value();
            if (view >= views_.size() || !views_[view].valid)
            {
                warn("Failed to load image from glTF.");
                return false;
            }
            const uint8_t* const base = buffers_[views_[view].buffer].data() + views_[view].offset;
            bytes.assign(base, base + views_[view].length);
        }
        else
        {
            warn("Unsupported glTF image source found.");
            return false;
        }
        std::string why;
        if (!decodeImageBytes(bytes.data(), bytes.size(), d.width, d.height, d.rgba, why))
        {
            warn("stbi: Failed to convert image. (" + what + ": " + why + ")");
            warn("Failed to load image from glTF.");
            jpegRefused_ += (bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) ? 1 : 0;
            return false;
        }
        d.ok = true;
        return true;
    }

    struct Overrides
    {
        std::optional<uint8_t> red, green, blue, alpha;
    };

    // uploadTextureFromIndex (assets.cpp:668-733) minus the upload
    bool textureFromIndex(size_t textureIndex, Overrides const& overrides, const std::string& materialName, const char* kind,
                          bool srgb, TextureData& out)
    {
        auto const& textures = doc_.list("textures");
        auto const& images = doc_.list("images");
        if (textureIndex >= textures.size())
        {
            warn("Out of bounds texture index.");
            return false;
        }
        // getTextureSources (assets.cpp:434-468)
        auto const source = textures[textureIndex].index("source");
        if (!source.has_value())
        {
            warn("Texture " + textures[textureIndex].text("name") + " was missing imageIndex.");
            warn("Texture index source was not loaded.");
            return false;
        }
        if (source.value() >= textures.size() || source.value() >= images.size())
        {
            // the reference compares against the TEXTURE count (assets.cpp:456) and then indexes the images
            warn("Texture " + textures[textureIndex].text("name") + " had imageIndex that was out of bounds.");
            warn("Texture index source was not loaded.");
            return false;
        }
        size_t const imageIndex = source.value();
        if (!decodeImage(imageIndex, std::string(kind) + " map of material '" + materialName + "'"))
        {
            return false;
        }
        DecodedImage const& d = images_[imageIndex];
        out.rgba = d.rgba;
        out.width = d.width;
        out.height = d.height;
        out.srgb = srgb ? 1u : 0u;
        // channel overrides, assets.cpp:551-573
        for (size_t i = 0; i + 3 < out.rgba.size(); i += 4)
        {
            if (overrides.red.has_value())
            {
                out.rgba[i] = overrides.red.value();
            }
            if (overrides.green.has_value())
            {
                out.rgba[i + 1] = overrides.green.value();
            }
            if (overrides.blue.has_value())
            {
                out.rgba[i + 2] = overrides.blue.value();
            }
            if (overrides.alpha.has_value())
            {
                out.rgba[i + 3] = overrides.alpha.value();
            }
        }
        std::string name = images[imageIndex].text("name");
        if (name.empty())
        {
            name = materialName + "_" + std::to_string(textureIndex) + "_" + kind; // assets.cpp:717-724
        }
        out.name = "texture_" + name; // assets.cpp:310
        out.present = true;
        return true;
    }

    static std::optional<size_t> textureIndexOf(const Json* info)
    {
        if (info == nullptr || info->type != Json::Object)
        {
            return std::nullopt;
        }
        return info->index("index");
    }

    void loadMaterials()
    {
        images_.resize(doc_.list("images").size());
        for (Json const& material : doc_.list("materials"))
        {
            out_.materials.emplace_back();
            MaterialRecord& record = out_.materials.back();
            record.name = material.text("name");
            // parseMaterialIndices, assets.cpp:579-645
            const Json* pbr = material.get("pbrMetallicRoughness");
            std::optional<size_t> const color = textureIndexOf(pbr != nullptr ? pbr->get("baseColorTexture") : nullptr);
            std::optional<size_t> const roughnessMetallic = textureIndexOf(pbr != nullptr ? pbr->get("metallicRoughnessTexture") : nullptr);
            std::optional<size_t> const normal = textureIndexOf(material.get("normalTexture"));
            std::optional<size_t> const occlusion = textureIndexOf(material.get("occlusionTexture"));
            if (!color.has_value())
            {
                warn("Material " + record.name + ": Missing color texture.");
            }
            if (!normal.has_value())
            {
                warn("Material " + record.name + ": Missing normal texture.");
            }
            if (!occlusion.has_value())
            {
                warn("Material " + record.name + ": Missing occlusion texture.");
            }
            if (!roughnessMetallic.has_value())
            {
                warn("Material " + record.name + ": Missing metallicRoughness texture");
            }
            // assets.cpp:763-815
            if (roughnessMetallic.has_value() || occlusion.has_value())
            {
                size_t index;
                Overrides overrides;
                if (occlusion.has_value() && occlusion != roughnessMetallic)
                {
                    warn("Material " + record.name + ": occlusion and roughnessMetallic textures differ. Loading roughnessMetallic and "
                                                     "overriding its occlusion channel.");
                }
                if (roughnessMetallic.has_value())
                {
                    index = roughnessMetallic.value();
                    overrides.red = 255;
                }
                else
                {
                    index = occlusion.value();
                    overrides.green = 0;
                    overrides.blue = 0;
                }
                if (!textureFromIndex(index, overrides, record.name, "orm", false, record.orm))
                {
                    warn("Material " + record.name + ": Failed to upload ORM texture.");
                }
            }
            if (color.has_value() && !textureFromIndex(color.value(), {}, record.name, "color", true, record.color))
            {
                warn("Material " + record.name + ": Failed to upload color texture.");
            }
            if (normal.has_value() && !textureFromIndex(normal.value(), {}, record.name, "normal", false, record.normal))
            {
                warn("Material " + record.name + ": Failed to upload normal texture.");
            }
        }
    }

    // ---- meshes (assets.cpp:887-1092)
    void loadMeshes()
    {
        auto const& meshes = doc_.list("meshes");
        for (size_t meshIndex = 0; meshIndex < meshes.size(); meshIndex++)
        {
            Json const& mesh = meshes[meshIndex];
            MeshRecord record;
            record.name = "mesh_" + mesh.text("name");
            record.gltfIndex = static_cast<int32_t>(meshIndex);
            for (Json const& primitive : mesh.list("primitives"))
            {
                const Json* attributes = primitive.get("attributes");
                auto const indicesAccessor = primitive.index("indices");
                if (!indicesAccessor.has_value() || indicesAccessor.value() >= accessors_.size())
                {
                    warn("glTF mesh primitive had no valid indices accessor. It will be skipped.");
                    continue;
                }
                auto const positionAccessor = attributes != nullptr ? attributes->index("POSITION") : std::nullopt;
                if (!positionAccessor.has_value())
                {
                    warn("glTF mesh primitive had no valid vertices accessor. It will be skipped.");
                    continue;
                }
                std::vector<double> positions, indexValues;
                size_t positionCount = 0, indexCount = 0;
                if (!readAccessor(positionAccessor.value(), 3, positions, positionCount) ||
                    !readAccessor(indicesAccessor.value(), 1, indexValues, indexCount))
                {
                    warn("glTF mesh primitive has a POSITION or indices accessor that cannot be read (out of range, wrong type). It will "
                         "be skipped.");
                    continue;
                }
                if (primitive.index("mode").value_or(4) != 4)
                {
                    warn("Loading glTF mesh primitive as Triangles mode when it is not.");
                }

                szg_asset_surface surface{};
                surface.first_index = static_cast<uint32_t>(record.indices.size());
                surface.index_count = static_cast<uint32_t>(indexCount);
                surface.material = -1;
                auto const materialIndex = primitive.index("material");
                if (!materialIndex.has_value())
                {
                    warn("Mesh " + mesh.text("name") + " has a primitive that is missing material index.");
                }
                else if (materialIndex.value() >= out_.materials.size())
                {
                    warn("Mesh " + mesh.text("name") + " has a primitive with out of bounds material index.");
                }
                else
                {
                    surface.material = static_cast<int32_t>(materialIndex.value());
                }
                record.surfaces.push_back(surface);

                size_t const initialVertexIndex = record.vertices.size();
                for (double const v : indexValues)
                {
                    // fastgltf converts the stored component to uint32_t; the rebasing add wraps like the reference's
                    record.indices.push_back(static_cast<uint32_t>(static_cast<int64_t>(v)) + static_cast<uint32_t>(initialVertexIndex));
                }
                for (size_t i = 0; i < positionCount; i++)
                {
                    szg_vertex_packed vertex{};
                    vertex.position[0] = static_cast<float>(positions[i * 3]);
                    vertex.position[1] = static_cast<float>(positions[i * 3 + 1]);
                    vertex.position[2] = static_cast<float>(positions[i * 3 + 2]);
                    vertex.uv_x = 0.0f;
                    vertex.normal[0] = 1.0f;
                    vertex.normal[1] = 0.0f;
                    vertex.normal[2] = 0.0f;
                    vertex.uv_y = 0.0f;
                    vertex.color[0] = vertex.color[1] = vertex.color[2] = vertex.color[3] = 1.0f;
                    record.vertices.push_back(vertex);
                }
                auto attribute = [&](const char* name, unsigned components, std::vector<double>& values, size_t& count) -> bool {
                    auto const accessor = attributes->index(name);
                    if (!accessor.has_value())
                    {
                        return false;
                    }
                    if (!readAccessor(accessor.value(), components, values, count))
                    {
                        return false;
                    }
                    if (count > positionCount)
                    {
                        warn(std::string("glTF mesh primitive attribute ") + name + " is longer than POSITION; the excess is ignored.");
                        count = positionCount;
                    }
                    return true;
                };
                std::vector<double> values;
                size_t count = 0;
                if (attribute("NORMAL", 3, values, count))
                {
                    for (size_t i = 0; i < count; i++)
                    {
                        szg_vertex_packed& vertex = record.vertices[initialVertexIndex + i];
                        vertex.normal[0] = static_cast<float>(values[i * 3]);
                        vertex.normal[1] = static_cast<float>(values[i * 3 + 1]);
                        vertex.normal[2] = static_cast<float>(values[i * 3 + 2]);
                    }
                }
                else if (attributes->index("NORMAL").has_value())
                {
                    warn("glTF mesh primitive attribute NORMAL cannot be read as VEC3; the default normal is kept.");
                }
                if (attribute("TEXCOORD_0", 2, values, count))
                {
                    for (size_t i = 0; i < count; i++)
                    {
                        szg_vertex_packed& vertex = record.vertices[initialVertexIndex + i];
                        vertex.uv_x = static_cast<float>(values[i * 2]);
                        vertex.uv_y = static_cast<float>(values[i * 2 + 1]);
                    }
                }
                else if (attributes->index("TEXCOORD_0").has_value())
                {
                    warn("glTF mesh primitive attribute TEXCOORD_0 cannot be read as VEC2; uv 0 is kept.");
                }
                if (attribute("COLOR_0", 4, values, count))
                {
                    for (size_t i = 0; i < count; i++)
                    {
                        szg_vertex_packed& vertex = record.vertices[initialVertexIndex + i];
                        for (int c = 0; c < 4; c++)
                        {
                            vertex.color[c] = static_cast<float>(values[i * 4 + static_cast<size_t>(c)]);
                        }
                    }
                }
                else if (attribute("COLOR_0", 3, values, count))
                {
                    // the reference reads COLOR_0 as vec4 only (assets.cpp:1031-1042); an RGB accessor gets alpha 1 here
                    for (size_t i = 0; i < count; i++)
                    {
                        szg_vertex_packed& vertex = record.vertices[initialVertexIndex + i];
                        for (int c = 0; c < 3; c++)
                        {
                            vertex.color[c] = static_cast<float>(values[i * 3 + static_cast<size_t>(c)]);
                        }
                        vertex.color[3] = 1.0f;
                    }
                }
                else if (attributes->index("COLOR_0").has_value())
                {
                    warn("glTF mesh primitive attribute COLOR_0 cannot be read as VEC3 or VEC4; colour 1 is kept.");
                }
            }

            // FLIP_Y, assets.cpp:1046-1054
            for (szg_vertex_packed& vertex : record.vertices)
            {
                vertex.normal[1] *= -1.0f;
                vertex.position[1] *= -1.0f;
            }
            if (record.surfaces.empty())
            {
                continue; // assets.cpp:1056-1059: the mesh stays nullptr and is not registered
            }
            float minimum[3], maximum[3];
            for (int c = 0; c < 3; c++)
            {
                minimum[c] = std::numeric_limits<float>::max();
                maximum[c] = std::numeric_limits<float>::lowest();
            }
            for (szg_vertex_packed const& vertex : record.vertices)
            {
                for (int c = 0; c < 3; c++)
                {
                    // glm::min(x, y) = (y < x) ? y : x with x the vertex (assets.cpp:1066-1067)
                    minimum[c] = (minimum[c] < vertex.position[c]) ? minimum[c] : vertex.position[c];
                    maximum[c] = (vertex.position[c] < maximum[c]) ? maximum[c] : vertex.position[c];
                }
            }
            szg_aabb_create(minimum, maximum, &record.bounds);
            out_.meshes.push_back(std::move(record));
        }
    }
};

int finish(int status, const std::string& error, std::unique_ptr<szg_gltf>& asset, szg_gltf** out)
{
    if (status != SZG_OK)
    {
        szg::set_last_error(error.c_str());
        return status;
    }
    *out = asset.release();
    return SZG_OK;
}

void fillTexture(const TextureData& t, szg_asset_texture* out)
{
    out->rgba = t.present ? t.rgba.data() : nullptr;
    out->width = t.present ? t.width : 0;
    out->height = t.present ? t.height : 0;
    out->srgb = t.srgb;
    out->name = t.name.c_str();
}
} // namespace

extern "C" {

int szg_gltf_load_memory(const void* bytes, size_t size, int is_glb, const char* asset_root, uint32_t flags, szg_gltf** out)
{
    if (bytes == nullptr || out == nullptr || size == 0)
    {
        szg::set_last_error("szg_gltf_load_memory: NULL or empty input");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    std::unique_ptr<szg_gltf> asset(new (std::nothrow) szg_gltf);
    if (!asset)
    {
        szg::set_last_error("szg_gltf_load_memory: out of memory");
        return SZG_ERR_OUT_OF_MEMORY;
    }
    std::string error;
    int status;
    try
    {
        Loader loader(*asset, asset_root != nullptr ? asset_root : "", flags);
        status = loader.load(static_cast<const uint8_t*>(bytes), size, is_glb != 0, error);
    }
    catch (const std::exception&) // bad_alloc / length_error from the containers
    {
        status = SZG_ERR_OUT_OF_MEMORY;
        error = "szg_gltf_load: out of memory";
    }
    return finish(status, error, asset, out);
}

int szg_gltf_load_file(const char* path, uint32_t flags, szg_gltf** out)
{
    if (path == nullptr || out == nullptr)
    {
        szg::set_last_error("szg_gltf_load_file: NULL argument");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    Bytes bytes;
    std::string const p(path);
    if (!readFile(p, bytes) || bytes.empty())
    {
        szg::set_last_error(("Unable to open file at " + p).c_str()); // assets.cpp:1103-1113
        return SZG_ERR_IO;
    }
    size_t const slash = p.find_last_of('/');
    std::string const root = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : p.substr(0, slash));
    bool const json = p.size() >= 5 && p.compare(p.size() - 5, 5, ".gltf") == 0; // assets.cpp:422
    return szg_gltf_load_memory(bytes.data(), bytes.size(), json ? 0 : 1, root.c_str(), flags, out);
}

void szg_gltf_destroy(szg_gltf* asset) { delete asset; }

uint32_t szg_gltf_mesh_count(const szg_gltf* asset) { return asset != nullptr ? static_cast<uint32_t>(asset->meshes.size()) : 0; }

int szg_gltf_mesh(const szg_gltf* asset, uint32_t index, szg_asset_mesh* out)
{
    if (asset == nullptr || out == nullptr || index >= asset->meshes.size())
    {
        szg::set_last_error("szg_gltf_mesh: NULL argument or index out of range");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    MeshRecord const& m = asset->meshes[index];
    out->name = m.name.c_str();
    out->vertices = m.vertices.data();
    out->vertex_count = static_cast<uint32_t>(m.vertices.size());
    out->indices = m.indices.data();
    out->index_count = static_cast<uint32_t>(m.indices.size());
    out->surfaces = m.surfaces.data();
    out->surface_count = static_cast<uint32_t>(m.surfaces.size());
    out->vertex_bounds = m.bounds;
    out->gltf_mesh_index = m.gltfIndex;
    return SZG_OK;
}

uint32_t szg_gltf_material_count(const szg_gltf* asset) { return asset != nullptr ? static_cast<uint32_t>(asset->materials.size()) : 0; }

int szg_gltf_material(const szg_gltf* asset, uint32_t index, szg_asset_material* out)
{
    if (asset == nullptr || out == nullptr || index >= asset->materials.size())
    {
        szg::set_last_error("szg_gltf_material: NULL argument or index out of range");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    MaterialRecord const& m = asset->materials[index];
    out->name = m.name.c_str();
    fillTexture(m.color, &out->color);
    fillTexture(m.normal, &out->normal);
    fillTexture(m.orm, &out->orm);
    return SZG_OK;
}

const char* szg_gltf_warnings(const szg_gltf* asset) { return asset != nullptr ? asset->warnings.c_str() : ""; }

int szg_default_material_map(int kind, uint8_t* rgba)
{
    if (rgba == nullptr || kind < SZG_MAP_COLOR || kind > SZG_MAP_ORM)
    {
        szg::set_last_error("szg_default_material_map: NULL buffer or unknown kind");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    int const n = SZG_DEFAULT_MAP_DIMENSIONS;
    for (int y = 0; y < n; y++)
    {
        for (int x = 0; x < n; x++)
        {
            uint8_t* const px = rgba + (static_cast<size_t>(y) * n + x) * 4;
            if (kind == SZG_MAP_COLOR)
            {
                // assets.cpp:1330-1356: squares of 4 x 4 texels, light (200) where the square indices sum to an even number
                uint8_t const grey = (((x / 4) + (y / 4)) % 2 == 0) ? 200 : 100;
                px[0] = px[1] = px[2] = grey;
                px[3] = 255;
            }
            else if (kind == SZG_MAP_NORMAL)
            {
                px[0] = 127; // assets.cpp:1375-1379
                px[1] = 127;
                px[2] = 255;
                px[3] = 0;
            }
            else
            {
                px[0] = 255; // assets.cpp:1309-1313
                px[1] = 60;
                px[2] = 0;
                px[3] = 0;
            }
        }
    }
    return SZG_OK;
}

int szg_default_mesh(int kind, szg_asset_mesh* out)
{
    if (out == nullptr || (kind != SZG_DEFAULT_MESH_CUBE && kind != SZG_DEFAULT_MESH_PLANE))
    {
        szg::set_last_error("szg_default_mesh: NULL output or unknown kind");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    struct Builtin
    {
        std::vector<szg_vertex_packed> vertices;
        std::vector<uint32_t> indices;
        szg_asset_surface surface{};
        szg_aabb bounds{};
        // one surface over all indices with the default material; bounds = AABB::create(min, max) of the positions
        void finish()
        {
            float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
            float hi[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
            for (szg_vertex_packed const& v : vertices)
            {
                for (int k = 0; k < 3; k++)
                {
                    lo[k] = v.position[k] < lo[k] ? v.position[k] : lo[k];
                    hi[k] = v.position[k] > hi[k] ? v.position[k] : hi[k];
                }
            }
            szg_aabb_create(lo, hi, &bounds);
            surface.first_index = 0;
            surface.index_count = static_cast<uint32_t>(indices.size());
            surface.material = -1;
        }
    };
    static const Builtin builtins[2] = {
        [] {
            // assets.cpp:1476-1570: addCubeFace(uvOrigin, uvX, uvY, normal), six times
            struct Face
            {
                float o[3], ex[3], ey[3], n[3];
            };
            static const Face faces[6] = {{{-1, -1, 1}, {2, 0, 0}, {0, 0, -2}, {0, -1, 0}}, {{-1, 1, -1}, {2, 0, 0}, {0, 0, 2}, {0, 1, 0}},
                                          {{1, -1, -1}, {0, 0, 2}, {0, 2, 0}, {1, 0, 0}},   {{-1, -1, 1}, {0, 0, -2}, {0, 2, 0}, {-1, 0, 0}},
                                          {{-1, -1, -1}, {2, 0, 0}, {0, 2, 0}, {0, 0, -1}}, {{1, -1, 1}, {-2, 0, 0}, {0, 2, 0}, {0, 0, 1}}};
            Builtin b;
            for (Face const& f : faces)
            {
                uint32_t const start = static_cast<uint32_t>(b.vertices.size());
                static const float corner[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
                for (auto const& c : corner)
                {
                    szg_vertex_packed v{};
                    for (int k = 0; k < 3; k++)
                    {
                        // uvOrigin, uvOrigin + uvX, uvOrigin + uvX + uvY, uvOrigin + uvY: sums of small integers, exact
                        v.position[k] = f.o[k] + c[0] * f.ex[k] + c[1] * f.ey[k];
                        v.normal[k] = f.n[k];
                    }
                    v.uv_x = c[0];
                    v.uv_y = c[1];
                    b.vertices.push_back(v);
                }
                for (uint32_t i : {0u, 1u, 2u, 0u, 2u, 3u})
                {
                    b.indices.push_back(start + i);
                }
            }
            b.finish();
            return b;
        }(),
        [] {
            // assets.cpp:1401-1434
            Builtin b;
            static const float corner[4][4] = {{-1, 1, 0, 0}, {1, 1, 1, 0}, {1, -1, 1, 1}, {-1, -1, 0, 1}};
            for (auto const& c : corner)
            {
                szg_vertex_packed v{};
                v.position[0] = c[0];
                v.position[1] = 0.0f;
                v.position[2] = c[1];
                v.uv_x = c[2];
                v.uv_y = c[3];
                v.normal[1] = -1.0f;
                v.color[0] = v.color[1] = v.color[2] = v.color[3] = 1.0f;
                b.vertices.push_back(v);
            }
            b.indices = {0, 1, 3, 1, 2, 3};
            b.finish();
            return b;
        }()};
    Builtin const& b = builtins[kind];
    out->name = kind == SZG_DEFAULT_MESH_CUBE ? "mesh_Cube" : "mesh_Plane";
    out->vertices = b.vertices.data();
    out->vertex_count = static_cast<uint32_t>(b.vertices.size());
    out->indices = b.indices.data();
    out->index_count = static_cast<uint32_t>(b.indices.size());
    out->surfaces = &b.surface;
    out->surface_count = 1;
    out->vertex_bounds = b.bounds;
    out->gltf_mesh_index = -1;
    return SZG_OK;
}

int szg_decode_image_rgba(const void* bytes, size_t size, uint32_t* out_width, uint32_t* out_height, uint8_t** out_rgba)
{
    if (bytes == nullptr || out_width == nullptr || out_height == nullptr || out_rgba == nullptr)
    {
        szg::set_last_error("szg_decode_image_rgba: NULL argument");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    *out_rgba = nullptr;
    try
    {
        Bytes rgba;
        std::string why;
        if (!decodeImageBytes(static_cast<const uint8_t*>(bytes), size, *out_width, *out_height, rgba, why))
        {
            szg::set_last_error(("stbi: Failed to convert image. (" + why + ")").c_str());
            return SZG_ERR_PARSE;
        }
        uint8_t* const copy = static_cast<uint8_t*>(std::malloc(rgba.size()));
        if (copy == nullptr)
        {
            szg::set_last_error("szg_decode_image_rgba: out of memory");
            return SZG_ERR_OUT_OF_MEMORY;
        }
        std::memcpy(copy, rgba.data(), rgba.size());
        *out_rgba = copy;
        return SZG_OK;
    }
    catch (const std::exception&)
    {
        szg::set_last_error("szg_decode_image_rgba: out of memory");
        return SZG_ERR_OUT_OF_MEMORY;
    }
}

int szg_load_image_file_rgba(const char* path, uint32_t* out_width, uint32_t* out_height, uint8_t** out_rgba)
{
    if (path == nullptr || out_width == nullptr || out_height == nullptr || out_rgba == nullptr)
    {
        szg::set_last_error("szg_load_image_file_rgba: NULL argument");
        return SZG_ERR_INVALID_ARGUMENT;
    }
    *out_rgba = nullptr;
    Bytes bytes;
    if (!readFile(path, bytes) || bytes.empty())
    {
        szg::set_last_error("Failed to open file for texture."); // assets.cpp:1143-1147
        return SZG_ERR_IO;
    }
    return szg_decode_image_rgba(bytes.data(), bytes.size(), out_width, out_height, out_rgba);
}

void szg_free_rgba(uint8_t* rgba) { std::free(rgba); }

} // extern "C"
