// host_jpeg.cpp — baseline / extended-sequential / progressive Huffman JPEG -> RGBA8 for szg/assets.h, the second image encoding
// glTF 2.0 allows. The reference decodes through stb_image (assets.cpp:319-364, stbi_load_from_memory(..., 4)), which is
// not under /root/reference; JPEG leaves the inverse DCT, the chroma upsampling filter and the colour conversion to the
// decoder, so stb_image's published integer arithmetic is what is restated here:
//   inverse DCT        the 12-bit fixed-point "islow" butterfly, columns then rows (+512 >> 10, then +65536+(128<<17) >> 17)
//   upsampling         2x2 and 2x1: the 3:1 triangle filter with its edge rules; 1x2: (3 near + far + 2) >> 2; else nearest
//   YCbCr -> RGB       20-bit fixed point, the Cb term of green masked to its high 16 bits
// Parity unpinned (no stb_image, no JPEG asset in the checkout): tests/test_assets.py checks against an independent numpy
// restatement of the same arithmetic and against the source picture within JPEG's own error.
// Progressive files (SOF2: spectral selection and successive approximation, T.81 annex G) accumulate their coefficients
// over the scans and are dequantised and transformed at the end, as stb_image does.
// Four-component Adobe files (CMYK, YCCK) are mapped to RGB the way stb_image does.
// Not decoded: arithmetic coding, lossless / hierarchical, 12-bit — such files fail like any undecodable image.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "szg_internal.hpp"

namespace
{
const uint8_t DEZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman
{
    bool present = false;
    uint8_t fastLength[512]; // codes of up to 9 bits: length (0 = longer) and symbol by the next 9 bits of the stream
    uint8_t fastSymbol[512];
    int maxCode[18]; // largest code of each length, -1 if none
    int firstCode[17];
    int firstIndex[17];
    uint8_t symbols[256];
    int count = 0;
};

bool buildHuffman(Huffman& h, const uint8_t counts[16], const uint8_t* symbols, int total)
{
    if (total > 256)
    {
        return false;
    }
    std::memset(h.fastLength, 0, sizeof h.fastLength);
    std::memcpy(h.symbols, symbols, static_cast<size_t>(total));
    h.count = total;
    int code = 0, index = 0;
    for (int len = 1; len <= 16; len++)
    {
        h.firstCode[len] = code;
        h.firstIndex[len] = index;
        for (int k = 0; k < counts[len - 1]; k++)
        {
            if (code >= (1 << len))
            {
                return false; // over-subscribed
            }
            if (len <= 9)
            {
                int const base = code << (9 - len);
                for (int fill = 0; fill < (1 << (9 - len)); fill++)
                {
                    h.fastLength[base + fill] = static_cast<uint8_t>(len);
                    h.fastSymbol[base + fill] = symbols[index];
                }
            }
            code++;
            index++;
        }
        h.maxCode[len] = counts[len - 1] != 0 ? code - 1 : -1;
        code <<= 1;
    }
    h.maxCode[17] = 0x7fffffff;
    h.present = true;
    return true;
}

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;
    int x = 0, y = 0;   // size in samples
    int w2 = 0, h2 = 0; // allocated size: whole MCUs
    int prediction = 0;
    std::vector<uint8_t> data;
    std::vector<int16_t> coefficients; // progressive only: 64 per block, blocks in rows of coefficientWidth
    int coefficientWidth = 0, coefficientHeight = 0;
};

class Decoder
{
  public:
    Decoder(const uint8_t* data, size_t size) : p_(data), end_(data + size) {}

    bool decode(uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& why)
    {
        if (end_ - p_ < 4 || p_[0] != 0xFF || p_[1] != 0xD8)
        {
            why = "not a JPEG";
            return false;
        }
        p_ += 2;
        bool haveFrame = false, sawScan = false;
        for (;;)
        {
            int const marker = nextMarker();
            if (marker < 0)
            {
                why = "JPEG ends before its end-of-image marker";
                break; // lenient like stb_image: use what was decoded
            }
            if (marker == 0xD9)
            {
                break;
            }
            if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2)
            {
                progressive_ = marker == 0xC2;
                if (haveFrame || !frameHeader(why))
                {
                    if (why.empty())
                    {
                        why = "second frame header";
                    }
                    return false;
                }
                haveFrame = true;
            }
            else if ((marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) || marker == 0xC8)
            {
                why = "unsupported JPEG coding process (lossless, hierarchical or arithmetic)";
                return false;
            }
            else if (marker == 0xC4)
            {
                if (!huffmanTables(why))
                {
                    return false;
                }
            }
            else if (marker == 0xDB)
            {
                if (!quantTables(why))
                {
                    return false;
                }
            }
            else if (marker == 0xDD)
            {
                const uint8_t* seg;
                int len;
                if (!segment(seg, len) || len != 2)
                {
                    why = "bad DRI";
                    return false;
                }
                restartInterval_ = (seg[0] << 8) | seg[1];
            }
            else if (marker == 0xDA)
            {
                if (!haveFrame)
                {
                    why = "scan before the frame header";
                    return false;
                }
                if (!scan(why))
                {
                    return false;
                }
                sawScan = true;
            }
            else if (marker == 0xE0 || marker == 0xEE)
            {
                const uint8_t* seg;
                int len;
                if (!segment(seg, len))
                {
                    why = "truncated JPEG segment";
                    return false;
                }
                if (marker == 0xE0 && len >= 5 && std::memcmp(seg, "JFIF", 5) == 0)
                {
                    jfif_ = true;
                }
                if (marker == 0xEE && len >= 12 && std::memcmp(seg, "Adobe", 6) == 0)
                {
                    adobeTransform_ = seg[11];
                }
            }
            else if ((marker >= 0xE0 && marker <= 0xEF) || marker == 0xFE || marker == 0xDC)
            {
                const uint8_t* seg;
                int len;
                if (!segment(seg, len))
                {
                    why = "truncated JPEG segment";
                    return false;
                }
            }
            else if (marker >= 0xD0 && marker <= 0xD7)
            {
                // a restart marker outside a scan: ignored
            }
            else
            {
                why = "unknown JPEG marker";
                return false;
            }
        }
        if (!haveFrame || !sawScan)
        {
            why = "JPEG without image data";
            return false;
        }
        why.clear();
        width = static_cast<uint32_t>(width_);
        height = static_cast<uint32_t>(height_);
        if (progressive_)
        {
            finishProgressive();
        }
        output(rgba);
        return true;
    }

  private:
    const uint8_t* p_;
    const uint8_t* end_;
    int width_ = 0, height_ = 0;
    int hMax_ = 1, vMax_ = 1, mcuX_ = 0, mcuY_ = 0;
    std::vector<Component> components_;
    uint16_t dequant_[4][64] = {};
    bool dequantPresent_[4] = {false, false, false, false};
    Huffman dc_[4], ac_[4];
    int restartInterval_ = 0;
    bool jfif_ = false;
    bool progressive_ = false;
    int eobRun_ = 0;
    int adobeTransform_ = -1;
    int rgbIds_ = 0;

    // entropy-coded segment reader
    uint32_t bitBuffer_ = 0;
    int bitCount_ = 0;
    int pendingMarker_ = -1; // a marker met inside entropy data
    bool exhausted_ = false;

    int nextMarker()
    {
        if (pendingMarker_ >= 0)
        {
            int const m = pendingMarker_;
            pendingMarker_ = -1;
            return m;
        }
        while (p_ < end_)
        {
            if (*p_++ != 0xFF)
            {
                continue; // garbage between segments is skipped
            }
            while (p_ < end_ && *p_ == 0xFF)
            {
                p_++;
            }
            if (p_ >= end_)
            {
                return -1;
            }
            int const m = *p_++;
            if (m != 0)
            {
                return m;
            }
        }
        return -1;
    }
    bool segment(const uint8_t*& body, int& length)
    {
        if (end_ - p_ < 2)
        {
            return false;
        }
        int const total = (p_[0] << 8) | p_[1];
        if (total < 2 || end_ - p_ < total)
        {
            return false;
        }
        body = p_ + 2;
        length = total - 2;
        p_ += total;
        return true;
    }

    bool quantTables(std::string& why)
    {
        const uint8_t* seg;
        int len;
        if (!segment(seg, len))
        {
            why = "truncated DQT";
            return false;
        }
        while (len > 0)
        {
            int const precision = seg[0] >> 4, table = seg[0] & 15;
            int const need = 1 + (precision != 0 ? 128 : 64);
            if (precision > 1 || table > 3 || len < need)
            {
                why = "bad DQT";
                return false;
            }
            for (int i = 0; i < 64; i++)
            {
                dequant_[table][DEZIGZAG[i]] =
                    static_cast<uint16_t>(precision != 0 ? ((seg[1 + 2 * i] << 8) | seg[2 + 2 * i]) : seg[1 + i]);
            }
            dequantPresent_[table] = true;
            seg += need;
            len -= need;
        }
        return true;
    }
    bool huffmanTables(std::string& why)
    {
        const uint8_t* seg;
        int len;
        if (!segment(seg, len))
        {
            why = "truncated DHT";
            return false;
        }
        while (len > 0)
        {
            if (len < 17)
            {
                why = "bad DHT";
                return false;
            }
            int const kind = seg[0] >> 4, table = seg[0] & 15;
            int total = 0;
            for (int i = 0; i < 16; i++)
            {
                total += seg[1 + i];
            }
            if (kind > 1 || table > 3 || total > 256 || len < 17 + total || !buildHuffman(kind == 0 ? dc_[table] : ac_[table], seg + 1, seg + 17, total))
            {
                why = "bad DHT";
                return false;
            }
            seg += 17 + total;
            len -= 17 + total;
        }
        return true;
    }
    bool frameHeader(std::string& why)
    {
        const uint8_t* seg;
        int len;
        if (!segment(seg, len) || len < 6)
        {
            why = "truncated SOF";
            return false;
        }
        int const precision = seg[0];
        height_ = (seg[1] << 8) | seg[2];
        width_ = (seg[3] << 8) | seg[4];
        int const n = seg[5];
        if (precision != 8)
        {
            why = "only 8-bit JPEG is decoded";
            return false;
        }
        if (width_ == 0 || height_ == 0)
        {
            why = "JPEG with a zero dimension";
            return false;
        }
        if ((n != 1 && n != 3 && n != 4) || len != 6 + 3 * n)
        {
            why = "bad SOF component count";
            return false;
        }
        if (static_cast<uint64_t>(width_) * static_cast<uint64_t>(height_) > (1ull << 28))
        {
            why = "JPEG too large";
            return false;
        }
        components_.assign(static_cast<size_t>(n), Component{});
        static const char rgb[3] = {'R', 'G', 'B'};
        for (int i = 0; i < n; i++)
        {
            Component& c = components_[static_cast<size_t>(i)];
            c.id = seg[6 + 3 * i];
            c.h = seg[7 + 3 * i] >> 4;
            c.v = seg[7 + 3 * i] & 15;
            c.tq = seg[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
            {
                why = "bad SOF sampling factors";
                return false;
            }
            if (n == 3 && c.id == rgb[i])
            {
                rgbIds_++;
            }
            hMax_ = c.h > hMax_ ? c.h : hMax_;
            vMax_ = c.v > vMax_ ? c.v : vMax_;
        }
        for (Component const& c : components_)
        {
            if (hMax_ % c.h != 0 || vMax_ % c.v != 0)
            {
                why = "JPEG sampling factors that do not divide the largest one";
                return false;
            }
        }
        mcuX_ = (width_ + 8 * hMax_ - 1) / (8 * hMax_);
        mcuY_ = (height_ + 8 * vMax_ - 1) / (8 * vMax_);
        for (Component& c : components_)
        {
            c.x = (width_ * c.h + hMax_ - 1) / hMax_;
            c.y = (height_ * c.v + vMax_ - 1) / vMax_;
            c.w2 = mcuX_ * c.h * 8;
            c.h2 = mcuY_ * c.v * 8;
            c.data.assign(static_cast<size_t>(c.w2) * static_cast<size_t>(c.h2), 0);
            if (progressive_)
            {
                c.coefficientWidth = c.w2 / 8;
                c.coefficientHeight = c.h2 / 8;
                c.coefficients.assign(static_cast<size_t>(c.w2) * static_cast<size_t>(c.h2), 0);
            }
        }
        return true;
    }

    // ---- bits of the entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other marker ends the data (zeros follow)
    void fill()
    {
        while (bitCount_ <= 24)
        {
            uint32_t byte = 0;
            if (!exhausted_ && pendingMarker_ < 0 && p_ < end_)
            {
                byte = *p_++;
                if (byte == 0xFF)
                {
                    int next = p_ < end_ ? *p_ : 0xD9;
                    while (next == 0xFF && p_ + 1 < end_)
                    {
                        p_++;
                        next = *p_;
                    }
                    if (p_ < end_)
                    {
                        p_++;
                    }
                    if (next != 0)
                    {
                        pendingMarker_ = next;
                        byte = 0;
                    }
                }
            }
            else if (p_ >= end_)
            {
                exhausted_ = true;
            }
            bitBuffer_ |= byte << (24 - bitCount_);
            bitCount_ += 8;
        }
    }
    int decodeSymbol(const Huffman& h)
    {
        if (bitCount_ < 16)
        {
            fill();
        }
        unsigned const look = bitBuffer_ >> 23;
        int len = h.fastLength[look];
        if (len != 0)
        {
            bitBuffer_ <<= len;
            bitCount_ -= len;
            return h.fastSymbol[look];
        }
        unsigned const top = bitBuffer_ >> 16;
        for (len = 10; len <= 16; len++)
        {
            int const code = static_cast<int>(top >> (16 - len));
            if (h.maxCode[len] >= 0 && code <= h.maxCode[len] && code >= h.firstCode[len])
            {
                int const index = h.firstIndex[len] + code - h.firstCode[len];
                bitBuffer_ <<= len;
                bitCount_ -= len;
                return index < h.count ? h.symbols[index] : -1;
            }
        }
        return -1;
    }
    // the next n bits as a signed difference (T.81 F.2.2.1 EXTEND)
    int receiveExtend(int n)
    {
        if (n == 0)
        {
            return 0;
        }
        if (bitCount_ < n)
        {
            fill();
        }
        int const value = static_cast<int>(bitBuffer_ >> (32 - n));
        bitBuffer_ <<= n;
        bitCount_ -= n;
        return value < (1 << (n - 1)) ? value - (1 << n) + 1 : value;
    }
    void resetEntropy()
    {
        bitBuffer_ = 0;
        bitCount_ = 0;
        eobRun_ = 0;
        for (Component& c : components_)
        {
            c.prediction = 0;
        }
    }
    int getBit() { return receiveBits(1); }
    int receiveBits(int n)
    {
        if (n == 0)
        {
            return 0;
        }
        if (bitCount_ < n)
        {
            fill();
        }
        int const value = static_cast<int>(bitBuffer_ >> (32 - n));
        bitBuffer_ <<= n;
        bitCount_ -= n;
        return value;
    }

    // T.81 G.1.2: one block of a progressive scan. DC scans (specStart == 0) code the prediction difference shifted by
    // the point transform, or one refinement bit; AC scans code bands of coefficients with end-of-band runs, first pass or
    // refinement (correction bits for coefficients that are already non-zero, newly non-zero ones of magnitude 1 << low).
    bool decodeProgressiveDc(Component& c, int16_t* data, int high, int low)
    {
        if (high == 0)
        {
            int const t = decodeSymbol(dc_[c.td]);
            if (t < 0 || t > 15)
            {
                return false;
            }
            c.prediction = static_cast<int>(static_cast<unsigned>(c.prediction) + static_cast<unsigned>(receiveExtend(t)));
            data[0] = static_cast<int16_t>(static_cast<unsigned>(c.prediction) * (1u << low));
        }
        else if (getBit() != 0)
        {
            data[0] = static_cast<int16_t>(data[0] + static_cast<int16_t>(1 << low));
        }
        return true;
    }
    bool decodeProgressiveAc(Component& c, int16_t* data, int specStart, int specEnd, int high, int low)
    {
        const Huffman& table = ac_[c.ta];
        if (high == 0)
        {
            if (eobRun_ != 0)
            {
                eobRun_--;
                return true;
            }
            int k = specStart;
            do
            {
                int const rs = decodeSymbol(table);
                if (rs < 0)
                {
                    return false;
                }
                int const size = rs & 15, run = rs >> 4;
                if (size == 0)
                {
                    if (run < 15)
                    {
                        eobRun_ = (1 << run);
                        if (run != 0)
                        {
                            eobRun_ += receiveBits(run);
                        }
                        eobRun_--;
                        break;
                    }
                    k += 16;
                }
                else
                {
                    k += run;
                    if (k > 63)
                    {
                        return false;
                    }
                    data[DEZIGZAG[k++]] = static_cast<int16_t>(static_cast<unsigned>(receiveExtend(size)) * (1u << low));
                }
            } while (k <= specEnd);
            return true;
        }
        int16_t const bit = static_cast<int16_t>(1 << low);
        auto correct = [&](int16_t& p) {
            if (getBit() != 0 && (p & bit) == 0)
            {
                p = static_cast<int16_t>(p > 0 ? p + bit : p - bit);
            }
        };
        if (eobRun_ != 0)
        {
            eobRun_--;
            for (int k = specStart; k <= specEnd; k++)
            {
                int16_t& p = data[DEZIGZAG[k]];
                if (p != 0)
                {
                    correct(p);
                }
            }
            return true;
        }
        int k = specStart;
        do
        {
            int const rs = decodeSymbol(table);
            if (rs < 0)
            {
                return false;
            }
            int size = rs & 15, run = rs >> 4;
            int16_t value = 0;
            if (size == 0)
            {
                if (run < 15)
                {
                    eobRun_ = (1 << run) - 1;
                    if (run != 0)
                    {
                        eobRun_ += receiveBits(run);
                    }
                    run = 64; // the rest of the band only takes correction bits
                }
                // run == 15: sixteen zero coefficients are skipped, the sixteenth by the "value" 0 below
            }
            else
            {
                if (size != 1)
                {
                    return false;
                }
                value = getBit() != 0 ? bit : static_cast<int16_t>(-bit);
            }
            while (k <= specEnd)
            {
                int16_t& p = data[DEZIGZAG[k++]];
                if (p != 0)
                {
                    correct(p);
                }
                else
                {
                    if (run == 0)
                    {
                        p = value;
                        break;
                    }
                    run--;
                }
            }
        } while (k <= specEnd);
        return true;
    }
    void finishProgressive()
    {
        for (Component& c : components_)
        {
            if (!dequantPresent_[c.tq])
            {
                continue;
            }
            const uint16_t* dq = dequant_[c.tq];
            int const bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
            {
                for (int i = 0; i < bw; i++)
                {
                    int16_t* data = c.coefficients.data() + 64 * (static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(c.coefficientWidth));
                    for (int k = 0; k < 64; k++)
                    {
                        data[k] = static_cast<int16_t>(static_cast<unsigned>(static_cast<int>(data[k])) * dq[k]);
                    }
                    idct(c.data.data() + static_cast<size_t>(c.w2) * static_cast<size_t>(j * 8) + static_cast<size_t>(i * 8), c.w2, data);
                }
            }
        }
    }

    bool decodeBlock(Component& c, int16_t block[64])
    {
        std::memset(block, 0, 64 * sizeof(int16_t));
        const uint16_t* dq = dequant_[c.tq];
        int const t = decodeSymbol(dc_[c.td]);
        if (t < 0 || t > 15)
        {
            return false;
        }
        // (unsigned arithmetic: a corrupt stream may wrap around, as it does in stb_image, but must not be undefined)
        c.prediction = static_cast<int>(static_cast<unsigned>(c.prediction) + static_cast<unsigned>(receiveExtend(t)));
        block[0] = static_cast<int16_t>(static_cast<unsigned>(c.prediction) * dq[0]);
        int k = 1;
        while (k < 64)
        {
            int const rs = decodeSymbol(ac_[c.ta]);
            if (rs < 0)
            {
                return false;
            }
            int const run = rs >> 4, size = rs & 15;
            if (size == 0)
            {
                if (rs != 0xF0)
                {
                    break; // end of block
                }
                k += 16;
                continue;
            }
            k += run;
            if (k > 63)
            {
                return false;
            }
            int const where = DEZIGZAG[k++];
            block[where] = static_cast<int16_t>(static_cast<unsigned>(receiveExtend(size)) * dq[where]);
        }
        return true;
    }

    using Wide = int64_t; // 32 bits hold every value of a valid stream; 64 keep corrupt ones defined
    static Wide fixed(double x) { return static_cast<Wide>(x * 4096.0 + 0.5); }
    // One 8-point pass of the inverse DCT; t0..t3 and x0..x3 as in the classic integer "islow" factorisation.
    static void idct1d(Wide s0, Wide s1, Wide s2, Wide s3, Wide s4, Wide s5, Wide s6, Wide s7, Wide& x0, Wide& x1, Wide& x2, Wide& x3,
                       Wide& t0, Wide& t1, Wide& t2, Wide& t3)
    {
        Wide p2 = s2, p3 = s6;
        Wide p1 = (p2 + p3) * fixed(0.5411961);
        t2 = p1 + p3 * fixed(-1.847759065);
        t3 = p1 + p2 * fixed(0.765366865);
        p2 = s0;
        p3 = s4;
        t0 = (p2 + p3) * 4096;
        t1 = (p2 - p3) * 4096;
        x0 = t0 + t3;
        x3 = t0 - t3;
        x1 = t1 + t2;
        x2 = t1 - t2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        Wide p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        Wide const p5 = (p3 + p4) * fixed(1.175875602);
        t0 = t0 * fixed(0.298631336);
        t1 = t1 * fixed(2.053119869);
        t2 = t2 * fixed(3.072711026);
        t3 = t3 * fixed(1.501321110);
        p1 = p5 + p1 * fixed(-0.899976223);
        p2 = p5 + p2 * fixed(-2.562915447);
        p3 = p3 * fixed(-1.961570560);
        p4 = p4 * fixed(-0.390180644);
        t3 += p1 + p4;
        t2 += p2 + p3;
        t1 += p2 + p4;
        t0 += p1 + p3;
    }
    static uint8_t clamp(Wide v) { return static_cast<uint8_t>(v < 0 ? 0 : v > 255 ? 255 : v); }
    static void idct(uint8_t* out, int stride, const int16_t d[64])
    {
        Wide v[64];
        for (int i = 0; i < 8; i++)
        {
            if (d[i + 8] == 0 && d[i + 16] == 0 && d[i + 24] == 0 && d[i + 32] == 0 && d[i + 40] == 0 && d[i + 48] == 0 && d[i + 56] == 0)
            {
                Wide const dc = d[i] * 4;
                for (int r = 0; r < 8; r++)
                {
                    v[i + 8 * r] = dc;
                }
                continue;
            }
            Wide x0, x1, x2, x3, t0, t1, t2, t3;
            idct1d(d[i], d[i + 8], d[i + 16], d[i + 24], d[i + 32], d[i + 40], d[i + 48], d[i + 56], x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512;
            x1 += 512;
            x2 += 512;
            x3 += 512;
            v[i] = (x0 + t3) >> 10;
            v[i + 56] = (x0 - t3) >> 10;
            v[i + 8] = (x1 + t2) >> 10;
            v[i + 48] = (x1 - t2) >> 10;
            v[i + 16] = (x2 + t1) >> 10;
            v[i + 40] = (x2 - t1) >> 10;
            v[i + 24] = (x3 + t0) >> 10;
            v[i + 32] = (x3 - t0) >> 10;
        }
        for (int r = 0; r < 8; r++)
        {
            const Wide* w = v + 8 * r;
            uint8_t* o = out + r * stride;
            Wide x0, x1, x2, x3, t0, t1, t2, t3;
            idct1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], x0, x1, x2, x3, t0, t1, t2, t3);
            Wide const bias = 65536 + (128 << 17);
            x0 += bias;
            x1 += bias;
            x2 += bias;
            x3 += bias;
            o[0] = clamp((x0 + t3) >> 17);
            o[7] = clamp((x0 - t3) >> 17);
            o[1] = clamp((x1 + t2) >> 17);
            o[6] = clamp((x1 - t2) >> 17);
            o[2] = clamp((x2 + t1) >> 17);
            o[5] = clamp((x2 - t1) >> 17);
            o[3] = clamp((x3 + t0) >> 17);
            o[4] = clamp((x3 - t0) >> 17);
        }
    }

    bool scan(std::string& why)
    {
        const uint8_t* seg;
        int len;
        if (!segment(seg, len) || len < 1)
        {
            why = "truncated SOS";
            return false;
        }
        int const n = seg[0];
        if (n < 1 || n > static_cast<int>(components_.size()) || len != 4 + 2 * n)
        {
            why = "bad SOS";
            return false;
        }
        std::vector<Component*> order;
        for (int i = 0; i < n; i++)
        {
            int const id = seg[1 + 2 * i], tables = seg[2 + 2 * i];
            Component* found = nullptr;
            for (Component& c : components_)
            {
                if (c.id == id)
                {
                    found = &c;
                }
            }
            if (found == nullptr || (tables >> 4) > 3 || (tables & 15) > 3)
            {
                why = "bad SOS component";
                return false;
            }
            found->td = tables >> 4;
            found->ta = tables & 15;
            order.push_back(found);
        }
        int const specStart = seg[1 + 2 * n], specEnd = seg[2 + 2 * n], high = seg[3 + 2 * n] >> 4, low = seg[3 + 2 * n] & 15;
        if (progressive_)
        {
            if (specStart > 63 || specEnd > 63 || specStart > specEnd || high > 13 || low > 13 || (specStart == 0 && specEnd != 0) ||
                (specStart != 0 && n != 1))
            {
                why = "bad SOS spectral selection for a progressive JPEG";
                return false;
            }
        }
        else if (specStart != 0 || specEnd != 63 || high != 0 || low != 0)
        {
            why = "bad SOS spectral selection for a sequential JPEG";
            return false;
        }
        for (Component* c : order)
        {
            // a progressive DC scan needs no AC table and an AC scan no DC table; the quantisation table may even follow
            bool const needDc = !progressive_ || (specStart == 0 && high == 0), needAc = !progressive_ || specStart != 0;
            if ((needDc && !dc_[c->td].present) || (needAc && !ac_[c->ta].present) || (!progressive_ && !dequantPresent_[c->tq]))
            {
                why = "JPEG scan refers to a table that was not defined";
                return false;
            }
        }
        resetEntropy();
        exhausted_ = false;
        int todo = restartInterval_ != 0 ? restartInterval_ : 0x7fffffff;
        int16_t block[64];
        auto restart = [&]() {
            // the next marker, if a restart marker, resets the predictions; anything else ends the scan early
            if (bitCount_ < 24)
            {
                fill();
            }
            if (pendingMarker_ >= 0xD0 && pendingMarker_ <= 0xD7)
            {
                pendingMarker_ = -1;
                resetEntropy();
                return true;
            }
            return false;
        };
        if (n == 1)
        {
            Component& c = *order[0];
            int const bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
            {
                for (int i = 0; i < bw; i++)
                {
                    if (progressive_)
                    {
                        int16_t* data = c.coefficients.data() + 64 * (static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(c.coefficientWidth));
                        if (!(specStart == 0 ? decodeProgressiveDc(c, data, high, low) : decodeProgressiveAc(c, data, specStart, specEnd, high, low)))
                        {
                            why = "corrupt JPEG entropy data";
                            return false;
                        }
                    }
                    else
                    {
                        if (!decodeBlock(c, block))
                        {
                            why = "corrupt JPEG entropy data";
                            return false;
                        }
                        idct(c.data.data() + static_cast<size_t>(c.w2) * static_cast<size_t>(j * 8) + static_cast<size_t>(i * 8), c.w2, block);
                    }
                    if (--todo <= 0)
                    {
                        if (!restart())
                        {
                            return finishScan();
                        }
                        todo = restartInterval_;
                    }
                }
            }
        }
        else
        {
            for (int j = 0; j < mcuY_; j++)
            {
                for (int i = 0; i < mcuX_; i++)
                {
                    for (Component* c : order)
                    {
                        for (int y = 0; y < c->v; y++)
                        {
                            for (int x = 0; x < c->h; x++)
                            {
                                size_t const bx = static_cast<size_t>((i * c->h + x) * 8), by = static_cast<size_t>((j * c->v + y) * 8);
                                if (progressive_)
                                {
                                    // interleaved progressive scans carry DC only (checked above)
                                    int16_t* data = c->coefficients.data() + 64 * (bx / 8 + (by / 8) * static_cast<size_t>(c->coefficientWidth));
                                    if (!decodeProgressiveDc(*c, data, high, low))
                                    {
                                        why = "corrupt JPEG entropy data";
                                        return false;
                                    }
                                    continue;
                                }
                                if (!decodeBlock(*c, block))
                                {
                                    why = "corrupt JPEG entropy data";
                                    return false;
                                }
                                idct(c->data.data() + static_cast<size_t>(c->w2) * by + bx, c->w2, block);
                            }
                        }
                    }
                    if (--todo <= 0)
                    {
                        if (!restart())
                        {
                            return finishScan();
                        }
                        todo = restartInterval_;
                    }
                }
            }
        }
        return finishScan();
    }
    // after the last MCU: skip to the marker that ends the entropy-coded data
    bool finishScan()
    {
        bitBuffer_ = 0;
        bitCount_ = 0;
        if (pendingMarker_ < 0)
        {
            // entropy data not consumed to its end: scan forward for the next marker
            while (p_ + 1 < end_)
            {
                if (p_[0] == 0xFF && p_[1] != 0 && p_[1] != 0xFF && !(p_[1] >= 0xD0 && p_[1] <= 0xD7))
                {
                    break;
                }
                p_++;
            }
        }
        exhausted_ = false;
        return true;
    }

    // ---- upsampling + colour conversion, row by row
    static void rowH2(uint8_t* out, const uint8_t* in, int w)
    {
        if (w == 1)
        {
            out[0] = out[1] = in[0];
            return;
        }
        out[0] = in[0];
        out[1] = static_cast<uint8_t>((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++)
        {
            int const n = 3 * in[i] + 2;
            out[i * 2] = static_cast<uint8_t>((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = static_cast<uint8_t>((n + in[i + 1]) >> 2);
        }
        out[i * 2] = static_cast<uint8_t>((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
    }
    static void rowHV2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w)
    {
        if (w == 1)
        {
            out[0] = out[1] = static_cast<uint8_t>((3 * near[0] + far[0] + 2) >> 2);
            return;
        }
        int t1 = 3 * near[0] + far[0];
        out[0] = static_cast<uint8_t>((t1 + 2) >> 2);
        for (int i = 1; i < w; i++)
        {
            int const t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = static_cast<uint8_t>((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = static_cast<uint8_t>((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = static_cast<uint8_t>((t1 + 2) >> 2);
    }

    void output(std::vector<uint8_t>& rgba)
    {
        size_t const n = components_.size();
        rgba.assign(static_cast<size_t>(width_) * static_cast<size_t>(height_) * 4, 255);
        struct Resample
        {
            int hs, vs, ystep, ypos, wLow;
            const uint8_t* line0;
            const uint8_t* line1;
            std::vector<uint8_t> buffer;
        };
        std::vector<Resample> rs(n);
        for (size_t k = 0; k < n; k++)
        {
            Component const& c = components_[k];
            Resample& r = rs[k];
            r.hs = hMax_ / c.h;
            r.vs = vMax_ / c.v;
            r.ystep = r.vs >> 1;
            r.ypos = 0;
            r.wLow = (width_ + r.hs - 1) / r.hs;
            r.line0 = r.line1 = c.data.data();
            r.buffer.assign(static_cast<size_t>(r.wLow) * static_cast<size_t>(r.hs) + 8, 0);
        }
        bool const isRgb = n == 3 && (rgbIds_ == 3 || (adobeTransform_ == 0 && !jfif_));
        std::vector<const uint8_t*> rows(n);
        for (int j = 0; j < height_; j++)
        {
            for (size_t k = 0; k < n; k++)
            {
                Resample& r = rs[k];
                bool const bottom = r.ystep >= (r.vs >> 1);
                const uint8_t* near = bottom ? r.line1 : r.line0;
                const uint8_t* far = bottom ? r.line0 : r.line1;
                if (r.hs == 1 && r.vs == 1)
                {
                    rows[k] = near;
                }
                else
                {
                    uint8_t* out = r.buffer.data();
                    if (r.hs == 1 && r.vs == 2)
                    {
                        for (int i = 0; i < r.wLow; i++)
                        {
                            out[i] = static_cast<uint8_t>((3 * near[i] + far[i] + 2) >> 2);
                        }
                    }
                    else if (r.hs == 2 && r.vs == 1)
                    {
                        rowH2(out, near, r.wLow);
                    }
                    else if (r.hs == 2 && r.vs == 2)
                    {
                        rowHV2(out, near, far, r.wLow);
                    }
                    else
                    {
                        for (int i = 0; i < r.wLow; i++)
                        {
                            for (int s = 0; s < r.hs; s++)
                            {
                                out[i * r.hs + s] = near[i];
                            }
                        }
                    }
                    rows[k] = out;
                }
                if (++r.ystep >= r.vs)
                {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < components_[k].y)
                    {
                        r.line1 += components_[k].w2;
                    }
                }
            }
            uint8_t* out = rgba.data() + static_cast<size_t>(j) * static_cast<size_t>(width_) * 4;
            if (n == 1)
            {
                for (int i = 0; i < width_; i++)
                {
                    out[i * 4] = out[i * 4 + 1] = out[i * 4 + 2] = rows[0][i];
                }
            }
            else if (n == 4)
            {
                // Adobe four-component files as stb_image maps them to RGB: transform 0 = CMYK, 2 = YCCK (YCbCr first, then
                // the inverted result times K), anything else = YCbCr with the fourth component ignored.
                auto blinn = [](unsigned x, unsigned y) {
                    unsigned const t = x * y + 128u;
                    return static_cast<uint8_t>((t + (t >> 8)) >> 8);
                };
                auto f2f = [](double x) { return static_cast<int>(static_cast<unsigned>(static_cast<int>(x * 4096.0 + 0.5)) << 8); };
                int const crR = f2f(1.40200), crG = -f2f(0.71414), cbG = -f2f(0.34414), cbB = f2f(1.77200);
                for (int i = 0; i < width_; i++)
                {
                    unsigned const k = rows[3][i];
                    if (adobeTransform_ == 0)
                    {
                        out[i * 4] = blinn(rows[0][i], k);
                        out[i * 4 + 1] = blinn(rows[1][i], k);
                        out[i * 4 + 2] = blinn(rows[2][i], k);
                        continue;
                    }
                    int const yFixed = (rows[0][i] << 20) + (1 << 19);
                    int const cb = rows[1][i] - 128, cr = rows[2][i] - 128;
                    int const r = (yFixed + cr * crR) >> 20;
                    int const g = (yFixed + cr * crG + static_cast<int>(static_cast<unsigned>(cb * cbG) & 0xffff0000u)) >> 20;
                    int const b = (yFixed + cb * cbB) >> 20;
                    if (adobeTransform_ == 2)
                    {
                        out[i * 4] = blinn(255u - clamp(r), k);
                        out[i * 4 + 1] = blinn(255u - clamp(g), k);
                        out[i * 4 + 2] = blinn(255u - clamp(b), k);
                    }
                    else
                    {
                        out[i * 4] = clamp(r);
                        out[i * 4 + 1] = clamp(g);
                        out[i * 4 + 2] = clamp(b);
                    }
                }
            }
            else if (isRgb)
            {
                for (int i = 0; i < width_; i++)
                {
                    out[i * 4] = rows[0][i];
                    out[i * 4 + 1] = rows[1][i];
                    out[i * 4 + 2] = rows[2][i];
                }
            }
            else
            {
                auto f2f = [](double x) { return static_cast<int>(static_cast<unsigned>(static_cast<int>(x * 4096.0 + 0.5)) << 8); };
                int const crR = f2f(1.40200), crG = -f2f(0.71414), cbG = -f2f(0.34414), cbB = f2f(1.77200);
                for (int i = 0; i < width_; i++)
                {
                    int const yFixed = (rows[0][i] << 20) + (1 << 19);
                    int const cb = rows[1][i] - 128, cr = rows[2][i] - 128;
                    int r = yFixed + cr * crR;
                    int g = yFixed + cr * crG + static_cast<int>(static_cast<unsigned>(cb * cbG) & 0xffff0000u);
                    int b = yFixed + cb * cbB;
                    r >>= 20;
                    g >>= 20;
                    b >>= 20;
                    out[i * 4] = clamp(r);
                    out[i * 4 + 1] = clamp(g);
                    out[i * 4 + 2] = clamp(b);
                }
            }
        }
    }
};
} // namespace

bool szg::decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& why)
{
    Decoder decoder(data, size);
    return decoder.decode(width, height, rgba, why);
}
