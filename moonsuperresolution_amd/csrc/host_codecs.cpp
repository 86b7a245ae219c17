// Host-only helpers of libmoonsr_hip.so (no device code): TIFF LZW codec for moonsuperresolution_amd/geotiff.py.
// TIFF 6.0 LZW: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, first free code 258, the
// width grows one code EARLY (at 511 / 1023 / 2047 entries), the table is cleared when it holds 4094 entries.
#include "../../include/moonsr.h"

#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

int64_t msr_lzw_decode(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t n_out) {
    if (!in || !out || n_in < 0 || n_out < 0) return -1;
    struct Entry { int32_t prev; uint8_t ch; uint8_t first; uint16_t len; };
    static thread_local std::vector<Entry> table(4096);
    for (int i = 0; i < 256; ++i) table[i] = {-1, (uint8_t)i, (uint8_t)i, 1};
    int next = 258, width = 9;
    int64_t ip = 0, op = 0;
    uint32_t bitbuf = 0;
    int nbits = 0;
    int prev = -1;
    while (true) {
        while (nbits < width) {
            if (ip >= n_in) return op;            // truncated stream: return what we have
            bitbuf = (bitbuf << 8) | in[ip++];
            nbits += 8;
        }
        const int code = (int)((bitbuf >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; prev = -1; continue; }
        int emit = code;
        uint8_t extra = 0;
        bool kwkwk = false;
        if (prev < 0) {
            if (code >= 256) return -1;
        } else if (code >= next) {
            if (code != next) return -1;
            emit = prev;                           // KwKwK: prev string + its own first byte
            extra = table[prev].first;
            kwkwk = true;
        }
        const int len = table[emit].len + (kwkwk ? 1 : 0);
        if (op + len > n_out) return -1;
        int64_t pos = op + table[emit].len - 1;
        for (int c = emit; c >= 0; c = table[c].prev) out[pos--] = table[c].ch;
        if (kwkwk) out[op + len - 1] = extra;
        if (prev >= 0 && next < 4096) {
            table[next] = {prev, kwkwk ? extra : table[emit].first, table[prev].first, (uint16_t)(table[prev].len + 1)};
            ++next;
        }
        op += len;
        prev = code;
        if (next == 511 || next == 1023 || next == 2047) ++width;   // early change
    }
    return op;
}

int64_t msr_lzw_encode(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t cap) {
    if (!in || !out || n_in < 0 || cap < 8) return -1;
    // trie over (prefix code, byte): child / sibling lists
    static thread_local std::vector<int16_t> child(4096), sibling(4096);
    static thread_local std::vector<uint8_t> chr(4096);
    int64_t op = 0;
    uint32_t bitbuf = 0;
    int nbits = 0, width = 9, next = 258;
    auto put = [&](int code) -> bool {
        bitbuf = (bitbuf << width) | (uint32_t)code;
        nbits += width;
        while (nbits >= 8) {
            if (op >= cap) return false;
            out[op++] = (uint8_t)(bitbuf >> (nbits - 8));
            nbits -= 8;
        }
        return true;
    };
    auto reset = [&]() {
        for (int i = 0; i < 4096; ++i) { child[i] = -1; sibling[i] = -1; }
        next = 258;
        width = 9;
    };
    reset();
    if (!put(256)) return -1;
    if (n_in == 0) { if (!put(257)) return -1; if (nbits) { if (op >= cap) return -1; out[op++] = (uint8_t)(bitbuf << (8 - nbits)); } return op; }
    int cur = in[0];
    for (int64_t i = 1; i < n_in; ++i) {
        const uint8_t c = in[i];
        int k = child[cur];
        while (k >= 0 && chr[k] != c) k = sibling[k];
        if (k >= 0) { cur = k; continue; }
        if (!put(cur)) return -1;
        chr[next] = c;
        sibling[next] = child[cur];
        child[cur] = (int16_t)next;
        ++next;
        if (next == 512 || next == 1024 || next == 2048) ++width;   // the decoder is one entry behind: same early change
        if (next == 4095) {                                           // table full: clear
            if (!put(256)) return -1;
            reset();
        }
        cur = c;
    }
    if (!put(cur)) return -1;
    // the decoder adds an entry for this last code too: mirror its width bump before EOI
    ++next;
    if (next == 512 || next == 1024 || next == 2048) ++width;
    if (!put(257)) return -1;
    if (nbits) { if (op >= cap) return -1; out[op++] = (uint8_t)((bitbuf << (8 - nbits)) & 0xFF); }
    return op;
}

}  // extern "C"
