// tokenizer.cpp — the reference's placeholder tokenizer and its inverse (host only; declared in engine.h).
#include <string>
#include <vector>
#include "common.h"
#include "tokenizer.h"

// LLMEngine::tokenize, llm_engine.rs:220-230: text.chars().map(|c| c as u32 as i64).take(100)
int nvr::tokenize(const char *utf8, size_t nbytes, std::vector<int64_t> &out) {
    out.clear();
    const unsigned char *p = (const unsigned char *)utf8;
    size_t i = 0;
    while (i < nbytes && out.size() < NVR_TOKENIZE_MAX_CHARS) {
        const unsigned c = p[i];
        unsigned cp; size_t len;
        if (c < 0x80) { cp = c; len = 1; }
        else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; len = 2; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; len = 3; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; len = 4; }
        else return nvr::fail(NVR_ERR_INVALID_ARG, "tokenize: invalid UTF-8 lead byte 0x%02x at offset %zu", c, i);
        if (i + len > nbytes) return nvr::fail(NVR_ERR_INVALID_ARG, "tokenize: truncated UTF-8 sequence at offset %zu", i);
        for (size_t k = 1; k < len; ++k) {
            if ((p[i + k] & 0xC0) != 0x80) return nvr::fail(NVR_ERR_INVALID_ARG, "tokenize: invalid UTF-8 continuation byte at offset %zu", i + k);
            cp = (cp << 6) | (p[i + k] & 0x3F);
        }
        // what a Rust String can never hold: overlong forms, surrogates, values past U+10FFFF
        static const unsigned kMin[5] = {0, 0, 0x80, 0x800, 0x10000};
        if (cp < kMin[len] || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF))
            return nvr::fail(NVR_ERR_INVALID_ARG, "tokenize: invalid UTF-8 scalar value U+%04X at offset %zu", cp, i);
        out.push_back((int64_t)cp);
        i += len;
    }
    return NVR_OK;
}

void nvr::detokenize(const int64_t *ids, size_t n, std::string &out) {
    out.clear();
    for (size_t i = 0; i < n; ++i) {
        uint32_t cp = (ids[i] < 0 || ids[i] > 0x10FFFF || (ids[i] >= 0xD800 && ids[i] <= 0xDFFF)) ? 0xFFFDu : (uint32_t)ids[i];
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
}

