// tokenizer.h — LLMEngine::tokenize (reference src/engine/llm_engine.rs:220-230): one id per Unicode scalar value of the first
// NVR_TOKENIZE_MAX_CHARS characters; detokenize is its inverse.  Host-only code (also built into the sanitizer self-test).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nvr {
int tokenize(const char *utf8, size_t nbytes, std::vector<int64_t> &out);      // llm_engine.rs:220-230
void detokenize(const int64_t *ids, size_t n, std::string &out);
}
