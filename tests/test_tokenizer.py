"""CPU tests of the text boundary: the C-ABI placeholder tokenizer / detokenizer against the
oracle's restatement of LLMEngine::tokenize (llm_engine.rs:220-230)."""
import ctypes as C

import pytest
from hypothesis import given, settings, strategies as st

import nvr_import
from oracle import engine_oracle as eo

nvr = nvr_import.load()


def test_reference_placeholder_known_answers():
    # text.chars().map(|c| c as u32 as i64).take(100)
    assert eo.tokenize("Hello") == [72, 101, 108, 108, 111] == nvr.tokenize("Hello")
    assert nvr.tokenize("") == [] == eo.tokenize("")
    s = "héllo wörld €😀"
    assert nvr.tokenize(s) == eo.tokenize(s) == [104, 233, 108, 108, 111, 32, 119, 246, 114, 108, 100, 32, 8364, 128512]
    long = "ab€" * 50                                          # 150 chars, 250 bytes: the cut is at 100 CHARS, not bytes
    assert nvr.tokenize(long) == eo.tokenize(long) and len(nvr.tokenize(long)) == eo.TOKENIZE_MAX_CHARS == 100
    assert nvr.detokenize(nvr.tokenize(long)) == long[:100]


@settings(max_examples=300, deadline=None)
@given(st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=160))
def test_tokenize_matches_oracle_and_round_trips(text):
    ids = nvr.tokenize(text)
    assert ids == eo.tokenize(text)
    assert nvr.detokenize(ids) == eo.detokenize(ids) == text[:100]


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(min_value=-5, max_value=0x110005), max_size=64))
def test_detokenize_matches_oracle_on_arbitrary_ids(ids):
    assert nvr.detokenize(ids) == eo.detokenize(ids)


def test_non_scalar_ids_become_replacement_characters():
    assert nvr.detokenize([72, 0xD800, -1, 0x110000, 105]) == "H���i" == eo.detokenize([72, 0xD800, -1, 0x110000, 105])


@pytest.mark.parametrize("raw", [b"\xff", b"\x80abc", b"ab\xc3", b"\xc0\x80", b"\xe0\x80\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80",
                                 b"\xf8\x88\x80\x80\x80", b"a\xe2\x28\xa1"])
def test_malformed_utf8_is_rejected(raw):
    # a Rust String cannot hold these; over the C ABI they are an argument error, never a silent token
    with pytest.raises(nvr.NvrError) as ei:
        nvr.tokenize(raw)
    assert ei.value.code == -7 and "UTF-8" in str(ei.value)            # NVR_ERR_INVALID_ARG


def test_malformed_tail_beyond_the_100_char_cut_is_not_looked_at():
    assert nvr.tokenize(b"a" * 100 + b"\xff") == [97] * 100


def test_length_query_and_small_buffers():
    lib = nvr.lib()
    n = C.c_size_t()
    assert lib.nvr_tokenize(b"abc", 3, None, 0, C.byref(n)) == 0 and n.value == 3
    out = (C.c_int64 * 2)()
    assert lib.nvr_tokenize(b"abc", 3, out, 2, C.byref(n)) != 0 and "buffer holds 2" in nvr.last_error()
    ids = (C.c_int64 * 2)(8364, 65)
    assert lib.nvr_detokenize(ids, 2, None, 0, C.byref(n)) == 0 and n.value == 4
    buf = C.create_string_buffer(3)
    assert lib.nvr_detokenize(ids, 2, buf, 3, C.byref(n)) != 0 and "buffer holds 3" in nvr.last_error()
    buf = C.create_string_buffer(8)
    assert lib.nvr_detokenize(ids, 2, buf, 8, C.byref(n)) == 0 and buf.value == "€A".encode()
