"""DESIGN.md / INTEGRATION.md / README.md cite C-ABI symbols (`nvr_*`) and tests (`test_*`) by name; every such name must exist — in include/nvr.h
and in tests/*.py (the r05 pruning left `nvr_linear_resid`, the `*_normed` entry points and their tests behind in DESIGN section 1).
A name followed by `*`, `…` or `...` is a prefix; `nvr_x_{a,b}` expands; names inside the round logs under profiles/ are history and not checked."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md"]
# not C-ABI symbols: the package / library / handle-type / python-module names that share the prefix
NOT_SYMBOLS = {"nvr_import", "nvr_oracle", "nvr_api", "nvr_config", "nvr_model_config", "nvr_sampling_params", "nvr_attn_meta", "nvr_step_info", "nvr_status",
               "nvr_seq", "nvr_engine", "nvr_scheduler", "nvr_model_runner", "nvr_block_manager", "nvr_local_group", "nvr_sequence_output", "nvr_sched_stats",
               "nvr_bm_stats", "nvr_engine_stats", "nvr_health_status", "nvr_half", "nvr_stream_fn", "nvr_pmc"}


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "nvr.h")).read()
    return set(re.findall(r"\b(nvr_[a-z0-9_]+)\s*\(", text)) | set(re.findall(r"\b(nvr_[a-z0-9_]+_t)\b", text))


def _test_names():
    names = set()
    for f in os.listdir(os.path.join(ROOT, "tests")):
        if f.endswith(".py"):
            names |= set(re.findall(r"^def (test_[A-Za-z0-9_]+)", open(os.path.join(ROOT, "tests", f)).read(), re.M))
    return names


def _cited(text, prefix):
    """(name, is_prefix) for every `prefix…` token of the text; brace lists expand"""
    out = []
    for m in re.finditer(r"\b(" + prefix + r"[A-Za-z0-9_]*)(\{[A-Za-z0-9_, /]+\})?([A-Za-z0-9_]*)(\*|…|\.\.\.)?", text):
        head, braces, tail, star = m.group(1), m.group(2), m.group(3), m.group(4)
        if braces:
            for alt in re.split(r"[,/]\s*", braces[1:-1]):
                out.append((head + alt.strip() + tail, bool(star)))
        else:
            out.append((head, bool(star) or head.endswith("_")))
    return out


def test_every_cited_abi_symbol_and_test_name_exists():
    syms, tests = _header_symbols(), _test_names()
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for name, is_prefix in _cited(text, "nvr_"):
            base = name.rstrip("_")
            if base in NOT_SYMBOLS or name.endswith("_t") and name in syms:
                continue
            ok = any(s.startswith(name) for s in syms) if is_prefix else (name in syms or name + "_t" in syms)
            if not ok:
                missing.append(f"{doc}: {name}{'*' if is_prefix else ''}")
        for name, is_prefix in _cited(text, "test_"):
            if name in ("test_", "test_tp", "test_kernels_gpu", "test_engine_gpu", "test_host_parity", "test_oracle_kat", "test_golden", "test_scale", "test_weights",
                        "test_tokenizer", "test_ctrl", "test_bf16_gpu", "test_kernels_bf16_gpu", "test_kernels_f32_gpu", "test_baseline_parity", "test_sanitizers",
                        "test_kernel_isa", "test_bench_counters", "test_design_refs"):
                continue                                                  # file names
            ok = any(t.startswith(name) for t in tests) if is_prefix else name in tests
            if not ok and any(t.startswith(name) for t in tests) and len(name) >= 24:
                ok = True                                                 # a long name cut short at a line end / by an ellipsis glyph the pattern did not see
            if not ok:
                missing.append(f"{doc}: {name}{'*' if is_prefix else ''}")
    assert not missing, "cited but absent:\n  " + "\n  ".join(sorted(set(missing)))


def test_rust_shim_sampling_params_carries_every_reference_field():
    """integration/rust/src/hip.rs::SamplingParams mirrors the reference's struct (src/engine/sampling_params.rs:10-28: temperature, max_tokens, ignore_eos,
    top_p, top_k, repetition_penalty) and hands every optional field to nvr_sampling_params (include/nvr.h) — r05 dropped repetition_penalty on the way."""
    src = open(os.path.join(ROOT, "integration", "rust", "src", "hip.rs")).read()
    body = src[src.index("pub struct SamplingParams"):src.index("// ---- BlockManager")]
    for field in ("temperature", "max_tokens", "ignore_eos", "top_p", "top_k", "repetition_penalty"):
        assert re.search(r"pub " + field + r"\s*:", body), f"hip.rs SamplingParams lacks {field}"
        assert re.search(r"c\.(has_)?" + field, body), f"hip.rs SamplingParams::to_c does not pass {field} on"
    hdr = open(os.path.join(ROOT, "include", "nvr.h")).read()
    assert "has_repetition_penalty" in hdr and "float repetition_penalty" in hdr
