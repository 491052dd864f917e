"""bench.py's in-run counter passes (roofline.traffic, prefill.mfma_busy_frac_pmc): the parsing of rocprofv3's counter table is pure and is
checked here on synthetic tables; the passes themselves run on the GPU box inside the bench."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("nvr_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _rows(kernel, counter, values, first_id=1):
    return [{"Dispatch_Id": str(first_id + i), "Kernel_Name": kernel, "Counter_Name": counter, "Counter_Value": str(v)} for i, v in enumerate(values)]


def test_attention_traffic_is_bytes_over_algorithmic_bytes_of_the_pass():
    m = bench.MODELS["qwen3-0.6b"]
    sh, B, P = m["shape"], m["batch"], m["prompt_len"]
    steps, warm = 6, 2
    n, L = steps + warm, sh["layers"]
    ctxs = [P + i for i in range(1, n + 1) for _ in range(L)] + [P + n + 1] * (2 * L)
    alg = lambda c: B * c * sh["kvh"] * sh["d"] * 2 * 2 + 2 * B * sh["h"] * sh["d"] * 2
    # a kernel that fetches exactly its algorithmic bytes: FETCH_SIZE is in KiB and counts half of them on gfx950
    rows = _rows("void nvr::k::attn_rows_kernel<128, 2, ...>(AttnParams)", "FETCH_SIZE", [alg(c) / 2048.0 for c in ctxs])
    rows += _rows("some_other_kernel", "FETCH_SIZE", [1e9] * 5, first_id=1000)
    got = bench.attention_traffic_from_rows(rows, "qwen3-0.6b", steps, warm)
    assert got["dispatches"] == L * (n + 2) and abs(got["traffic_over_algorithmic"] - 1.0) < 2e-4, got
    assert abs(got["mean_context"] - sum(ctxs) / len(ctxs)) < 0.01
    # 7 % over-fetch shows up as 1.07; a pass with the wrong number of dispatches is refused with a reason
    over = bench.attention_traffic_from_rows([dict(r, Counter_Value=str(float(r["Counter_Value"]) * 1.07)) for r in rows], "qwen3-0.6b", steps, warm)
    assert abs(over["traffic_over_algorithmic"] - 1.07) < 1e-3
    assert isinstance(bench.attention_traffic_from_rows(rows[:-20], "qwen3-0.6b", steps, warm), str)
    # the live timing as a replayed graph adds one more sweep at the final context
    more = _rows("void nvr::k::attn_rows_kernel<128, 2, ...>(AttnParams)", "FETCH_SIZE", [alg(P + n + 1) / 2048.0] * L, first_id=5000)
    got3 = bench.attention_traffic_from_rows(rows + more, "qwen3-0.6b", steps, warm)
    assert got3["dispatches"] == L * (n + 3) and abs(got3["traffic_over_algorithmic"] - 1.0) < 2e-4, got3


def test_prefill_busy_share_takes_the_last_prefill_step_only():
    rows = []
    did = [0]

    def disp(kernel, busy, grbm):
        did[0] += 1
        for c, v in (("SQ_VALU_MFMA_BUSY_CYCLES", busy), ("GRBM_GUI_ACTIVE", grbm)):
            rows.append({"Dispatch_Id": str(did[0]), "Kernel_Name": kernel, "Counter_Name": c, "Counter_Value": str(v)})
    for rep in range(2):                                   # warm-up prefill, then the measured one: only the second counts
        disp("embedding_kernel", 0, 8 * 1000)
        for l in range(3):
            scale = 1 if rep else 5                        # (the warm-up has other numbers)
            disp("gemm256_kernelILi3ELi128E", 1024 * 400 * scale, 8 * 1000)      # 40 % busy
            disp("flash_prefill_kernel<128, 2>", 1024 * 300 * scale, 8 * 1000)   # 30 %
            disp("gemm256_kernelILi1ELi0E", 1024 * 500 * scale, 8 * 1000)        # 50 %
            disp("rmsnorm_kernel", 0, 8 * 100)
    disp("embed_rmsnorm_kernel", 0, 8 * 10)                # first decode step: ends the prefill step
    disp("gemm256_kernelILi1ELi0E", 1024 * 999, 8 * 1000)
    got = bench.prefill_busy_from_rows(rows)
    assert got["per_kernel"]["qkv+RoPE+store"] == 0.4 and got["per_kernel"]["flash_prefill"] == 0.3 and got["per_kernel"]["o / down + residual"] == 0.5
    busy = 3 * (400 + 300 + 500) * 1024; cyc = 1000 + 3 * (3 * 1000 + 100)
    assert abs(got["step_weighted"] - busy / (1024 * cyc)) < 1e-3 and got["dispatches"] == 1 + 3 * 4
